"""Workload for rocprofv3: the fp32 parity tower at b = 256 (and b = 32), inputs resident.
    rocprofv3 --kernel-trace --stats -d OUT -o f --output-format csv -- python3 tools/fp32_profile.py"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if "--lib" in sys.argv:   # a probe build of the library (tools/probe/build_variant.sh)
    from image_search_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from image_search_amd import synth
from image_search_amd.clip import Model, PRECISION_F32

cfg = synth.VitConfig.vit_l14()
path = os.path.join(tempfile.gettempdir(), f"fp32_profile_{os.getpid()}.safetensors")
synth.save_safetensors(synth.vit_weights(cfg, 0), path, {"num_attention_heads": cfg.heads})
m = Model.from_file(path, 0, PRECISION_F32)
os.unlink(path)
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
px = synth.preprocess_rgb8(synth.images_u8(100, n, cfg.image))
d_in = torch.from_numpy(px).cuda()
d_out = torch.empty((n, 768), dtype=torch.float32, device="cuda")
st = torch.cuda.Stream()
for _ in range(3):
    m.forward_device(d_in.data_ptr(), n, d_out.data_ptr(), st.cuda_stream)
st.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st)
for _ in range(3):
    m.forward_device(d_in.data_ptr(), n, d_out.data_ptr(), st.cuda_stream)
b.record(st)
st.synchronize()
print(f"fp32 b={n}: {a.elapsed_time(b) / 3:.2f} ms per forward")
m.close()
