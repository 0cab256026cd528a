"""What clock does the chip hold where attention runs?  (attn32 reads 162 us in the tower, 146-152 alone.)
mi_op_clock_probe (one wave, ~0.1 ms of dependent FMAs, delta s_memtime / delta s_memrealtime) enqueued
  idle      on a chip that has done nothing for a while
  tower     on the tower's stream right behind a 256-image bf16 forward, repeated behind 10 forwards back to back
  attention behind 200 stand-alone launches of the attention op (what the stand-alone bench measures)
    python tools/clock_probe.py > profiles/r03_clock_probe.json"""
import ctypes, json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth
from image_search_amd._lib import check, lib
from image_search_amd.clip import PRECISION_BF16, Model


def probe(stream=0):
    v = ctypes.c_float()
    check(lib().mi_op_clock_probe(0, stream, ctypes.byref(v)))
    return round(v.value, 1)


cfg = synth.VitConfig.vit_l14()
path = "/tmp/mi355clip_bench_vitl14_seed0.safetensors"
if not os.path.exists(path):
    synth.save_safetensors(synth.vit_weights(cfg, 0), path, {"num_attention_heads": cfg.heads})
out = {}
torch.cuda.init()
time.sleep(1.0)
out["idle_mhz"] = [probe() for _ in range(3)]
m = Model.from_file(path, 0, PRECISION_BF16)
st = torch.cuda.Stream()
d_img = torch.from_numpy(synth.preprocess_rgb8(synth.images_u8(5, 256, cfg.image))).cuda()
d_out = torch.empty((256, 768), dtype=torch.float32, device="cuda")
for parts in (2, 1):
    m.set_option("parts", parts)
    for _ in range(3):
        m.forward_device(d_img.data_ptr(), 256, d_out.data_ptr(), st.cuda_stream)
    st.synchronize()
    res = []
    for rep in range(3):
        for _ in range(10):
            m.forward_device(d_img.data_ptr(), 256, d_out.data_ptr(), st.cuda_stream)
        res.append(probe(st.cuda_stream))     # behind 10 forwards (0.4 s of MFMA load) on the same stream
    out[f"behind_10_forwards_parts{parts}_mhz"] = res
# samples taken WHILE the tower runs: the probe's one wave fits beside any kernel of the forward (a few VGPRs, no LDS)
m.set_option("parts", 2)
sb = torch.cuda.Stream()
for _ in range(25):
    m.forward_device(d_img.data_ptr(), 256, d_out.data_ptr(), st.cuda_stream)
samples = []
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.8 and len(samples) < 400:
    samples.append(probe(sb.cuda_stream))
st.synchronize()
samples = np.array(samples)
out["during_forwards_mhz"] = {"n": int(samples.size), "min": float(samples.min()), "p10": float(np.percentile(samples, 10)),
                              "median": float(np.median(samples)), "p90": float(np.percentile(samples, 90)), "max": float(samples.max())}
time.sleep(1.0)
out["idle_again_mhz"] = [probe() for _ in range(2)]
m.close()
print(json.dumps(out, indent=1))
