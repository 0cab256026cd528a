import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from image_search_amd import synth
from image_search_amd.clip import Model, PRECISION_BF16
cfg = synth.VitConfig.vit_l14()
path = os.path.join(tempfile.gettempdir(), f"two_{os.getpid()}.safetensors")
synth.save_safetensors(synth.vit_weights(cfg, 0), path, {"num_attention_heads": cfg.heads})
ms = [Model.from_file(path, 0, PRECISION_BF16) for _ in range(2)]
os.unlink(path)
n = 256
px = synth.preprocess_rgb8(synth.images_u8(100, n, cfg.image))
d_in = torch.from_numpy(px).cuda()
outs = [torch.empty((n, 768), dtype=torch.float32, device="cuda") for _ in range(2)]
sts = [torch.cuda.Stream() for _ in range(2)]
torch.cuda.synchronize()
def run(k, handles):
    for i in range(k):
        for h in handles:
            ms[h].forward_device(d_in.data_ptr(), n, outs[h].data_ptr(), sts[h].cuda_stream)
res = {}
for name, handles in (("one handle", [0]), ("two handles interleaved", [0, 1])):
    run(2, handles); torch.cuda.synchronize()
    ts = []
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(torch.cuda.current_stream())
        for s in sts: s.wait_event(e0)
        run(10, handles)
        evs = []
        for h in handles:
            e = torch.cuda.Event(); e.record(sts[h]); evs.append(e)
        for e in evs: torch.cuda.current_stream().wait_event(e)
        e1.record(torch.cuda.current_stream()); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / (10 * len(handles)))
    res[name] = [round(t, 3) for t in ts]
print(json.dumps(res))
