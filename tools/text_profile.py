"""One bf16 text query, 200 times: the workload for `rocprofv3 --kernel-trace --stats -- python3 tools/text_profile.py`
(per-kernel durations of forward_text_one; MI_CLIP_TEXT_FAST=0 profiles the batched kernels on the same query)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth
from image_search_amd.clip import PRECISION_BF16, TextModel

cfg = synth.TextConfig.clip_l14()
path = "/tmp/text_l14.safetensors"
if not os.path.exists(path):
    synth.save_safetensors(synth.vit_weights(cfg, 3), path, {"num_attention_heads": cfg.heads})
m = TextModel.from_file(path, 0, PRECISION_BF16)
ids = synth.token_ids(cfg, 11, 1)
for _ in range(5):
    m.embed(ids)
t0 = time.perf_counter()
for _ in range(200):
    m.embed(ids)
print(f"{(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per query")
m.close()
