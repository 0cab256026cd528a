"""Dev harness (GPU): the 24-bit residual stream (option x24) against the fp32-residual bf16 tower and the transformers golden: error and time.  profiles/r04_x24_residual.json."""
import os, sys, tempfile, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth
from image_search_amd.clip import Model, PRECISION_BF16
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "vit_l14.npz"))
cfg = synth.VitConfig.vit_l14()
w = synth.vit_weights(cfg, int(g["seed"]))
path = os.path.join(tempfile.gettempdir(), f"x24_{os.getpid()}.safetensors")
synth.save_safetensors(w, path, {"num_attention_heads": cfg.heads})
px = synth.preprocess_rgb8(synth.images_u8(int(g["image_seed"]), int(g["n_img"]), cfg.image))
ref = g["embeds_f64"]; rms = float(np.sqrt((ref ** 2).mean()))
m = Model.from_file(path, 0, PRECISION_BF16); os.unlink(path)
m.set_option("ln_fold", 0)   # x24 is the residual format of the LayerNorm-kernel form of the layer loop (round 5: ln_fold = 1 by default has its own planes)
out = {}
for v in (0, 1):
    m.set_option("x24", v)
    o = m.forward(px)
    out[f"x24={v}"] = {"max_err_over_rms": float(np.abs(o - ref).max() / rms), "rms_err_over_rms": float(np.sqrt(((o - ref) ** 2).mean()) / rms)}
# a larger sample: 64 seeded images, fp32 tower as the reference
from image_search_amd.clip import PRECISION_F32
px64 = synth.preprocess_rgb8(synth.images_u8(7, 64, cfg.image))
path = os.path.join(tempfile.gettempdir(), f"x24b_{os.getpid()}.safetensors")
synth.save_safetensors(w, path, {"num_attention_heads": cfg.heads})
m32 = Model.from_file(path, 0, PRECISION_F32); os.unlink(path)
r32 = m32.forward(px64); rms = float(np.sqrt((r32.astype(np.float64) ** 2).mean()))
for v in (0, 1):
    m.set_option("x24", v)
    o = m.forward(px64)
    out[f"64 images vs fp32 tower, x24={v}"] = {"max_err_over_rms": float(np.abs(o - r32).max() / rms), "rms_err_over_rms": float(np.sqrt(((o - r32) ** 2).mean()) / rms)}
print(json.dumps(out, indent=1))
