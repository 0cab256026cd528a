"""Dev harness (GPU): ViT parity vs the numpy oracle / golden vectors. Not a test."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth
from image_search_amd.clip import Model, PRECISION_F32, PRECISION_BF16
from oracle import vit_numpy

def run(cfg, name, seed, n_img, full_gold=None):
    w = synth.vit_weights(cfg, seed)
    path = os.path.join(tempfile.gettempdir(), f"w_{name}.safetensors")
    synth.save_safetensors(w, path, {"num_attention_heads": cfg.heads})
    px = synth.preprocess_rgb8(synth.images_u8(seed + 100, n_img, cfg.image))
    if full_gold is not None:
        ref = full_gold
    else:
        ref = vit_numpy.vit_forward(w, cfg, px, np.float64)
    rms = np.sqrt((ref ** 2).mean())
    for prec, pn in ((PRECISION_F32, "f32"), (PRECISION_BF16, "bf16")):
        t0 = time.time(); m = Model.from_file(path, 0, prec); t1 = time.time()
        out = m.forward(px); t2 = time.time()
        err = np.abs(out - ref)
        print(f"[{name} {pn}] load {t1-t0:.2f}s fwd {t2-t1:.3f}s  max|err| {err.max():.3e}  rel-to-rms {err.max()/rms:.3e}  "
              f"allclose(1e-4) {np.allclose(out, ref, rtol=1e-4, atol=1e-4*rms)}  nan {np.isnan(out).any()}")
        m.close()

run(synth.VitConfig.tiny(), "tiny", 1, 3)
if os.environ.get("FULL", "1") == "1":
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "vit_l14.npz"))
    run(synth.VitConfig.vit_l14(), "l14", 0, 2, g["embeds_f64"])
