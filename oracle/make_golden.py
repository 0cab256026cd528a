"""Generates tests/golden/*.npz.  Run in the build container only:

    python oracle/make_golden.py            # tiny config + kNN + preprocess
    python oracle/make_golden.py --full     # also full ViT-L/14 on 2 images (~1 min)

The reference itself cannot run here (Rust workspace, un-vendored crates, model
is a network download: DESIGN.md §3), so the ViT vectors are produced by the
locally installed `transformers` CLIPVisionModelWithProjection — the PyTorch
class the reference's ONNX graph was exported from — constructed from a config
(no hub access) and loaded with the build's seeded weights.  They pin
oracle/vit_numpy.py (the restatement) and, through it, the HIP path.
kNN vectors come from oracle/oracle.c's fp32 order and are cross-checked against
its fp64 distances; a vector is only recorded where the k / k+1 gap is wide
enough that fp64 and fp32 agree on the ordering.
"""
from __future__ import annotations

import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from image_search_amd import synth  # noqa: E402
from oracle import vit_numpy  # noqa: E402
from oracle.binding import load_oracle, orc_knn, orc_cosine_dist_f64  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hf_forward(cfg, weights, pixels):
    import torch
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    hc = CLIPVisionConfig(hidden_size=cfg.hidden, intermediate_size=cfg.ff, num_hidden_layers=cfg.layers,
                          num_attention_heads=cfg.heads, patch_size=cfg.patch, image_size=cfg.image,
                          projection_dim=cfg.proj, hidden_act="quick_gelu", layer_norm_eps=cfg.eps)
    hc._attn_implementation = "eager"
    m = CLIPVisionModelWithProjection(hc).eval()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in weights.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "position_ids" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    with torch.no_grad():
        out = m(pixel_values=torch.from_numpy(pixels))
    return out.image_embeds.numpy(), out.last_hidden_state.numpy()


def vit_golden(cfg, name, n_img, seed):
    w = synth.vit_weights(cfg, seed)
    u8 = synth.images_u8(seed + 100, n_img, cfg.image)
    px = synth.preprocess_rgb8(u8)
    emb_hf, _ = hf_forward(cfg, w, px)
    emb_np = vit_numpy.vit_forward(w, cfg, px, np.float32)
    emb_64 = vit_numpy.vit_forward(w, cfg, px, np.float64)
    rms = float(np.sqrt((emb_64 ** 2).mean()))
    print(f"[{name}] rms={rms:.4f}  |hf-f64|max={np.abs(emb_hf - emb_64).max():.3e}  "
          f"|np32-f64|max={np.abs(emb_np - emb_64).max():.3e}  |hf-np32|max={np.abs(emb_hf - emb_np).max():.3e}")
    np.savez_compressed(os.path.join(GOLD, f"vit_{name}.npz"), seed=seed, n_img=n_img,
                        image_seed=seed + 100, embeds_hf_f32=emb_hf.astype(np.float32),
                        embeds_f64=emb_64.astype(np.float64))


def hf_text_forward(cfg, weights, ids):
    import torch
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    hc = CLIPTextConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.ff,
                        num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                        max_position_embeddings=cfg.positions, projection_dim=cfg.proj, hidden_act="quick_gelu",
                        layer_norm_eps=cfg.eps, eos_token_id=2, bos_token_id=0, pad_token_id=1)  # eos 2: the argmax rule
    hc._attn_implementation = "eager"
    m = CLIPTextModelWithProjection(hc).eval()
    sd = {k: torch.from_numpy(np.array(v)) for k, v in weights.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if "position_ids" not in k]
    assert not missing and not unexpected, (missing, unexpected)
    with torch.no_grad():
        out = m(input_ids=torch.from_numpy(ids.astype(np.int64)))
    return out.text_embeds.numpy()


def text_golden(cfg, name, n_seq, seed):
    """The text tower (server/src/clip.rs:19-23): transformers'
    CLIPTextModelWithProjection on seeded weights pins oracle/vit_numpy.py:text_forward."""
    w = synth.vit_weights(cfg, seed)
    ids = synth.token_ids(cfg, seed + 7, n_seq)
    emb_hf = hf_text_forward(cfg, w, ids)
    emb_np = vit_numpy.text_forward(w, cfg, ids, np.float32)
    emb_64 = vit_numpy.text_forward(w, cfg, ids, np.float64)
    rms = float(np.sqrt((emb_64 ** 2).mean()))
    print(f"[text {name}] rms={rms:.4f}  |hf-f64|max={np.abs(emb_hf - emb_64).max():.3e}  "
          f"|np32-f64|max={np.abs(emb_np - emb_64).max():.3e}  |hf-np32|max={np.abs(emb_hf - emb_np).max():.3e}")
    np.savez_compressed(os.path.join(GOLD, f"text_{name}.npz"), seed=seed, n_seq=n_seq, ids_seed=seed + 7,
                        embeds_hf_f32=emb_hf.astype(np.float32), embeds_f64=emb_64.astype(np.float64))


def knn_golden():
    lib = load_oracle()
    cases = {}
    for tag, n, seed in (("n1k", 1000, 11), ("n100k", 100_000, 12)):
        rows = synth.corpus_rows(seed, 0, n, 768)
        qs = synth.corpus_rows(seed + 1000, 0, 4, 768)
        for k in (1, 10, 1000):
            kk = k
            idx_all, dist_all, gap_all = [], [], []
            for q in qs:
                idx, dist = orc_knn(lib, q, rows, kk, 0)
                d64 = orc_cosine_dist_f64(lib, q, rows)
                order64 = np.lexsort((np.arange(n), d64))[:min(kk, n) + 1]
                ok = np.array_equal(order64[:min(kk, n)], idx[:min(kk, n)].astype(np.int64))
                gaps = np.diff(d64[order64])
                idx_all.append(idx); dist_all.append(dist)
                gap_all.append(gaps.min() if ok else -1.0)
            cases[f"{tag}_k{k}_idx"] = np.stack(idx_all)
            cases[f"{tag}_k{k}_dist"] = np.stack(dist_all)
            cases[f"{tag}_k{k}_mingap64"] = np.array(gap_all)
            print(f"[knn {tag} k={k}] fp64 order agrees: {[g >= 0 for g in gap_all]}  min gap {min(gap_all):.3e}")
        cases[f"{tag}_seed"] = np.array([seed, seed + 1000, n])
    np.savez_compressed(os.path.join(GOLD, "knn.npz"), **cases)


def misc_golden():
    # the reference's one numeric known answer: server/src/search.rs:157-160
    np.savez(os.path.join(GOLD, "average_slices.npz"),
             a=np.array([1.0, 2.0, 4.0, 4.0, 10.0], np.float32),
             b=np.array([1.0, 1.0, 2.0, 4.0, 0.0], np.float32),
             expect=np.array([1.0, 1.5, 3.0, 4.0, 5.0], np.float32))
    # generator known answers (platform independence of synth.gen_f32)
    np.savez(os.path.join(GOLD, "gen.npz"), s7=synth.gen_f32(7, 0, 64), s7_off=synth.gen_f32(7, 1 << 40, 64, 0.02))


def resize_golden():
    """image_prepare_resnet's resize (server/src/clip.rs:154; image-0.25.8 CatmullRom).  Nothing in
    the reference pins a resized pixel, so two things are stored per case: the oracle's own output
    (regression pin) and an INDEPENDENT implementation of the same filter -- torch's antialiased
    bicubic (A = -0.5 Catmull-Rom, support scaled by the ratio, normalised weights; the
    Pillow-compatible resampler) -- which must agree within one grey level."""
    import torch
    import torch.nn.functional as F
    from oracle.binding import load_oracle, orc_resize_catmullrom
    lib = load_oracle()
    out = {}
    for name, seed, h, w in (("down", 3, 300, 401), ("up", 4, 37, 53), ("mixed", 5, 1000, 61)):
        img = synth.photo_u8(seed, h, w)
        t = torch.from_numpy(img).permute(2, 0, 1)[None].float()
        ind = F.interpolate(t, size=(224, 224), mode="bicubic", antialias=True, align_corners=False)
        out[f"{name}_seed"], out[f"{name}_hw"] = seed, np.array([h, w])
        out[f"{name}_oracle"] = orc_resize_catmullrom(lib, img, 224, 224)
        aa = ind[0].permute(1, 2, 0).clamp(0, 255).round().numpy().astype(np.uint8)
        diff = np.flatnonzero(aa.ravel() != out[f"{name}_oracle"].ravel())  # stored sparsely: a few pixels, +-1
        out[f"{name}_torch_aa_diff_idx"], out[f"{name}_torch_aa_diff_val"] = diff, aa.ravel()[diff]
    np.savez_compressed(os.path.join(GOLD, "resize.npz"), **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    misc_golden()
    resize_golden()
    knn_golden()
    vit_golden(synth.VitConfig.tiny(), "tiny", 3, 1)
    text_golden(synth.TextConfig.tiny(), "tiny", 4, 2)
    if a.full:
        vit_golden(synth.VitConfig.vit_l14(), "l14", 2, 0)
        text_golden(synth.TextConfig.clip_l14(), "l14", 3, 3)
