"""What the bf16 throughput path costs downstream.  Run on the GPU box:

    python tools/bf16_acceptance.py [--images 4096] [--queries 1000] > profiles/r06_bf16_acceptance.json

(i)  bf16 vs fp32 embeddings of the same images, both through the HIP library, seeded ViT-L/14 weights:
     as generated (HF init scales) and with planted outlier channels (x50 on four LayerNorm gains, the
     shape trained CLIP towers have).  error = max |bf16 - fp32| / rms(fp32), and the cosine distance
     between the two embeddings of one image against the distance to its nearest OTHER image.
     (profiles/r02_bf16_acceptance.json also holds a third mode, a bf16 RESIDUAL stream, built for that run and then
     removed: 3x the error for no speed — 40.99 vs 40.65 ms per 256 images; the LayerNorms it was meant to relieve
     already overlap the other half-chunk's GEMMs.)
(ii) retrieval: the same images indexed twice (fp32 embeddings, bf16 embeddings); held-out query images
     embedded in the index's own precision; top-1 / top-10 / top-1000 id agreement between the two systems.
(iii) round 6 — where the LayerNorm-free loop ("ln_fold", the default) stops being safe.  It rounds the UN-normalised
     residual row to bf16, so what it is sensitive to is a row's mean against its deviation, which the studies above never
     plant (gamma outliers are absorbed into W' = W diag(gamma)).  A constant c on every channel of pre_layrnorm.bias puts
     a common offset of c sigma on the whole residual stream (sigma = 1 behind the pre-LayerNorm); every later LayerNorm
     removes it, so the function — and the fp32 tower — does not move.  c in {0, 1, 4, 16, 64}; and one channel of
     pre_layrnorm.bias at +100 (a massive-activation channel: changes the function, not the row mean).  ln_fold 1 against
     0, with the library's own counter (mi_clip_ln_fold_stats: rows with mean^2 > 16 var) beside each — once with
     the weights as read (MI_CLIP_LN_CENTER=0: the sensitivity) and once as the library loads them by default (the common mode
     of everything written to the stream removed at load: the cure).
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth  # noqa: E402
from image_search_amd.clip import PRECISION_BF16, PRECISION_BF16_SPLIT, PRECISION_F32, Model  # noqa: E402
from image_search_amd.search import EmbeddingTable  # noqa: E402


LAST_FOLD_STATS = None


def embed(path, px, prec, center=None, **opts):
    global LAST_FOLD_STATS
    if center is not None:
        os.environ["MI_CLIP_LN_CENTER"] = str(center)   # read at load: it shapes the out_proj / fc2 weights
    try:
        m = Model.from_file(path, 0, prec)
    finally:
        os.environ.pop("MI_CLIP_LN_CENTER", None)
    for k, v in opts.items():
        m.set_option(k, v)
    out = np.concatenate([m.forward(px[i:i + 256]) for i in range(0, len(px), 256)])
    LAST_FOLD_STATS = m.ln_fold_stats()
    m.close()
    return out


def cos_dist(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return 1.0 - (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1))


def study(name, weights, cfg, px_index, px_query, ks, split=True, e32_ref=None):
    path = os.path.join(tempfile.gettempdir(), f"mi355clip_acc_{os.getpid()}.safetensors")
    synth.save_safetensors(weights, path, {"num_attention_heads": cfg.heads})
    print("study:", name, file=sys.stderr, flush=True)
    try:
        e32 = embed(path, px_index, PRECISION_F32)
        q32 = embed(path, px_query, PRECISION_F32)
        i1 = embed(path, px_index, PRECISION_BF16, ln_fold=1)
        q1 = embed(path, px_query, PRECISION_BF16, ln_fold=1)
        counter = LAST_FOLD_STATS   # of the query batch: (rows more than 4 sigma off zero, rows looked at)
        i1r = embed(path, px_index, PRECISION_BF16, center=0, ln_fold=1)
        q1r = embed(path, px_query, PRECISION_BF16, center=0, ln_fold=1)
        counter_r = LAST_FOLD_STATS
        out = [compare(name, "MI_PRECISION_BF16 (default: ln_fold = 1, no LayerNorm kernels in the layer loop; residual writers centred at load)", e32, q32, i1, q1, ks),
               compare(name, "MI_PRECISION_BF16 ln_fold = 1, weights as read (MI_CLIP_LN_CENTER=0)", e32, q32, i1r, q1r, ks),
               compare(name, "MI_PRECISION_BF16 ln_fold = 0 (LayerNorm kernels, the tower of rounds 1-4)", e32, q32,
                       embed(path, px_index, PRECISION_BF16, ln_fold=0), embed(path, px_query, PRECISION_BF16, ln_fold=0), ks)]
        out[0]["ln_fold_stats_of_the_query_pass"] = {"rows_over_4_sigma_off_zero": counter[0], "rows_looked_at": counter[1]}
        out[1]["ln_fold_stats_of_the_query_pass"] = {"rows_over_4_sigma_off_zero": counter_r[0], "rows_looked_at": counter_r[1]}
        if e32_ref is not None:   # a study whose weights leave the function alone: how far did the fp32 tower itself move?
            rms = float(np.sqrt((e32_ref.astype(np.float64) ** 2).mean()))
            out[0]["fp32_embedding_moved_by_the_planted_offset_over_rms"] = float(np.abs(e32 - e32_ref).max() / rms)
        if split:
            out.append(compare(name, "MI_PRECISION_BF16_SPLIT", e32, q32, embed(path, px_index, PRECISION_BF16_SPLIT),
                               embed(path, px_query, PRECISION_BF16_SPLIT), ks))
    finally:
        os.unlink(path)
    return out, e32


def with_pre_ln_bias(weights, add, channels=None):
    """pre_layrnorm.bias + add on `channels` (None: every channel — a common offset on the whole residual stream)."""
    out = dict(weights)
    name = [k for k in weights if k.endswith("pre_layrnorm.bias")][0]
    b = weights[name].copy()
    if channels is None:
        b += np.float32(add)
    else:
        b[list(channels)] += np.float32(add)
    out[name] = b
    return out


def compare(name, mode, e32, q32, e16, q16, ks):
    px_index, px_query = e32, q32
    rms = float(np.sqrt((e32.astype(np.float64) ** 2).mean()))
    err = np.abs(e16 - e32).max(-1) / rms
    own = cos_dist(e16, e32)                                   # bf16 vs fp32 embedding of the SAME image
    t32, t16 = EmbeddingTable(768, 0), EmbeddingTable(768, 0)
    t32.insert(e32); t16.insert(e16)
    kmax = max(ks)
    i32, d32 = t32.knn(q32, kmax)
    i16, _ = t16.knn(q16, kmax)
    nn_other = d32[:, 0]                                       # query -> nearest indexed image, fp32 system
    res = {"weights": name, "mode": mode, "indexed_images": int(len(px_index)), "queries": int(len(px_query)),
           "embedding_rms": rms,
           "bf16_max_abs_err_over_rms": {"max": float(err.max()), "median": float(np.median(err))},
           "cosine_distance_bf16_vs_fp32_same_image": {"max": float(own.max()), "median": float(np.median(own))},
           "cosine_distance_query_to_nearest_indexed_image_fp32": {"min": float(nn_other.min()), "median": float(np.median(nn_other))},
           "agreement": {}}
    for k in ks:
        inter = [len(set(i32[u, :k].tolist()) & set(i16[u, :k].tolist())) / k for u in range(len(q32))]
        res["agreement"][f"top{k}_set_overlap_mean"] = float(np.mean(inter))
        res["agreement"][f"top{k}_set_overlap_min"] = float(np.min(inter))
    res["agreement"]["top1_same_id"] = float((i32[:, 0] == i16[:, 0]).mean())
    res["agreement"]["top10_same_order"] = float(np.mean([(i32[u, :10] == i16[u, :10]).all() for u in range(len(q32))]))
    t32.close(); t16.close()
    return res


def timing(weights, cfg, px):
    import time
    path = os.path.join(tempfile.gettempdir(), f"mi355clip_acc_{os.getpid()}.safetensors")
    synth.save_safetensors(weights, path, {"num_attention_heads": cfg.heads})
    out = {}
    try:
        for name, prec in (("MI_PRECISION_BF16", PRECISION_BF16), ("MI_PRECISION_BF16_SPLIT", PRECISION_BF16_SPLIT)):
            import torch
            m = Model.from_file(path, 0, prec)
            d_in = torch.from_numpy(px).cuda()
            d_out = torch.empty((len(px), 768), dtype=torch.float32, device="cuda")
            st = torch.cuda.Stream()
            for _ in range(3):
                m.forward_device(d_in.data_ptr(), len(px), d_out.data_ptr(), st.cuda_stream)
            st.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st)
            for _ in range(10):
                m.forward_device(d_in.data_ptr(), len(px), d_out.data_ptr(), st.cuda_stream)
            b.record(st)
            st.synchronize()
            out[name] = round(a.elapsed_time(b) / 10, 3)   # inputs resident, HIP events on the launch stream
            m.close()
    finally:
        os.unlink(path)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=4096)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--offsets", default="1,4,16,64", help="common offsets c (in sigma) of study (iii); empty = skip")
    args = ap.parse_args()
    cfg = synth.VitConfig.vit_l14()
    w = synth.vit_weights(cfg, 0)
    px_index = synth.preprocess_rgb8(synth.scenes_u8(5, args.images, cfg.image))
    px_query = synth.preprocess_rgb8(synth.scenes_u8(6, args.queries, cfg.image))
    ks = [k for k in (1, 10, 100, 1000) if k <= args.images]
    base, e32_base = study("as generated (HF init scales)", w, cfg, px_index, px_query, ks)
    studies = base
    studies += study("outlier channels planted, function preserved: x50 on 4 LayerNorm gains/biases, /50 on the matching q/k/v/fc1 input columns",
                     synth.plant_outlier_channels(w, compensate=True), cfg, px_index, px_query, ks)[0]
    studies += study("outlier channels planted, function changed: x50 on 4 LayerNorm gains/biases only (attention logits grow ~10x)",
                     synth.plant_outlier_channels(w), cfg, px_index, px_query, ks)[0]
    for c in [float(x) for x in args.offsets.split(",") if x]:
        studies += study(f"common offset of {c:g} sigma on the residual stream (pre_layrnorm.bias + {c:g} on every channel; function unchanged)",
                         with_pre_ln_bias(w, c), cfg, px_index, px_query, ks, split=False, e32_ref=e32_base)[0]
    if args.offsets:
        studies += study("one residual channel at +100 sigma (pre_layrnorm.bias[123] + 100; function changed, row mean 0.1 sigma)",
                         with_pre_ln_bias(w, 100.0, [123]), cfg, px_index, px_query, ks, split=False)[0]
    out = {"what": "bf16 tower vs fp32 tower (both HIP), seeded ViT-L/14, structured synthetic images (synth.scenes_u8)",
           "studies": studies,
           "tower_ms_per_256_images": timing(w, cfg, px_index[:256])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
