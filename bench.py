"""bench.py — the reference's headline workload on MI355X.

BASELINE.json metric: "images/sec embedded (ViT-L/14 b=256) + queries/sec cosine
top-10 over 10M x 768".  One step = one pass of the hot path over one batch
(BASELINE config[3]): embed 256 synthetic 224x224x3 images with the bf16 ViT-L/14,
then answer one top-10 cosine query over this GPU's 10M x 768 fp32 shard (the
reference serves one query per request, server/src/search.rs:20-102).  Inputs are
resident in HBM when the timed region starts.  With N > 1 ranks (one process per
GPU) every rank embeds its own batch (replicas, no collective) and owns its own
10M-row shard of an N x 10M table; the per-shard top-k are all-gathered over RCCL
and merged on every rank — weak scaling.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `value` is whole-job images/s over the timed region
(ViT + query); the per-phase rates (HIP events on the launch stream) are under
"vit" and "knn", each with the roofline of its dominant kernel.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VIT_FLOP_PER_IMAGE = 2 * 81_012_768_768  # SURVEY.md §2.1 / BASELINE.md §2: algorithmic MACs x 2
# Executed by this implementation: behind the last layer's attention only the CLS row is live, so that
# layer's out_proj / MLP / attention rows for the other 256 tokens are not computed (bit-identical output,
# tests/test_vit_gpu.py::test_cls_only_last_layer_is_bit_identical_to_the_full_one):
#   2*256*(2*1024*1024 + 2*1024*4096) + 4*257*64*16*256 FLOP per image less (q_proj and out_proj, the MLP,
#   the attention rows of the 256 non-CLS tokens).
VIT_FLOP_SKIPPED_PER_IMAGE = 2 * 256 * (2 * 1024 * 1024 + 2 * 1024 * 4096) + 4 * 257 * 64 * 16 * 256
PEAK_BF16_TFLOPS = 2500.0                # dense bf16 MFMA, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0                    # HBM3E spec, same table


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(weights, cfg, orc_threads_hint):
    """The oracle ("port") timed on this host's cores: a bounded sample of the same
    workload.  Only this function and the parity tests touch oracle/."""
    from image_search_amd import synth
    from oracle import vit_numpy
    from oracle.binding import load_oracle, orc_gen_f32, orc_knn

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # ViT: numpy fp32 restatement (BLAS threads = all cores), 4 images
    px = synth.preprocess_rgb8(synth.images_u8(100, 4, cfg.image))
    vit_numpy.vit_forward(weights, cfg, px[:1], np.float32)  # warm BLAS
    t0 = time.perf_counter()
    vit_numpy.vit_forward(weights, cfg, px, np.float32)
    t_vit = time.perf_counter() - t0
    # kNN: C restatement (OpenMP), 1M x 768 rows, 5 queries
    orc = load_oracle()
    n = 1_000_000
    rows = orc_gen_f32(orc, 0, 0, n * 768, 1.0).reshape(n, 768)
    qs = synth.corpus_rows(1, 0, 5)
    orc_knn(orc, qs[0], rows, 10)
    t0 = time.perf_counter()
    for q in qs:
        orc_knn(orc, q, rows, 10)
    t_knn = (time.perf_counter() - t0) / len(qs)
    return {
        "value": round(4 / t_vit, 3), "unit": "images/s", "cores": cores, "kind": "port",
        "sample": "oracle/vit_numpy.py fp32 ViT-L/14 on 4 images (numpy+BLAS, all cores); "
                  "kNN below: oracle/oracle.c orc_knn (OpenMP) top-10 over 1M x 768, mean of 5 queries",
        "knn": {"value": round(1.0 / t_knn, 2), "unit": "queries/s over 1M rows",
                "equiv_10M": round(0.1 / t_knn, 3), "threads": int(orc.orc_threads())},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--rows", type=int, default=10_000_000, help="table rows per GPU")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo for rehearsals)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from image_search_amd import synth
    from image_search_amd.clip import PRECISION_BF16, Model
    from image_search_amd.search import EmbeddingTable, merge_candidates

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    local = local % max(1, torch.cuda.device_count())  # rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    xdev = "cuda" if args.backend == "nccl" else "cpu"  # where the all-gather buffers live

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- untimed setup -------------------------------------------------------------
    cfg = synth.VitConfig.vit_l14()
    wpath = os.path.join(tempfile.gettempdir(), "mi355clip_bench_vitl14_seed0.safetensors")
    weights = None
    if rank == 0:
        t0 = time.time()
        weights = synth.vit_weights(cfg, 0)
        synth.save_safetensors(weights, wpath + ".tmp", {"num_attention_heads": cfg.heads})
        os.replace(wpath + ".tmp", wpath)
        log(f"[bench] seeded ViT-L/14 weights written in {time.time() - t0:.1f}s")
    barrier()
    model = Model.from_file(wpath, local, PRECISION_BF16)
    px = synth.preprocess_rgb8(synth.images_u8(1000 + rank, args.batch, cfg.image))
    d_img = torch.from_numpy(px).cuda()
    d_emb = torch.empty((args.batch, cfg.proj), dtype=torch.float32, device="cuda")

    table = EmbeddingTable(768, local, base=rank * args.rows)
    table.reserve(args.rows)
    table.insert_synthetic(0, rank * args.rows, args.rows)
    n_q = 64
    d_q = torch.from_numpy(synth.corpus_rows(1, 0, n_q)).cuda()
    d_idx = torch.empty((1, args.k), dtype=torch.int64, device="cuda")
    d_dist = torch.empty((1, args.k), dtype=torch.float32, device="cuda")
    g_idx = torch.empty((world, args.k), dtype=torch.int64, device=xdev)
    g_dist = torch.empty((world, args.k), dtype=torch.float32, device=xdev)

    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    s = stream.cuda_stream
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def step(i, marks=None):
        if marks is not None:
            marks[0].record(stream)
        model.forward_device(d_img.data_ptr(), args.batch, d_emb.data_ptr(), s)
        if marks is not None:
            marks[1].record(stream)
        table.knn_device(d_q[i % n_q].data_ptr(), 1, args.k, d_idx.data_ptr(), d_dist.data_ptr(), s)
        if marks is not None:
            marks[2].record(stream)
        if world > 1:  # the one exchange step: 12*k bytes per rank and query
            if xdev == "cpu":
                stream.synchronize()
            dist.all_gather_into_tensor(g_idx, d_idx.to(xdev))
            dist.all_gather_into_tensor(g_dist, d_dist.to(xdev))
            gi = g_idx.cpu().numpy().view(np.uint64)
            gd = g_dist.cpu().numpy()
            return merge_candidates(gi, gd, args.k)
        return None

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    marks = [[ev(), ev(), ev()] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, marks[i])
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_vit = float(np.mean([m[0].elapsed_time(m[1]) for m in marks]))
    ms_knn = float(np.mean([m[1].elapsed_time(m[2]) for m in marks]))

    if rank == 0:
        imgs = world * args.batch * args.steps
        vit_tflops = args.batch * VIT_FLOP_PER_IMAGE / (ms_vit * 1e-3) / 1e12
        skipped = 0 if os.environ.get("MI_CLIP_FULL_LAST", "0") not in ("", "0") else VIT_FLOP_SKIPPED_PER_IMAGE
        knn_gbs = args.rows * 768 * 4 / (ms_knn * 1e-3) / 1e9
        pmc = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
                pmc = json.load(f)
        except OSError:
            pass
        traffic_ok = bool(pmc) and pmc.get("rows") == args.rows and pmc.get("batch") == args.batch
        out = {
            "metric": "images/sec embedded (ViT-L/14 b=256) + queries/sec cosine top-10 over 10M x 768",
            "value": round(imgs / elapsed, 2),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"embed batch={args.batch} 224x224x3 (bf16 ViT-L/14, random-init seeded weights) "
                                   f"+ cosine top-{args.k} query over {args.rows} x 768 fp32 rows per GPU",
                       "batch": args.batch, "rows_per_gpu": args.rows, "k": args.k, "queries_per_step": 1,
                       "sharding": "ViT replicas; table row-sharded, all-gather of per-shard top-k"},
            "vit": {"images_per_sec": round(world * args.batch / (ms_vit * 1e-3), 1), "ms_per_batch": round(ms_vit, 3)},
            "knn": {"queries_per_sec": round(1e3 / ms_knn, 2), "ms_per_query": round(ms_knn, 4),
                    "rows_scanned_per_sec": round(world * args.rows / (ms_knn * 1e-3), 0), "dtype": "f32"},
            "roofline": {"bound": "mfma", "kernel": "ViT-L/14 forward (gemm_bf16_pp_kernel x 96 + attention + LayerNorm, two half-chunk streams)",
                         "achieved": round(vit_tflops, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(vit_tflops / PEAK_BF16_TFLOPS, 4),
                         "traffic": pmc.get("vit_hbm_bytes") if traffic_ok else None,
                         "algorithmic_gflop_per_image": round(VIT_FLOP_PER_IMAGE / 1e9, 2),
                         "executed_gflop_per_image": round((VIT_FLOP_PER_IMAGE - skipped) / 1e9, 2),
                         "frac_of_peak_on_executed_flops": round(vit_tflops * (1 - skipped / VIT_FLOP_PER_IMAGE) / PEAK_BF16_TFLOPS, 4),
                         "note": "achieved/frac use the algorithmic count (SURVEY.md 8d); the last layer runs on the CLS rows only (dead rows of the reference graph are not computed, output bit-identical), MI_CLIP_FULL_LAST=1 restores them"},
            "roofline_knn": {"bound": "hbm", "kernel": "knn_scan_kernel<12,WaveTopReg> (+2 merge launches)",
                             "achieved": round(knn_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": round(knn_gbs / PEAK_HBM_GBS, 4),
                             "traffic": pmc.get("knn_scan_hbm_bytes") if traffic_ok else None},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(weights, cfg, None)
        print(json.dumps(out), flush=True)

    model.close()
    table.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
