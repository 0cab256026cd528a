"""bench.py — the reference's headline workload on MI355X.

BASELINE.json metric: "images/sec embedded (ViT-L/14 b=256) + queries/sec cosine
top-10 over 10M x 768".  One step = one pass of the hot path over one batch
(BASELINE config[3], "fused on HIP streams"): upload a pinned-host batch of 256
synthetic 224x224x3 images, embed it with the bf16 ViT-L/14, append the 256 rows to
the table on the device, answer one top-10 cosine query over this GPU's 10M x 768
fp32 shard plus the appended rows (the reference serves one query per request,
server/src/search.rs:20-102) and read the k results back.  The H2D of the batch and
the D2H of the results are INSIDE the timed region; the upload runs
under the previous batch's tower, the scan is queued asynchronously but on the device it
runs between two towers (step = tower + scan; measured) (mi_pipeline_*).
The query runs as the two-stage EXACT search (a byte mirror of the rows prefilters, the fp32
rows decide: ids and distance bits of the single pass; --prefilter 1 = the bf16
mirror, --no-prefilter = one pass over the fp32 rows, which is also what `roofline_knn` times).  With N > 1 ranks (one process per
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
PEAK_F32_TFLOPS = 157.3                  # exact-f32 MFMA, same table


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(weights, cfg, orc_threads_hint):
    """The oracle ("port") timed on this host's cores: a bounded sample of the same
    workload.  Only this function and the parity tests touch oracle/."""
    from image_search_amd import synth
    from oracle import vit_numpy
    from oracle.binding import load_oracle, orc_gen_f32, orc_knn

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # ViT: numpy fp32 restatement (BLAS threads = all cores): b = 1 (latency) and b = 8 (BASELINE.md §3
    # planned b = 32; at < 1 image/s that alone would be > 40 s of a run that must finish in minutes)
    px = synth.preprocess_rgb8(synth.images_u8(100, 8, cfg.image))
    vit_numpy.vit_forward(weights, cfg, px[:1], np.float32)  # warm BLAS
    t0 = time.perf_counter()
    vit_numpy.vit_forward(weights, cfg, px[:1], np.float32)
    t_one = time.perf_counter() - t0
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
        "value": round(8 / t_vit, 3), "unit": "images/s", "cores": cores, "kind": "port",
        "sample": "oracle/vit_numpy.py fp32 ViT-L/14, ONE batch of 8 images (numpy+BLAS, all cores; b=1 latency beside it); "
                  "kNN below: oracle/oracle.c orc_knn (OpenMP) top-10 over 1M x 768, mean of 5 queries",
        "vit_b1_seconds": round(t_one, 3), "vit_b8_seconds": round(t_vit, 3),
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
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the BASELINE config 1 / 2 sub-lines")
    ap.add_argument("--serial", action="store_true", help="A/B: synchronise after every step (no cross-step overlap)")
    ap.add_argument("--no-prefilter", action="store_true", help="A/B: the query as ONE pass over the fp32 rows (no mirror)")
    ap.add_argument("--prefilter", type=int, default=2, choices=(1, 2), help="mirror of the two-stage exact search: 2 = bytes (default), 1 = bf16")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL; gloo for rehearsals)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from image_search_amd import synth
    from image_search_amd.clip import PRECISION_BF16, PRECISION_F32, Model
    from image_search_amd.search import EmbeddingTable, PinnedBuffer, Pipeline, merge_candidates

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
    total_steps = args.warmup + args.steps
    # the step's input: a batch in PINNED host memory (what the server's decode threads would fill),
    # two buffers so that batch i+1 uploads under the forward of batch i
    pins = [PinnedBuffer((args.batch, 3, cfg.image, cfg.image)) for _ in range(2)]
    for j, pb in enumerate(pins):
        pb.array[:] = synth.preprocess_rgb8(synth.images_u8(1000 + 2 * rank + j, args.batch, cfg.image))

    table = EmbeddingTable(768, local, base=rank * (args.rows + total_steps * args.batch))
    table.reserve(args.rows + total_steps * args.batch)  # the appended rows never reallocate the table
    table.insert_synthetic(0, rank * args.rows, args.rows)
    if not args.no_prefilter:
        table.set_option("prefilter", args.prefilter)  # the mirror is built by the first (warm-up) query and caught up by every later one
    n_q = 64
    queries = synth.corpus_rows(1, 0, n_q)
    pipe = Pipeline(model, table)
    g_idx = torch.empty((world, args.k), dtype=torch.int64, device=xdev)
    g_dist = torch.empty((world, args.k), dtype=torch.float32, device=xdev)
    merged = []

    def exchange(res):
        """the one exchange step of the sharded search: 12*k bytes per rank and query"""
        li = torch.from_numpy(res[0].view(np.int64)).to(xdev)
        ld = torch.from_numpy(res[1]).to(xdev)
        dist.all_gather_into_tensor(g_idx, li.reshape(1, -1))
        dist.all_gather_into_tensor(g_dist, ld.reshape(1, -1))
        merged.append(merge_candidates(g_idx.cpu().numpy().view(np.uint64), g_dist.cpu().numpy(), args.k))

    pending = []

    def step(i):
        # BASELINE config 4: H2D of the batch -> bf16 tower -> rows appended on the device -> top-k over
        # table + appended rows -> D2H of the k results; the scan of step i runs on the search stream
        # under the tower of step i+1 (mi_pipeline_*, include/mi355clip.h)
        pipe.ingest(pins[i & 1].array)
        pending.append(pipe.query(queries[i % n_q], args.k))
        if args.serial:
            pipe.sync()
        if world > 1:
            pipe.drain(0 if args.serial else 1)        # results of the previous query are on the host now
            while len(pending) > (0 if args.serial else 1):
                exchange(pending.pop(0))

    def finish():
        pipe.sync()
        if world > 1:
            while pending:
                exchange(pending.pop(0))
        pending.clear()

    for i in range(args.warmup):
        step(i)
    finish()
    pipe.stats(reset=True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    finish()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    n_f, ms_f, n_s, ms_s = pipe.stats()
    ms_vit = ms_f / max(n_f, 1)          # HIP events on the ingest stream around each forward, timed region only
    ms_knn_overlapped = ms_s / max(n_s, 1)

    extra = {}
    if rank == 0:
        # ---- the kNN kernel by itself (same table): HIP events on the launch stream --------------
        stream = torch.cuda.Stream()
        d_q = torch.from_numpy(queries).cuda()
        ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

        def time_knn(tbl, k, reps):
            d_i = torch.empty((1, k), dtype=torch.int64, device="cuda")
            d_d = torch.empty((1, k), dtype=torch.float32, device="cuda")
            for i in range(2):
                tbl.knn_device(d_q[i].data_ptr(), 1, k, d_i.data_ptr(), d_d.data_ptr(), stream.cuda_stream)
            stream.synchronize()
            a, b = ev(), ev()
            a.record(stream)
            for i in range(reps):
                tbl.knn_device(d_q[i % n_q].data_ptr(), 1, k, d_i.data_ptr(), d_d.data_ptr(), stream.cuda_stream)
            b.record(stream)
            stream.synchronize()
            return a.elapsed_time(b) / reps

        # the single pass over the fp32 rows (the roofline kernel), then what the step actually ran
        table.set_option("prefilter", 0)
        ms_knn = time_knn(table, args.k, 20)
        ms_knn_two, pref_cand, pref_fell_back = None, 0, False
        if not args.no_prefilter:
            table.set_option("prefilter", args.prefilter)
            ms_knn_two = time_knn(table, args.k, 20)
            pref_cand, pref_fell_back = table.prefilter_stats()
    pipe.close()

    if rank == 0 and world == 1 and not args.no_extra_configs:
        # BASELINE config 3 (1 M rows; k = 10 and the reference's k = 1000) and config 2 (fp32 b = 32)
        rows_total = len(table)
        t1m = EmbeddingTable(768, local)
        t1m.insert_synthetic(0, 0, 1_000_000)
        for k in (10, 1000):
            ms = time_knn(t1m, k, 50)
            extra[f"knn_1m_k{k}"] = {"config": f"cosine top-{k} over 1M x 768 f32, 1 query per pass", "ms_per_query": round(ms, 4),
                                     "queries_per_sec": round(1e3 / ms, 1), "GB_per_s": round(3.072 / ms * 1e3, 1),
                                     "frac_of_hbm_peak": round(3.072 / ms * 1e3 / PEAK_HBM_GBS, 4)}
        if not args.no_prefilter:
            t1m.set_option("prefilter", args.prefilter)
            for k in (10, 1000):
                ms = time_knn(t1m, k, 50)
                extra[f"knn_1m_k{k}"]["two_stage_ms_per_query"] = round(ms, 4)
        t1m.close()
        table.set_option("prefilter", 0)
        ms = time_knn(table, 1000, 10)
        extra["knn_10m_k1000"] = {"config": f"cosine top-1000 over {rows_total} x 768 f32 (the reference's K)", "ms_per_query": round(ms, 4),
                                  "GB_per_s": round(rows_total * 3072 / ms / 1e6, 1),
                                  "frac_of_hbm_peak": round(rows_total * 3072 / ms / 1e6 / PEAK_HBM_GBS, 4)}
        if not args.no_prefilter:
            table.set_option("prefilter", args.prefilter)
            extra["knn_10m_k1000"]["two_stage_ms_per_query"] = round(time_knn(table, 1000, 10), 4)
            table.set_option("prefilter", 0)
        m32 = Model.from_file(wpath, local, PRECISION_F32)
        d_img = torch.from_numpy(np.ascontiguousarray(pins[0].array[:32])).cuda()
        d_emb = torch.empty((32, cfg.proj), dtype=torch.float32, device="cuda")
        for _ in range(2):
            m32.forward_device(d_img.data_ptr(), 32, d_emb.data_ptr(), stream.cuda_stream)
        stream.synchronize()
        a, b = ev(), ev()
        a.record(stream)
        for _ in range(5):
            m32.forward_device(d_img.data_ptr(), 32, d_emb.data_ptr(), stream.cuda_stream)
        b.record(stream)
        stream.synchronize()
        ms = a.elapsed_time(b) / 5
        tf = 32 * VIT_FLOP_PER_IMAGE / (ms * 1e-3) / 1e12
        extra["vit_fp32_b32"] = {"config": "ViT-L/14 image encoder, batch=32 fp32 (exact-f32 MFMA, the parity path), inputs resident",
                                 "ms_per_batch": round(ms, 3), "images_per_sec": round(32e3 / ms, 1), "TFLOP_per_s": round(tf, 1),
                                 "frac_of_f32_mfma_peak": round(tf / PEAK_F32_TFLOPS, 4)}
        m32.close()

    if rank == 0:
        imgs = world * args.batch * args.steps
        executed = VIT_FLOP_PER_IMAGE - VIT_FLOP_SKIPPED_PER_IMAGE
        tf_exec = args.batch * executed / (ms_vit * 1e-3) / 1e12
        tf_alg = args.batch * VIT_FLOP_PER_IMAGE / (ms_vit * 1e-3) / 1e12
        knn_gbs = len(table) * 768 * 4 / (ms_knn * 1e-3) / 1e9
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
            "config": {"workload": f"BASELINE config 4: per step, H2D of a pinned batch of {args.batch} 224x224x3 f32 images -> bf16 ViT-L/14 "
                                   f"(random-init seeded weights) -> {args.batch} rows appended to the table on the device -> cosine "
                                   f"top-{args.k} query over {args.rows}+ x 768 fp32 rows per GPU -> D2H of the k results; fused on HIP "
                                   "streams (the upload runs under the previous batch's tower; the scan is asynchronous to the host and runs between towers)"
                                   + ("; query = ONE pass over the fp32 rows" if args.no_prefilter else
                                      ("; query = two-stage EXACT search: a byte mirror of the rows with a per-row scale (+25 % HBM) prefilters, the rows a "
                                       "rigorous per-row error bound cannot exclude" if args.prefilter == 2 else
                                       "; query = two-stage EXACT search: a bf16 mirror of the rows (+50 % HBM) prefilters, the rows within a "
                                       "data-independent error bound of the k-th") + " are re-evaluated from the fp32 rows: ids and distance bits of the single pass")
                                   + ("; --serial: no overlap" if args.serial else ""),
                       "batch": args.batch, "rows_per_gpu": args.rows, "k": args.k, "queries_per_step": 1,
                       "transfers_in_timed_region": True,
                       "sharding": "ViT replicas; table row-sharded, all-gather of per-shard top-k"},
            "vit": {"images_per_sec": round(world * args.batch / (ms_vit * 1e-3), 1), "ms_per_batch": round(ms_vit, 3),
                    "note": "HIP events on the ingest stream around each forward of the timed region (one forward at a time on the ingest stream)"},
            "knn": {"queries_per_sec": round(1e3 / (ms_knn_two or ms_knn), 2), "ms_per_query": round(ms_knn_two or ms_knn, 4),
                    "ms_per_query_in_the_pipeline": round(ms_knn_overlapped, 4),
                    "ms_per_query_single_pass": round(ms_knn, 4),
                    "mode": "single pass over the fp32 rows" if args.no_prefilter else
                            f"two-stage exact ({'byte' if args.prefilter == 2 else 'bf16'} mirror prefilter + fp32 re-evaluation)",
                    "rows_searched_per_sec": round(world * len(table) / ((ms_knn_two or ms_knn) * 1e-3), 0), "dtype": "f32"},
            "roofline": {"bound": "mfma", "kernel": "ViT-L/14 forward (gemm_bf16_pp_kernel x 96 + attention + LayerNorm, two half-chunk streams)",
                         "achieved": round(tf_exec, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf_exec / PEAK_BF16_TFLOPS, 4),
                         "traffic": pmc.get("vit_hbm_bytes") if traffic_ok else None,
                         "executed_gflop_per_image": round(executed / 1e9, 2),
                         "algorithmic_gflop_per_image": round(VIT_FLOP_PER_IMAGE / 1e9, 2),
                         "frac_on_algorithmic_flops": round(tf_alg / PEAK_BF16_TFLOPS, 4),
                         "note": "achieved/frac count the FLOPs this implementation EXECUTES (the last layer runs on the CLS rows only: "
                                 "3.5 % of the algorithmic 162.03 GFLOP/image are dead rows of the reference graph, output bit-identical); "
                                 "the algorithmic-count fraction is beside it"},
            "roofline_knn": {"bound": "hbm", "kernel": "knn_scan_kernel<12,WaveTopReg> (+2 merge launches), alone on the chip",
                             "achieved": round(knn_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": round(knn_gbs / PEAK_HBM_GBS, 4),
                             "traffic": pmc.get("knn_scan_hbm_bytes") if traffic_ok else None,
                             "note": "the single-pass scan over the fp32 rows: algorithmic bytes = rows x 768 x 4"},
        }
        if ms_knn_two:
            row_bytes = 768 + 12 if args.prefilter == 2 else 768 * 2 + 4      # mirror row + its per-row floats
            two_bytes = len(table) * (row_bytes + 4 * 5 + (4 if args.prefilter == 2 else 0))  # + the coarse keys written once and read four times (+ the bound factors in the collect)
            out["roofline_knn"]["two_stage"] = {
                "ms_per_query": round(ms_knn_two, 4), "bytes_read_and_written": two_bytes,
                "GB_per_s_of_those_bytes": round(two_bytes / (ms_knn_two * 1e-3) / 1e9, 1),
                "frac_of_hbm_peak_on_those_bytes": round(two_bytes / (ms_knn_two * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                "speedup_over_single_pass": round(ms_knn / ms_knn_two, 3),
                "rows_re_evaluated_last_query": pref_cand, "fell_back_to_single_pass": pref_fell_back,
                "traffic_stage1": pmc.get("knn_two_stage_stage1_hbm_bytes") if traffic_ok and args.prefilter == 2 else None}
        if extra:
            out["other_configs"] = extra
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(weights, cfg, None)
        print(json.dumps(out), flush=True)

    model.close()
    table.close()
    for pb in pins:
        pb.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
