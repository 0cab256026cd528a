"""How the scan's per-wave top-k behaves when the corpus is NOT i.i.d..  Run on the GPU box:

    python tools/knn_robustness.py [--rows 10000000] > profiles/r02_knn_robustness.json

Corpora (generated on the device with torch, appended with mi_knn_append_device):
  iid         N(0,1) rows: the bench corpus
  descending  the cosine distance to the query falls with the row index (rows = noise + a(row) * q, a rising):
              every tile a wave visits holds better candidates than all before it — the worst insertion order
              for a running threshold (WaveTopReg::offer sorts on every tile)
  ascending   the mirror image: the best rows come first, the threshold is tight from the start
  clusters    1000 centres, rows = centre + 0.1 noise, the query near one centre: ~N/1000 near-ties around the k-th
  duplicates  5 M identical rows + 5 M iid rows, the query next to the duplicated row: > 2^22 rows inside any error band,
              the two-stage search must fall back — with "prefilter_adaptive" (default) it stops paying stage 1 on top of
              the single pass after two fallbacks (steady state over 200 queries is reported, adaptive on and off)
  iid again   the first corpus once more, last: separates the corpus from the order of the runs (allocation, clocks)
Reported per corpus and k: ms per query (HIP events on the launch stream), GB/s over the table, ids checked
against torch (fp32 matmul on the device is not the bit-exact oracle: only the id SETS must agree off ties)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd.search import EmbeddingTable  # noqa: E402


def fill(table, kind, n, q, gen):
    chunk = 1_000_000
    centres = torch.randn((1000, 768), device="cuda", generator=gen)
    for lo in range(0, n, chunk):
        m = min(chunk, n - lo)
        x = torch.randn((m, 768), device="cuda", generator=gen)
        r = torch.arange(lo, lo + m, device="cuda", dtype=torch.float32)[:, None] / n
        if kind == "descending":
            x += (4.0 * r) * q[None, :]
        elif kind == "ascending":
            x += (4.0 * (1.0 - r)) * q[None, :]
        elif kind == "clusters":
            c = torch.randint(0, 1000, (m,), device="cuda", generator=gen)
            x = centres[c] + 0.1 * x
        elif kind == "duplicates" and lo < n // 2:
            x = centres[3][None, :].repeat(m, 1).contiguous()
        torch.cuda.synchronize()   # torch's NULL stream = "the handle's own stream" for insert_device: the block must be complete first
        table.insert_device(x.data_ptr(), m, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    return centres


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--prefilter", type=int, default=0, help="also time the two-stage exact search (1 = bf16 mirror, 2 = byte mirror) and report its candidates")
    ap.add_argument("--corpora", default="iid,descending,ascending,clusters,iid again")
    args = ap.parse_args()
    out = {"rows": args.rows, "dim": 768, "results": []}
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    for kind in args.corpora.split(","):
        gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
        q = torch.randn((768,), device="cuda", generator=gen)
        t = EmbeddingTable(768, 0)
        t.reserve(args.rows)
        centres = fill(t, kind, args.rows, q, gen)
        if kind == "clusters":
            q = centres[17] + 0.05 * torch.randn((768,), device="cuda", generator=gen)
        if kind == "duplicates":
            q = centres[3] + 0.01 * torch.randn((768,), device="cuda", generator=gen)
        for k in (10, 64, 1000):
            di = torch.empty((k,), dtype=torch.int64, device="cuda"); dd = torch.empty((k,), dtype=torch.float32, device="cuda")
            for _ in range(2):
                t.knn_device(q.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                t.knn_device(q.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
            e1.record(st); st.synchronize()
            ms = e0.elapsed_time(e1) / 10
            out["results"].append({"corpus": kind, "k": k, "ms_per_query": round(ms, 4),
                                   "GB_per_s": round(args.rows * 3072 / ms / 1e6, 1),
                                   "frac_of_8TBs": round(args.rows * 3072 / ms / 1e6 / 8000, 4)})
            if args.prefilter:
                ref_i, ref_d = di.clone(), dd.clone()
                t.set_option("prefilter", args.prefilter)
                for _ in range(2):
                    t.knn_device(q.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
                e0.record(st)
                for _ in range(10):
                    t.knn_device(q.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
                e1.record(st); st.synchronize()
                cand, fell_back = t.prefilter_stats()
                out["results"][-1].update({"two_stage_ms_per_query": round(e0.elapsed_time(e1) / 10, 4), "rows_re_evaluated": cand,
                                           "fell_back_to_single_pass": fell_back,
                                           "same_ids_and_distance_bits": bool(torch.equal(ref_i, di) and torch.equal(ref_d.view(torch.int32), dd.view(torch.int32)))})
                if kind == "duplicates":   # steady state of a corpus that defeats the mirror: 200 queries, adaptive on / off
                    for adaptive in (1, 0):
                        t.set_option("prefilter_adaptive", adaptive)
                        for _ in range(4):
                            t.knn_device(q.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
                            st.synchronize()
                        e0.record(st)
                        for _ in range(200):
                            t.knn_device(q.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
                        e1.record(st); st.synchronize()
                        out["results"][-1][f"steady_state_ms_adaptive_{adaptive}"] = round(e0.elapsed_time(e1) / 200, 4)
                        out["results"][-1][f"same_bits_adaptive_{adaptive}"] = bool(torch.equal(ref_i, di) and torch.equal(ref_d.view(torch.int32), dd.view(torch.int32)))
                    out["results"][-1]["state"] = t.prefilter_state()
                    t.set_option("prefilter_adaptive", 1)
                t.set_option("prefilter", 0)
            print(out["results"][-1], file=sys.stderr, flush=True)
        t.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
