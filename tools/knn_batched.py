"""Queries per second of the batched searches over NP rows (default 10 M x 768): one query per call, and 4 / 8 queries per
call through mi_knn_search_batched_device — as one pass over the fp32 rows and, with the byte mirror, as ONE stage-1 pass
for all queries + a stage 2 each.  Ids and distance bits are compared with single searches.
    python tools/knn_batched.py [rows] > profiles/r03_knn_batched.json"""
import json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth
from image_search_amd.search import EmbeddingTable

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
t = EmbeddingTable(768, 0)
t.reserve(n)
t.insert_synthetic(0, 0, n)
qs = torch.from_numpy(synth.corpus_rows(1, 0, 32)).cuda()
st = torch.cuda.Stream()
out = {"rows": n, "dim": 768, "results": []}


def run(nq, k, batched, reps):
    di = torch.empty((nq, k), dtype=torch.int64, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    for _ in range(2):
        t.knn_device(qs.data_ptr(), nq, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream, batched=batched)
    st.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps):
        t.knn_device(qs.data_ptr(), nq, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream, batched=batched)
    b.record(st); st.synchronize()
    return a.elapsed_time(b) / reps, di.cpu().numpy().copy(), dd.cpu().numpy().view(np.uint32).copy()


for k in (10, 1000):
    t.set_option("prefilter", 0)
    ms1, ri, rd = run(32, k, False, 2)             # 32 single passes: the reference answers
    row = {"k": k, "single_pass_ms_per_query": round(ms1 / 32, 4)}
    for nq in (4, 8):
        ms, i_, d_ = run(nq, k, True, 10)
        row[f"fp32_batched{nq}_ms_per_call"] = round(ms, 4)
        row[f"fp32_batched{nq}_qps"] = round(nq * 1e3 / ms, 1)
        row[f"fp32_batched{nq}_equal"] = bool(np.array_equal(i_, ri[:nq]) and np.array_equal(d_, rd[:nq]))
    t.set_option("prefilter", 2)
    ms, i_, d_ = run(8, k, False, 10)
    row["two_stage_single_ms_per_query"] = round(ms / 8, 4)
    row["two_stage_single_qps"] = round(8e3 / ms, 1)
    row["two_stage_single_equal"] = bool(np.array_equal(i_, ri[:8]) and np.array_equal(d_, rd[:8]))
    for nq in (2, 4, 5, 8, 12, 16):
        ms, i_, d_ = run(nq, k, True, 10)
        row[f"two_stage_batched{nq}_ms_per_call"] = round(ms, 4)
        row[f"two_stage_batched{nq}_qps"] = round(nq * 1e3 / ms, 1)
        if nq <= 32:
            row[f"two_stage_batched{nq}_equal"] = bool(np.array_equal(i_, ri[:nq]) and np.array_equal(d_, rd[:nq]))
    out["results"].append(row)
    print(row, file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
