"""Dev harness (GPU): A/B of the two-stage search's sampled threshold (option prefilter_sample = 0 / 1 / 2) at 10 M rows, single queries and groups."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth
from image_search_amd.search import EmbeddingTable
n = 10_000_000
t = EmbeddingTable(768, 0); t.reserve(n); t.insert_synthetic(0, 0, n)
t.set_option("prefilter", 2)
qs = torch.from_numpy(synth.corpus_rows(1, 0, 32)).cuda()
st = torch.cuda.Stream()
def run(nq, k, batched, reps):
    di = torch.empty((nq, k), dtype=torch.int64, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    for _ in range(2): t.knn_device(qs.data_ptr(), nq, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream, batched=batched)
    st.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(reps): t.knn_device(qs.data_ptr(), nq, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream, batched=batched)
    b.record(st); st.synchronize()
    return a.elapsed_time(b) / reps
out = {}
for rnd in range(3):
    for flag in (0, 1):
        t.set_option("prefilter_sample", flag)
        for k in (10, 64):
            out.setdefault(f"sample={flag} k={k} one query ms", []).append(round(run(16, k, False, 4) / 16, 4))
        out.setdefault(f"sample={flag} k=10 16 per call ms", []).append(round(run(16, 10, True, 10), 4))
print(json.dumps(out, indent=1))
