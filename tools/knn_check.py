"""Dev harness (GPU): kNN parity vs the oracle + achieved HBM rate. Not a test."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_search_amd import synth
from image_search_amd.search import EmbeddingTable
from oracle.binding import load_oracle, orc_knn, orc_cosine_dist

orc = load_oracle()
N = int(os.environ.get("N", 100_000))
rows = synth.corpus_rows(12, 0, N)
t = EmbeddingTable(768, 0)
t.insert_synthetic(12, 0, N)
back = t.rows(0, 1000)
print("gen parity:", np.array_equal(back, rows[:1000]), np.array_equal(t.rows(N - 7, 7), rows[N - 7:]))
qs = synth.corpus_rows(1012, 0, 4)
for k in (1, 10, 64, 100, 256, 1000, 1500):
    ok = True
    for q in qs[:2]:
        gi, gd = t.knn(q, k)
        oi, od = orc_knn(orc, q, rows, k)
        same_i = np.array_equal(gi, oi); same_d = np.array_equal(gd.view(np.uint32), od.view(np.uint32))
        if not (same_i and same_d):
            ok = False
            bad = np.nonzero(gi != oi)[0]
            print(f"  k={k} mismatch: idx_equal={same_i} dist_bits_equal={same_d} first bad {bad[:5]} gpu {gi[bad[:3]]} {gd[bad[:3]]} cpu {oi[bad[:3]]} {od[bad[:3]]}")
    print(f"k={k}: {'OK' if ok else 'FAIL'}")
t.close()

# perf
NP = int(os.environ.get("NP", 10_000_000))
t = EmbeddingTable(768, 0)
t.reserve(NP)
t0 = time.time(); t.insert_synthetic(0, 0, NP); print(f"generated {NP} rows in {time.time()-t0:.2f}s")
dq = torch.from_numpy(synth.corpus_rows(1, 0, 16)).cuda()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for k in (10, 64, 100, 1000):
    di = torch.empty((16, k), dtype=torch.int64, device="cuda"); dd = torch.empty((16, k), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3): t.knn_device(dq.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr(), s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for it in range(iters): t.knn_device(dq[it % 16].data_ptr(), 1, k, di.data_ptr(), dd.data_ptr(), s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"k={k}: {ms:.3f} ms/query  {NP*3072/ms/1e6:.1f} GB/s  ({NP*3072/ms/1e6/8000*100:.1f}% of 8 TB/s)  qps {1000/ms:.1f}")
# batched: nq queries per table pass
for nq in (2, 4, 8):
    k = 10
    di = torch.empty((nq, k), dtype=torch.int64, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2): t.knn_device(dq.data_ptr(), nq, k, di.data_ptr(), dd.data_ptr(), s, batched=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(10): t.knn_device(dq.data_ptr(), nq, k, di.data_ptr(), dd.data_ptr(), s, batched=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"batched nq={nq}: {ms:.3f} ms/pass  {nq*1000/ms:.0f} queries/s  pass rate {NP*3072/ms/1e6:.0f} GB/s")
