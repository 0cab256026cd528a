import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from image_search_amd import synth
from image_search_amd.search import EmbeddingTable
NP = int(os.environ.get("NP", 10_000_000)); K = int(os.environ.get("K", 1000))
t = EmbeddingTable(768, 0); t.reserve(NP); t.insert_synthetic(0, 0, NP)
dq = torch.from_numpy(synth.corpus_rows(1, 0, 16)).cuda()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
di = torch.empty((1, K), dtype=torch.int64, device="cuda"); dd = torch.empty((1, K), dtype=torch.float32, device="cuda")
for i in range(3): t.knn_device(dq[i].data_ptr(), 1, K, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10): t.knn_device(dq[i].data_ptr(), 1, K, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
e1.record(); torch.cuda.synchronize()
print(f"N={NP} k={K}: {e0.elapsed_time(e1)/10:.3f} ms/query")
