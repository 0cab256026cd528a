"""k = 1000 (the reference's K) over NP rows: a few queries, for rocprofv3 --kernel-trace --stats.
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_k1000 -o k1000 --output-format csv -- python3 tools/knn_k1000.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_search_amd import synth
from image_search_amd.search import EmbeddingTable
NP = int(os.environ.get("NP", 10_000_000)); K = int(os.environ.get("K", 1000))
t = EmbeddingTable(768, 0); t.reserve(NP); t.insert_synthetic(0, 0, NP)
dq = torch.from_numpy(synth.corpus_rows(1, 0, 8)).cuda()
di = torch.empty((K,), dtype=torch.int64, device="cuda"); dd = torch.empty((K,), dtype=torch.float32, device="cuda")
st = torch.cuda.Stream()
for i in range(10):
    t.knn_device(dq[i % 8].data_ptr(), 1, K, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
st.synchronize()
t.close()
