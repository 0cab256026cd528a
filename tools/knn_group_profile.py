"""rocprofv3 target: a few groups of NQ queries through the two-stage search over 10 M rows (per-kernel times of one group).
    rocprofv3 --kernel-trace --stats -d out -o g --output-format csv -- python3 tools/knn_group_profile.py [nq] [k]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth
from image_search_amd.search import EmbeddingTable
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 16
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 10_000_000
t = EmbeddingTable(768, 0)
t.reserve(n)
t.insert_synthetic(0, 0, n)
t.set_option("prefilter", 2)
qs = torch.from_numpy(synth.corpus_rows(1, 0, 16)).cuda()
st = torch.cuda.Stream()
di = torch.empty((nq, k), dtype=torch.int64, device="cuda"); dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
for _ in range(12):
    t.knn_device(qs.data_ptr(), nq, k, di.data_ptr(), dd.data_ptr(), st.cuda_stream, batched=True)
st.synchronize()
