"""A/B of the byte stage-1 scan's prefetch ring (MI_KNN_RING is read when a table is created): ms per two-stage query."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth
from image_search_amd.search import EmbeddingTable
qs = torch.from_numpy(synth.corpus_rows(1, 0, 16)).cuda()
st = torch.cuda.Stream()
for ring in (4, 8, 4, 8):   # (MI_KNN_RING_BPC = 2 | 4: workgroups per CU launched for ring 8, read once per process)
    os.environ["MI_KNN_RING"] = str(ring)
    t = EmbeddingTable(768, 0)
    t.reserve(10_000_000); t.insert_synthetic(0, 0, 10_000_000); t.set_option("prefilter", 2)
    di = torch.empty((1, 10), dtype=torch.int64, device="cuda"); dd = torch.empty((1, 10), dtype=torch.float32, device="cuda")
    for i in range(3):
        t.knn_device(qs[i].data_ptr(), 1, 10, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
    st.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for i in range(40):
        t.knn_device(qs[i % 16].data_ptr(), 1, 10, di.data_ptr(), dd.data_ptr(), st.cuda_stream)
    b.record(st); st.synchronize()
    print(f"ring={ring}: {a.elapsed_time(b) / 40:.4f} ms per two-stage query", flush=True)
    t.close()
