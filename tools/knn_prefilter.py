"""Dev harness (GPU): the two-stage exact search (mi_knn_set_option "prefilter") against the single pass at 10 M rows:
time per query, candidates re-evaluated, and equality of ids and distance bits.  Writes a JSON summary to argv[1]."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_search_amd import synth
from image_search_amd.search import EmbeddingTable

NP = int(os.environ.get("NP", 10_000_000))
t = EmbeddingTable(768, 0)
t.reserve(NP)
t.insert_synthetic(0, 0, NP)
dq = torch.from_numpy(synth.corpus_rows(1, 0, 16)).cuda()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
s = torch.cuda.current_stream().cuda_stream
out = {"rows": NP, "dim": 768, "results": []}
for k in (1, 10, 64, 1000):
    res = {}
    for mode in (0, 1, 2):
        t.set_option("prefilter", mode)
        di = torch.empty((16, k), dtype=torch.int64, device="cuda"); dd = torch.empty((16, k), dtype=torch.float32, device="cuda")
        for _ in range(3): t.knn_device(dq.data_ptr(), 1, k, di.data_ptr(), dd.data_ptr(), s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 20
        e0.record()
        for it in range(iters): t.knn_device(dq[it % 16].data_ptr(), 1, k, di[it % 16].data_ptr(), dd[it % 16].data_ptr(), s)
        e1.record(); torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / iters, di.cpu().numpy().copy(), dd.cpu().numpy().copy(), t.prefilter_stats())
    same = bool(np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2].view(np.uint32), res[1][2].view(np.uint32)))
    same8 = bool(np.array_equal(res[0][1], res[2][1]) and np.array_equal(res[0][2].view(np.uint32), res[2][2].view(np.uint32)))
    r = {"k": k, "single_pass_ms": round(res[0][0], 4), "two_stage_ms": round(res[1][0], 4),
         "candidates_last_query": res[1][3][0], "fell_back": res[1][3][1], "ids_and_distance_bits_equal_16_queries": same,
         "bytes_two_stage_ms": round(res[2][0], 4), "bytes_candidates_last_query": res[2][3][0], "bytes_fell_back": res[2][3][1],
         "bytes_ids_and_distance_bits_equal_16_queries": same8,
         "single_pass_GBps_algorithmic": round(NP * 3072 / res[0][0] / 1e6, 1),
         "two_stage_GBps_of_bytes_it_reads": round(NP * (1536 + 4 + 4 * 5) / res[1][0] / 1e6, 1)}
    print(r, flush=True)
    out["results"].append(r)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
