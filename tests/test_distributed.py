"""The N > 1 path on CPU: two gloo ranks, each owning a contiguous row shard, all-gather
of per-shard top-k and the identical merge on every rank.  The per-shard search is the
oracle here (the HIP scan needs a GPU; it is checked against the same oracle in
test_knn_gpu.py) — what this covers is the sharding arithmetic, the collective and the
merge through the C ABI."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_rows, k, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from image_search_amd import synth
    from image_search_amd.search import gather_and_merge, shard_bounds
    from oracle.binding import load_oracle, orc_knn
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    orc = load_oracle()
    lo, hi = shard_bounds(n_rows, world, rank)
    rows = synth.corpus_rows(77, lo, hi - lo)          # this rank's shard only
    qs = synth.corpus_rows(78, 0, 3)
    li = np.empty((3, k), np.uint64); ld = np.empty((3, k), np.float32)
    for u in range(3):
        li[u], ld[u] = orc_knn(orc, qs[u], rows, k, base=lo)
    gi, gd = gather_and_merge(li, ld, k)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), idx=gi, dist=gd)
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("n_rows,k,world", [(5001, 25, 2), (4099, 10, 3), (100, 25, 3)])
def test_two_rank_sharded_search_equals_single_table(tmp_path, orc, n_rows, k, world):
    """(5001, 25, 2): the plain case; (4099, 10, 3): an odd number of ranks, uneven shards; (100, 25, 3): shards of 33-34 rows,
    barely more than k (every shard's list is almost its whole shard)."""
    from image_search_amd import synth
    from oracle.binding import orc_knn
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_rows, k, str(tmp_path)), nprocs=world, join=True)
    rows = synth.corpus_rows(77, 0, n_rows)
    qs = synth.corpus_rows(78, 0, 3)
    r0 = np.load(tmp_path / "r0.npz")
    for r in range(1, world):
        r1 = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(r0["idx"], r1["idx"]) and np.array_equal(r0["dist"], r1["dist"])
    for u in range(3):
        fi, fd = orc_knn(orc, qs[u], rows, k)
        assert np.array_equal(r0["idx"][u], fi)
        assert np.array_equal(r0["dist"][u].view(np.uint32), fd.view(np.uint32))
