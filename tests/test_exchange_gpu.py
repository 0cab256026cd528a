"""The exchange step of the row-sharded search on the device (one process per GPU): the per-shard list
leaves the scan in device memory (mi_pipeline_query_device), and the merge of the all-gathered lists can run there too
(mi_knn_merge_device).  Both must be bit-identical to their host forms (mi_pipeline_query, mi_knn_merge), which the
oracle pins (tests/test_pipeline_gpu.py, tests/test_distributed.py)."""
import numpy as np
import pytest

from image_search_amd import synth
from image_search_amd.clip import PRECISION_F32, Model
from image_search_amd.search import EmbeddingTable, Pipeline, merge_candidates, merge_candidates_device

pytestmark = pytest.mark.gpu
NO_ID = np.uint64(0xFFFFFFFFFFFFFFFF)


def _lists(rng, lists, nq, k, short=True, ties=True, nan=True):
    """[lists][nq][k] result lists as the search entry points return them: ascending by (distance key, id), ids unique
    across lists (a row lives in one shard), some lists shorter than k (NO_ID / +inf tail)."""
    idx = np.full((lists, nq, k), NO_ID, np.uint64)
    dist = np.full((lists, nq, k), np.inf, np.float32)
    for u in range(nq):
        ids = rng.permutation(lists * k * 4)[: lists * k].astype(np.uint64)
        d = rng.standard_normal(lists * k).astype(np.float32) * np.float32(0.1) + np.float32(1.0)
        if ties:
            d[rng.integers(0, lists * k, lists * k // 2)] = np.float32(0.875)   # many equal distances: order by id
        if nan:
            d[rng.integers(0, lists * k, 3)] = np.float32(np.nan)              # NaN sorts last among real entries
        for l in range(lists):
            n = k if not short else int(rng.integers(0, k + 1)) if l % 3 == 1 else k
            li, ld = ids[l * k: l * k + n], d[l * k: l * k + n]
            key = np.where(np.isnan(ld), np.uint32(0xFFFFFFFF),
                           np.where(ld.view(np.uint32) & 0x80000000, ~ld.view(np.uint32), ld.view(np.uint32) | 0x80000000)).astype(np.uint64)
            order = np.lexsort((li, key))
            idx[l, u, :n], dist[l, u, :n] = li[order], ld[order]
    return idx, dist


@pytest.mark.parametrize("lists,nq,k", [(1, 1, 1), (2, 1, 10), (8, 1, 10), (8, 3, 1000), (3, 2, 4096), (64, 1, 7), (8, 16, 64)])
def test_device_merge_is_the_host_merge(built, lists, nq, k):
    import torch
    rng = np.random.default_rng(lists * 1000 + k)
    idx, dist = _lists(rng, lists, nq, k)
    d_i = torch.from_numpy(idx.view(np.int64)).cuda()
    d_d = torch.from_numpy(dist).cuda()
    o_i = torch.full((nq, k), -7, dtype=torch.int64, device="cuda")
    o_d = torch.full((nq, k), -7.0, dtype=torch.float32, device="cuda")
    merge_candidates_device(0, d_i.data_ptr(), d_d.data_ptr(), lists, nq, k, o_i.data_ptr(), o_d.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
    gi, gd = o_i.cpu().numpy().view(np.uint64), o_d.cpu().numpy()
    for u in range(nq):
        hi, hd = merge_candidates(idx[:, u, :], dist[:, u, :], k)
        assert np.array_equal(gi[u], hi), (u, np.nonzero(gi[u] != hi)[0][:5])
        assert np.array_equal(gd[u].view(np.uint32), hd.view(np.uint32))


def test_device_merge_of_empty_lists_is_all_none(built):
    import torch
    idx = np.full((4, 1, 10), NO_ID, np.uint64)
    dist = np.full((4, 1, 10), np.inf, np.float32)
    d_i, d_d = torch.from_numpy(idx.view(np.int64)).cuda(), torch.from_numpy(dist).cuda()
    o_i = torch.zeros((1, 10), dtype=torch.int64, device="cuda")
    o_d = torch.zeros((1, 10), dtype=torch.float32, device="cuda")
    merge_candidates_device(0, d_i.data_ptr(), d_d.data_ptr(), 4, 1, 10, o_i.data_ptr(), o_d.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert (o_i.cpu().numpy().view(np.uint64) == NO_ID).all() and np.isinf(o_d.cpu().numpy()).all()


def test_pipeline_query_device_leaves_the_host_results_on_the_device(built, tmp_path):
    """query_device(q) == query(q), bit for bit, and a consumer stream that was made to wait reads the finished list
    without any host synchronisation in between (what ShardExchange.query relies on)."""
    import torch
    cfg = synth.VitConfig.tiny()
    path = str(tmp_path / "tiny.safetensors")
    synth.save_safetensors(synth.vit_weights(cfg, 1), path, {"num_attention_heads": cfg.heads})
    m = Model.from_file(path, 0, PRECISION_F32)
    t = EmbeddingTable(64, 0)
    t.reserve(400_000)
    t.insert_synthetic(3, 0, 300_000)
    pipe = Pipeline(m, t)
    px = synth.preprocess_rgb8(synth.images_u8(3, 40, cfg.image))
    consumer = torch.cuda.Stream()
    qs = synth.corpus_rows(4, 0, 20, 64)
    host, dev, copies = [], [], []
    for u in range(20):
        if u % 5 == 0:
            pipe.ingest(px)                                       # the table grows between queries
        k = 10 if u % 2 == 0 else 1000
        host.append(pipe.query(qs[u], k))
        buf = torch.empty(12 * k, dtype=torch.uint8, device="cuda")
        pipe.query_device(qs[u], k, buf.data_ptr(), buf.data_ptr() + 8 * k, consumer.cuda_stream)
        with torch.cuda.stream(consumer):
            c = torch.empty_like(buf)
            c.copy_(buf, non_blocking=True)                        # ordered behind the scan by the library's event only
        dev.append((k, buf))
        copies.append(c)
    pipe.sync()
    consumer.synchronize()
    for (hi, hd), (k, buf), c in zip(host, dev, copies):
        raw = c.cpu().numpy()
        assert np.array_equal(raw, buf.cpu().numpy())
        assert np.array_equal(raw[:8 * k].view(np.uint64), hi)
        assert np.array_equal(raw[8 * k:].view(np.uint32), hd.view(np.uint32))
    pipe.close()
    t.close()
    m.close()


def test_shard_exchange_over_rccl_one_rank(built, tmp_path):
    """The nccl (= RCCL) form of ShardExchange end to end with the one rank a one-GPU box allows: scan -> packed device
    list -> all_gather_into_tensor -> pinned readback -> merge, several exchanges in flight; the merged list of one
    shard is that shard's own list.  (Two ranks on one GPU are refused by RCCL; bench.py rehearses those over gloo.)"""
    import socket
    import torch
    import torch.distributed as dist
    from image_search_amd.search import ShardExchange
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = synth.VitConfig.tiny()
        path = str(tmp_path / "tiny.safetensors")
        synth.save_safetensors(synth.vit_weights(cfg, 1), path, {"num_attention_heads": cfg.heads})
        m = Model.from_file(path, 0, PRECISION_F32)
        t = EmbeddingTable(64, 0)
        t.reserve(400_000)
        t.insert_synthetic(3, 0, 300_000)
        pipe = Pipeline(m, t)
        px = synth.preprocess_rgb8(synth.images_u8(3, 40, cfg.image))
        qs = synth.corpus_rows(4, 0, 12, 64)
        for k in (10, 1000):
            ex = ShardExchange(k, depth=4)
            assert ex.on_device and ex.world == 1
            want, got = [], []
            for u in range(12):
                if u % 4 == 0:
                    pipe.ingest(px)
                want.append(pipe.query(qs[u], k))
                ex.query(pipe, qs[u])
                while ex.in_flight() > 3:
                    got.append(ex.collect())
            pipe.sync()
            while ex.in_flight():
                got.append(ex.collect())
            assert len(got) == 12
            for (wi, wd), (gi, gd) in zip(want, got):
                assert np.array_equal(wi, gi) and np.array_equal(wd.view(np.uint32), gd.view(np.uint32))
            ex.submit(*want[0])                       # the host-list form over the same backend
            gi, gd = ex.collect()
            assert np.array_equal(want[0][0], gi) and np.array_equal(want[0][1].view(np.uint32), gd.view(np.uint32))
        pipe.close()
        t.close()
        m.close()
    finally:
        dist.destroy_process_group()
