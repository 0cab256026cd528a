"""mi_knn_sharded on a real MI355X through the C ABI: the table row-sharded inside ONE process
(BASELINE config 5 in miniature; the reference's one-handle shape, server/src/main.rs:30-35).

A one-GPU box has one device: n > 1 shards are put on device 0 several times (the lists meet by device-to-device copies);
the RCCL transport runs its collective and merge on a one-rank communicator (counted: mi_knn_sharded_stats).  On an 8-GPU node the same entry points
take distinct devices and the all-gather runs over xGMI — unmeasured on hardware so far.
Everything is compared bit for bit with ONE mi_knn holding every row and with the oracle."""
import os

import numpy as np
import pytest

from image_search_amd import synth
from image_search_amd._lib import MiError
from image_search_amd.search import EmbeddingTable, ShardedTable
from oracle.binding import orc_knn

pytestmark = pytest.mark.gpu
NO_ID = np.uint64(0xFFFFFFFFFFFFFFFF)


def _same(a, b):
    return np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


def test_one_shard_equals_mi_knn_search_bit_for_bit(built):
    n = 30_000
    one = EmbeddingTable(768, 0)
    one.insert_synthetic(31, 0, n)
    sh = ShardedTable(768, [0])
    sh.insert_synthetic(31, 0, n)
    assert sh.info() == {"rows": n, "shards": 1, "block_rows": 4096, "transport": "single shard"}
    assert sh.stats()["collectives"] == 0
    qs = synth.corpus_rows(32, 0, 5)
    for k in (1, 10, 100, 1000):
        assert _same(sh.knn(qs, k), one.knn(qs, k))
    assert np.array_equal(sh.rows(123, 4000), one.rows(123, 4000))
    sh.close(); one.close()


@pytest.mark.parametrize("n_shards,block", [(2, 4096), (3, 64), (8, 1024)])
def test_shards_on_one_gpu_equal_the_single_table_and_the_oracle(built, orc, n_shards, block):
    n = 50_000 + 17                                   # a ragged last block
    rows = synth.corpus_rows(41, 0, n)
    rows[7] = rows[49_000]                            # a tie across shards: the global id decides
    rows[20_000] = 0.0                                # NaN distance: last
    one = EmbeddingTable(768, 0)
    one.insert(rows)
    sh = ShardedTable(768, [0] * n_shards, block)
    assert sh.insert(rows[:10_000]) == 0
    assert sh.insert(rows[10_000:10_001]) == 10_000   # appends of any size keep ids global and shards contiguous
    assert sh.insert(rows[10_001:]) == 10_001
    assert sh.info() == {"rows": n, "shards": n_shards, "block_rows": block, "transport": "device copies"}
    assert np.array_equal(sh.rows(0, n), rows)
    qs = np.concatenate([synth.corpus_rows(42, 0, 3), rows[7:8]])
    for k in (10, 1000):
        got = sh.knn(qs, k)
        assert _same(got, one.knn(qs, k))
        for u in range(len(qs)):
            oi, od = orc_knn(orc, qs[u], rows, k)
            assert np.array_equal(got[0][u], oi) and np.array_equal(got[1][u].view(np.uint32), od.view(np.uint32))
    assert list(sh.knn(rows[7], 2)[0]) == [7, 49_000]
    # k above the table: the tail is NO_ID / +inf, once, after the merge
    gi, gd = sh.knn(qs[0], 2 * n)
    assert (gi[n - 1] != NO_ID) and (gi[n:] == NO_ID).all() and np.isinf(gd[n:]).all()
    sh.close(); one.close()


def test_save_load_and_rebalancing_to_another_shard_count(built, tmp_path):
    n = 20_000
    rows = synth.corpus_rows(51, 0, n)
    a = ShardedTable(768, [0, 0, 0], 256)
    a.insert(rows)
    prefix = str(tmp_path / "table")
    a.save(prefix)
    q = synth.corpus_rows(52, 0, 2)
    want = a.knn(q, 50)
    same = ShardedTable(768, [0, 0, 0], 256)
    same.load(prefix)
    assert len(same) == n and _same(same.knn(q, 50), want)
    for devices, block in (([0], 0), ([0, 0], 1024), ([0] * 5, 64)):     # re-dealt through the host
        b = ShardedTable(768, devices, block)
        b.load(prefix)
        assert len(b) == n and np.array_equal(b.rows(0, n), rows) and _same(b.knn(q, 50), want)
        with pytest.raises(MiError):
            b.load(prefix)                                                   # needs an empty table
        b.close()
    os.remove(prefix + ".g1.1of3.miknn")
    c = ShardedTable(768, [0, 0], 256)
    with pytest.raises(MiError):
        c.load(prefix)
    assert len(c) == 0 and [c.shard_rows(s) for s in range(2)] == [0, 0]    # a failed load leaves the table empty ...
    c.insert(rows[:700])                                                     # ... and usable
    assert np.array_equal(c.rows(0, 700), rows[:700])
    c.close(); same.close(); a.close()


def test_a_save_that_fails_midway_leaves_the_previous_generation_loadable(built, tmp_path, monkeypatch):
    """shard files used to be replaced one by one under the same names, so an error after the first rename left
    mixed generations that no longer loaded.  Every save now writes a new generation and the manifest names it last."""
    n = 9_000
    rows = synth.corpus_rows(53, 0, n)
    a = ShardedTable(768, [0, 0, 0], 256)
    a.insert(rows[:6_000])
    prefix = str(tmp_path / "t")
    a.save(prefix)                                                           # generation 1
    a.insert(rows[6_000:])
    monkeypatch.setenv("MI_KNN_SHARDED_SAVE_FAIL_AFTER", "2")                # the third shard file cannot be written
    with pytest.raises(MiError):
        a.save(prefix)
    monkeypatch.delenv("MI_KNN_SHARDED_SAVE_FAIL_AFTER")
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".miknn")) == [f"t.g1.{s}of3.miknn" for s in range(3)]
    b = ShardedTable(768, [0, 0, 0], 256)
    b.load(prefix)                                                           # what the crash left: generation 1, complete
    assert len(b) == 6_000 and np.array_equal(b.rows(0, 6_000), rows[:6_000])
    b.close()
    a.save(prefix)                                                           # generation 2; generation 1 is deleted after the manifest
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".miknn")) == [f"t.g2.{s}of3.miknn" for s in range(3)]
    c = ShardedTable(768, [0, 0], 1024)
    c.load(prefix)
    assert len(c) == n and np.array_equal(c.rows(0, n), rows)
    c.close(); a.close()


def test_rows_born_on_the_device_reach_their_shards_without_the_host(built, orc):
    """mi_knn_sharded_append_device: runs of a device buffer routed to their shards (here: device-to-device; between GPUs:
    hipMemcpyPeerAsync), on the producer's stream, searches ordered behind them by events only."""
    import torch
    n = 30_000 + 5
    rows = synth.corpus_rows(71, 0, n)
    one = EmbeddingTable(768, 0)
    one.insert(rows)
    sh = ShardedTable(768, [0, 0, 0], 192)
    producer = torch.cuda.Stream()
    with torch.cuda.stream(producer):
        d = torch.from_numpy(rows).cuda(non_blocking=True)                  # "produced" on the stream the append is given
    assert sh.insert(rows[:1_000]) == 0                                      # host and device appends interleave
    assert sh.insert_device(d[1_000:].data_ptr(), 10_000, 0, producer.cuda_stream) == 1_000
    assert sh.insert_device(d[11_000:].data_ptr(), 1, 0, producer.cuda_stream) == 11_000
    assert sh.insert_device(d[11_001:].data_ptr(), n - 11_001, 0, producer.cuda_stream) == 11_001
    qs = synth.corpus_rows(72, 0, 3)
    got = sh.knn(qs, 10)                                                     # no synchronisation in between
    assert _same(got, one.knn(qs, 10))
    for u in range(3):
        oi, od = orc_knn(orc, qs[u], rows, 10)
        assert np.array_equal(got[0][u], oi) and np.array_equal(got[1][u].view(np.uint32), od.view(np.uint32))
    assert np.array_equal(sh.rows(0, n), rows)
    assert sum(sh.shard_rows(s) for s in range(3)) == n
    sh.close(); one.close()


def test_async_searches_in_flight_and_the_ring(built):
    n = 40_000
    sh = ShardedTable(768, [0, 0, 0, 0], 512)
    sh.insert_synthetic(81, 0, n)
    one = EmbeddingTable(768, 0)
    one.insert_synthetic(81, 0, n)
    qs = synth.corpus_rows(82, 0, 20)
    pend = [sh.knn_async(qs[u], 10 if u % 3 else 1000) for u in range(20)]   # more than the 8 slots: the oldest are delivered on the way
    sh.sync()
    for u, (gi, gd) in enumerate(pend):
        assert _same((gi[0], gd[0]), one.knn(qs[u], 10 if u % 3 else 1000))
    gi, gd = sh.knn_async(qs[:5], 64)                                        # several queries per call
    sh.sync()
    assert _same((gi, gd), one.knn(qs[:5], 64))
    sh.close(); one.close()


def test_a_live_table_changes_its_layout_device_to_device(built):
    """mi_knn_sharded_rebalance: the f3 remainder — a shard-count change of a live table without a trip through the host."""
    n = 25_000 + 33
    rows = synth.corpus_rows(91, 0, n)
    a = ShardedTable(768, [0, 0, 0], 256)
    a.insert(rows)
    q = synth.corpus_rows(92, 0, 2)
    want = a.knn(q, 100)
    for devices, block in (([0, 0], 1024), ([0] * 5, 64), ([0], 0)):
        b = ShardedTable(768, devices, block)
        b.rebalance_from(a)
        assert len(b) == n and np.array_equal(b.rows(0, n), rows) and _same(b.knn(q, 100), want)
        with pytest.raises(MiError):
            b.rebalance_from(a)                                              # needs an empty destination
        b.close()
    assert len(a) == n and _same(a.knn(q, 100), want)                        # the source is unchanged
    a.close()


def test_rccl_transport_with_its_one_rank_communicator_really_issues_the_collective(built, monkeypatch):
    """VERDICT r3 / ADVICE r3: the exchange block used to sit under `n > 1`, so this test reached dlopen +
    ncclCommInitAll(1) and nothing else.  A table made with the RCCL transport now runs the exchange whatever its rank
    count: ONE ncclAllGather of the packed [k x u64 | k x f32] record per shard and search (under ncclGroupStart/End, on
    the shard's stream), then the device merge of the gathered records, then one readback — counted by the handle.
    With more devices only the rank count changes; that run is still to come (unmeasured on > 1 GPU)."""
    monkeypatch.setenv("MI_KNN_SHARDED_TRANSPORT", "rccl")
    sh = ShardedTable(768, [0])
    monkeypatch.delenv("MI_KNN_SHARDED_TRANSPORT")
    assert sh.info()["transport"] == "rccl all-gather"
    sh.insert_synthetic(61, 0, 10_000)
    one = EmbeddingTable(768, 0)
    one.insert_synthetic(61, 0, 10_000)
    qs = synth.corpus_rows(62, 0, 3)
    assert sh.stats() == {"searches": 0, "collectives": 0, "copies": 0, "merges": 0}
    assert _same(sh.knn(qs, 10), one.knn(qs, 10))            # nq = 3, k = 10: record = 360 bytes -> padded to 368
    assert sh.stats() == {"searches": 1, "collectives": 1, "copies": 0, "merges": 1}
    assert _same(sh.knn(qs[0], 1000), one.knn(qs[0], 1000))
    assert _same(sh.knn(qs[1], 1), one.knn(qs[1], 1))        # an odd record: 12 bytes, padded to 16
    pend = [sh.knn_async(qs[u % 3], 7) for u in range(10)]   # through the slot ring, collectives back to back
    sh.sync()
    for u, (gi, gd) in enumerate(pend):
        assert _same((gi[0], gd[0]), one.knn(qs[u % 3], 7))
    assert sh.stats() == {"searches": 13, "collectives": 13, "copies": 0, "merges": 13}
    sh.close(); one.close()


def test_the_copy_transport_moves_one_packed_record_per_shard(built):
    """ids and distances of a shard's answer travel as ONE piece (they were two copies / two all-gathers per shard)."""
    sh = ShardedTable(768, [0, 0, 0], 64)
    sh.insert_synthetic(63, 0, 5_000)
    one = EmbeddingTable(768, 0)
    one.insert_synthetic(63, 0, 5_000)
    qs = synth.corpus_rows(64, 0, 5)
    for k in (1, 3, 10, 1000):
        assert _same(sh.knn(qs, k), one.knn(qs, k))
    assert sh.stats() == {"searches": 4, "collectives": 0, "copies": 12, "merges": 4}
    single = ShardedTable(768, [0])
    single.insert_synthetic(63, 0, 5_000)
    assert _same(single.knn(qs, 10), one.knn(qs, 10))
    assert single.stats() == {"searches": 1, "collectives": 0, "copies": 0, "merges": 0}   # one shard gathers nothing
    single.close(); sh.close(); one.close()


def test_device_append_to_an_unreserved_table_behind_a_busy_producer(built):
    """a run of the same call landing on a shard a second time could reallocate that shard while the call's
    earlier copies were still queued on the producer's stream (grow() only waits for EARLIER calls' work): rows lost,
    or a freed buffer written.  Every touched shard is now grown to its final size before the first copy is enqueued.
    More than n_shards * block rows, unreserved, behind a producer stream that is kept busy."""
    import torch
    n_sh, block = 3, 64
    n = 20 * n_sh * block + 11
    rows = synth.corpus_rows(73, 0, n)
    producer = torch.cuda.Stream()
    sh = ShardedTable(768, [0] * n_sh, block)
    with torch.cuda.stream(producer):
        d = torch.from_numpy(rows).cuda(non_blocking=True)
        busy = torch.empty((8192, 8192), device="cuda")
        for _ in range(30):                                 # tens of milliseconds of work in front of the copies
            busy = torch.mm(busy.fill_(1e-3), busy)
    assert sh.insert_device(d.data_ptr(), n, 0, producer.cuda_stream) == 0
    more = synth.corpus_rows(74, 0, 5 * n_sh * block)
    with torch.cuda.stream(producer):
        d2 = torch.from_numpy(more).cuda(non_blocking=True)
        for _ in range(10):
            busy = torch.mm(busy.fill_(1e-3), busy)
    assert sh.insert_device(d2.data_ptr(), len(more), 0, producer.cuda_stream) == n   # grows every shard again, copies of call 1 maybe still queued
    allrows = np.concatenate([rows, more])
    assert np.array_equal(sh.rows(0, len(allrows)).view(np.uint32), allrows.view(np.uint32))
    one = EmbeddingTable(768, 0)
    one.insert(allrows)
    q = synth.corpus_rows(75, 0, 2)
    assert _same(sh.knn(q, 25), one.knn(q, 25))
    torch.cuda.synchronize()
    sh.close(); one.close()


def test_argument_errors(built):
    with pytest.raises(MiError):
        ShardedTable(768, [0, 0], 100)          # block not a multiple of 64
    with pytest.raises(MiError):
        ShardedTable(768, [0, 99])              # no such device
    sh = ShardedTable(768, [0, 0])
    with pytest.raises(MiError):
        sh.rows(0, 1)                           # empty
    gi, gd = sh.knn(np.ones(768, np.float32), 3)
    assert (gi == NO_ID).all() and np.isinf(gd).all()
    sh.close()
