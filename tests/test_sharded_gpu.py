"""mi_knn_sharded on a real MI355X through the C ABI: the table row-sharded inside ONE process
(BASELINE config 5 in miniature; the reference's one-handle shape, server/src/main.rs:30-35).

A one-GPU box has one device: n > 1 shards are put on device 0 several times (host gather transport);
the RCCL transport is exercised with its one-rank communicator.  On an 8-GPU node the same entry points
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
    assert sh.info() == {"rows": n, "shards": n_shards, "block_rows": block, "transport": "host gather"}
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
    os.remove(prefix + ".1of3.miknn")
    c = ShardedTable(768, [0, 0], 256)
    with pytest.raises(MiError):
        c.load(prefix)
    c.close(); same.close(); a.close()


def test_rccl_transport_with_its_one_rank_communicator(built, monkeypatch):
    """The all-gather path (dlopen'ed librccl, ncclCommInitAll, ncclAllGather under a group) end to end on the
    one device there is; with more devices only the rank count changes."""
    monkeypatch.setenv("MI_KNN_SHARDED_TRANSPORT", "rccl")
    sh = ShardedTable(768, [0])
    monkeypatch.delenv("MI_KNN_SHARDED_TRANSPORT")
    sh.insert_synthetic(61, 0, 10_000)
    one = EmbeddingTable(768, 0)
    one.insert_synthetic(61, 0, 10_000)
    qs = synth.corpus_rows(62, 0, 3)
    assert _same(sh.knn(qs, 10), one.knn(qs, 10))
    assert _same(sh.knn(qs[0], 1000), one.knn(qs[0], 1000))
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
