"""Two-stage exact search (mi_knn_set_option "prefilter"): ids AND distance bits must equal the single-pass scan's on
every corpus, including the ones built to defeat the bf16 prefilter (it then hands over to the single pass on the device)."""
import numpy as np
import pytest

from image_search_amd.search import EmbeddingTable

pytestmark = pytest.mark.gpu
DIM = 768
N = 300_000  # above the 2^18-row threshold of the prefilter; 0.9 GB of fp32 rows


MODE = 1  # which mirror the helpers switch on: 1 = bf16 (most of this file), 2 = bytes (the tests at the end)


def _both(t, q, k):
    t.set_option("prefilter", 0)
    a = t.knn(q, k)
    t.set_option("prefilter", MODE)
    b = t.knn(q, k)
    return a, b


def _same(a, b):
    assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))


@pytest.fixture(scope="module")
def table(built):
    t = EmbeddingTable(DIM, 0)
    t.reserve(N + 4096)
    t.insert_synthetic(7, 0, N)
    yield t
    t.close()


@pytest.mark.parametrize("k", [1, 10, 64, 65, 1000, 4096])
def test_random_corpus_matches_the_single_pass_bit_for_bit(table, k):
    rng = np.random.default_rng(k)
    for j in range(4):
        q = rng.standard_normal(DIM).astype(np.float32)
        if j == 3:
            q = table.rows(12345, 1)[0].copy()  # a stored row as the query: distance ~0 at rank 0
        _same(*_both(table, q, k))
        cand, fell_back = table.prefilter_stats()
        assert not fell_back and k <= cand <= 6 * k + 64, (cand, fell_back)  # the two stages answered, from a handful of rows


def test_rows_appended_after_the_mirror_was_built_are_found(table):
    rng = np.random.default_rng(3)
    q = rng.standard_normal(DIM).astype(np.float32)
    table.set_option("prefilter", 1)
    before = table.knn(q, 5)
    n0 = len(table)
    table.insert((q[None, :] * np.array([[1.0], [2.5], [0.5]], np.float32)).astype(np.float32))  # three rows at distance ~0
    after = table.knn(q, 5)
    assert set(after[0][:3].tolist()) == {n0, n0 + 1, n0 + 2} and np.array_equal(after[0][3:], before[0][:2])
    _same(*_both(table, q, 5))


def test_rows_the_error_bound_does_not_cover_are_always_rescored(table):
    """non-finite, huge and vanishing rows: marked in the mirror, evaluated from the fp32 rows like everything that matters"""
    rng = np.random.default_rng(4)
    q = rng.standard_normal(DIM).astype(np.float32)
    odd = rng.standard_normal((6, DIM)).astype(np.float32)
    odd[0, 5] = np.nan
    odd[1, 9] = np.inf
    odd[2] *= np.float32(3.2e38) / np.abs(odd[2]).max()   # finite in fp32, inf in bf16
    odd[3] = 0.0                                          # zero norm: NaN distance, ranked last by both paths
    odd[4] = q * np.float32(1e-17)                        # squared norm below the stored-norm window; true distance ~0
    odd[5] = q * np.float32(1e19)
    table.insert(odd)
    for k in (3, 64, 1000):
        _same(*_both(table, q, k))
    table.set_option("prefilter", 1)
    assert len(table) - 2 in table.knn(q, 3)[0].tolist()  # the 1e-17-scaled copy of the query is among the nearest


def test_rows_at_the_worst_case_of_the_bf16_rounding(table):
    """Rows whose every element sits just under / just over half a bf16 ulp, signs aligned with the query: the coarse distance
    of one kind errs by up to +2^-8, of the other by -2^-8 — exact neighbours trade places with decoys by the full width of
    the band.  (A band built on 2^-9 — bf16 mistaken for 9 significant bits — loses true neighbours here.)"""
    rng = np.random.default_rng(6)
    q = rng.standard_normal(DIM).astype(np.float32)
    n = 1500
    base = (np.abs(q)[None, :] * (1.0 + 0.5 * rng.random((2 * n, DIM)))).astype(np.float32)
    u = base.view(np.uint32) & np.uint32(0xFFFF0000)
    low = (u[:n] | np.uint32(0x7FFF)).view(np.float32)    # rounds DOWN by almost half an ulp: coarse dot too small
    high = (u[n:] | np.uint32(0x8001)).view(np.float32)   # rounds UP: coarse dot too large
    rows = (np.concatenate([low, high]) * np.sign(q)[None, :]).astype(np.float32)
    rows = rows[rng.permutation(2 * n)]
    table.insert(rows)
    for k in (1, 10, 64, 1000):
        _same(*_both(table, q, k))
        cand, fell_back = table.prefilter_stats()
        assert not fell_back and cand >= k, (k, cand, fell_back)


def test_a_corpus_inside_the_error_band_is_re_evaluated_whole(built):
    """every row within 2 eps of the k-th: stage 2 re-evaluates them all (the radix select over the candidates scales)"""
    rng = np.random.default_rng(5)
    base = rng.standard_normal(DIM).astype(np.float32)
    t = EmbeddingTable(DIM, 0)
    chunk = 50_000
    for i in range(6):  # 300 k rows, all within 1e-4 (relative) of one direction
        rows = base[None, :] * (1.0 + 1e-4 * rng.standard_normal((chunk, 1))).astype(np.float32)
        rows += (1e-4 * rng.standard_normal((chunk, DIM))).astype(np.float32)
        t.insert(rows.astype(np.float32))
    q = (base + 0.01 * rng.standard_normal(DIM)).astype(np.float32)
    for k in (1, 10, 64, 1000):
        _same(*_both(t, q, k))
        cand, fell_back = t.prefilter_stats()
        assert not fell_back and cand == len(t)
    # and exact duplicates: ties must break by id on both paths
    t2 = EmbeddingTable(DIM, 0)
    for i in range(6):
        t2.insert(np.repeat(base[None, :], chunk, 0))
    for k in (10, 1000):
        _same(*_both(t2, q, k))
        assert t2.knn(q, k)[0].tolist() == list(range(k))
    t.close()
    t2.close()


def test_more_candidates_than_stage_two_accepts_fall_back_to_the_single_pass(built):
    """4.4 M identical rows (> 2^22 candidates): the gated single pass answers, on the device, with the same bits"""
    import torch
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    base = torch.randn((DIM,), device="cuda", generator=gen)
    t = EmbeddingTable(DIM, 0)
    t.reserve(4_400_000)
    x = base[None, :].repeat(100_000, 1).contiguous()  # exact duplicates: one coarse key for all of them
    torch.cuda.synchronize()   # torch's NULL stream means "the handle's own stream" to insert_device: x must be complete first
    for i in range(44):
        t.insert_device(x.data_ptr(), 100_000, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    q = (base + 0.01 * torch.randn((DIM,), device="cuda", generator=gen)).cpu().numpy()
    for k in (10, 1000):
        _same(*_both(t, q, k))
        cand, fell_back = t.prefilter_stats()
        assert fell_back and cand > (1 << 22), (k, cand, fell_back, len(t))
        assert t.knn(q, k)[0].tolist() == list(range(k))  # ties by id
    t.close()


def test_sharded_table_forwards_the_option_and_stays_bit_identical(built):
    from image_search_amd.search import ShardedTable
    one = EmbeddingTable(DIM, 0)
    one.insert_synthetic(9, 0, 3 * N)
    st = ShardedTable(DIM, [0, 0, 0], block_rows=4096)  # three shards of >= 2^18 rows each, on one device
    st.insert_synthetic(9, 0, 3 * N)
    rng = np.random.default_rng(8)
    for mode in (1, 2):
        st.set_option("prefilter", mode)
        for k in (10, 1000):
            q = rng.standard_normal(DIM).astype(np.float32)
            _same(one.knn(q, k), st.knn(q, k))
    one.close()
    st.close()


@pytest.mark.parametrize("dim", [128, 256, 512, 1024])
def test_other_row_widths(built, dim):
    """every instantiation of the mirror / coarse / rescore kernels (dim / 64 in {2, 4, 8, 16}; 12 is the rest of this file)"""
    import torch
    gen = torch.Generator(device="cuda"); gen.manual_seed(dim)
    t = EmbeddingTable(dim, 0)
    n = 270_000
    x = torch.randn((n, dim), device="cuda", generator=gen)
    torch.cuda.synchronize()   # (insert_device runs on the handle's own stream when handed torch's NULL stream)
    t.insert_device(x.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for j, k in enumerate((1, 10, 64, 300)):
        q = torch.randn((dim,), device="cuda", generator=gen).cpu().numpy()
        t.set_option("prefilter", 0)
        a = t.knn(q, k)
        t.set_option("prefilter", 1)
        b = t.knn(q, k)
        _same(a, b)
        cand, fell_back = t.prefilter_stats()
        assert not fell_back and cand >= k
    t.close()


def test_a_table_that_does_not_qualify_is_served_by_the_single_pass(built):
    t = EmbeddingTable(64, 0)   # dim 64: rows of half a 256-byte bf16 chunk; the option is accepted, the search ignores it
    t.insert_synthetic(1, 0, 300_000)
    t.set_option("prefilter", 1)
    q = np.random.default_rng(0).standard_normal(64).astype(np.float32)
    a = t.knn(q, 10)
    assert t.prefilter_stats() == (0, False)
    t.set_option("prefilter", 0)
    _same(a, t.knn(q, 10))
    t.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_through_the_pipeline_rows_written_by_the_tower_reach_the_mirror(built, tmp_path, mode):
    """mi_pipeline_ingest writes embeddings straight into the table on the ingest stream; the next query (search stream)
    must mirror those rows before it scans: the chunk's own images are their own nearest neighbours, on both paths."""
    from image_search_amd import synth
    from image_search_amd.clip import PRECISION_F32, Model
    from image_search_amd.search import Pipeline
    cfg = synth.VitConfig(hidden=128, layers=2, heads=2, ff=512, patch=14, image=56, proj=256)
    path = str(tmp_path / "small.safetensors")
    synth.save_safetensors(synth.vit_weights(cfg, 5), path, {"num_attention_heads": cfg.heads})
    m = Model.from_file(path, 0, PRECISION_F32)
    t = EmbeddingTable(cfg.proj, 0)
    t.reserve(270_000 + 64)
    t.insert_synthetic(3, 0, 270_000)
    t.set_option("prefilter", mode)
    pipe = Pipeline(m, t)
    px = synth.preprocess_rgb8(synth.images_u8(77, 16, cfg.image))
    emb = m.forward(px)
    results = []
    for c in range(2):
        first = pipe.ingest(px[8 * c:8 * c + 8])
        assert first == 270_000 + 8 * c
        results.append(pipe.query(emb[8 * c + 3], 5))       # one of the images just ingested: distance ~0 to its own row
    pipe.sync()
    for c, (ids, dist) in enumerate(results):
        assert ids[0] == 270_000 + 8 * c + 3 and abs(float(dist[0])) < 1e-6
        assert t.prefilter_stats()[1] is False
    # and what the pipeline answered is what the single pass answers now
    t.set_option("prefilter", 0)
    for c, (ids, dist) in enumerate(results[-1:]):
        a = t.knn(emb[8 + 3], 5)
        assert np.array_equal(a[0], ids) and np.array_equal(a[1].view(np.uint32), dist.view(np.uint32))
    pipe.close()
    t.close()
    m.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_against_the_oracle_directly(built, orc, mode):
    """not only equal to the single pass (which the other tests pin to the oracle): ids and distance bits of oracle/oracle.c"""
    from image_search_amd import synth
    from oracle.binding import orc_knn
    n = 270_000
    rows = synth.corpus_rows(31, 0, n)
    t = EmbeddingTable(DIM, 0)
    t.insert_synthetic(31, 0, n)
    t.set_option("prefilter", mode)
    for q in synth.corpus_rows(1031, 0, 2):
        for k in (10, 1000):
            gi, gd = t.knn(q, k)
            oi, od = orc_knn(orc, q, rows, k)
            assert np.array_equal(gi, oi) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))
            assert t.prefilter_stats()[1] is False
    t.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_degenerate_queries_give_what_the_single_pass_gives(table, mode, monkeypatch):
    """a zero query (0 / 0), a NaN in the query, an infinite one: every distance is NaN, the answer is rows 0 .. k-1 with NaN
    distances on both paths (the coarse stage cannot exclude anything and says so)"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "MODE", mode)
    q0 = np.zeros(DIM, np.float32)
    qn = np.random.default_rng(1).standard_normal(DIM).astype(np.float32)
    qn[17] = np.nan
    qi = np.random.default_rng(2).standard_normal(DIM).astype(np.float32)
    qi[3] = np.inf
    for q in (q0, qn, qi):
        for k in (5, 1000):
            a, b = _both(table, q, k)
            _same(a, b)
            assert np.isnan(b[1]).all() and b[0].tolist() == list(range(k))


def test_option_errors(built):
    t = EmbeddingTable(DIM, 0)
    with pytest.raises(RuntimeError, match="unknown option"):
        t.set_option("nope", 1)
    with pytest.raises(RuntimeError, match="0, 1 or 2"):
        t.set_option("prefilter", 3)
    t.close()


# ---- the byte mirror ("prefilter" = 2): the same corpora, the per-row bound -----------------------------------------------------

@pytest.fixture()
def bytes_mode(monkeypatch):
    import sys
    monkeypatch.setattr(sys.modules[__name__], "MODE", 2)


@pytest.mark.parametrize("k", [1, 10, 64, 1000])
def test_bytes_random_corpus(table, bytes_mode, k):
    rng = np.random.default_rng(100 + k)
    for j in range(3):
        q = rng.standard_normal(DIM).astype(np.float32)
        if j == 2:
            q = table.rows(54321, 1)[0].copy()
        _same(*_both(table, q, k))
        cand, fell_back = table.prefilter_stats()
        assert not fell_back and cand >= k, (cand, fell_back)


def test_bytes_rows_at_the_worst_case_of_the_quantisation(built, bytes_mode):
    """every element half a quantisation step off, the direction chosen against (or for) the query: the coarse distance errs by
    the whole per-row bound, exact neighbours hide behind decoys"""
    rng = np.random.default_rng(16)
    t = EmbeddingTable(DIM, 0)
    t.insert_synthetic(21, 0, N)
    q = rng.standard_normal(DIM).astype(np.float32)
    n = 1500
    steps = rng.integers(-100, 101, (2 * n, DIM)).astype(np.float32)
    steps[:, 0] = 127.0                       # the row maximum: element 0 = 127 steps (scale = 1 step exactly)
    half = np.where(np.sign(q)[None, :] > 0, 0.49, -0.49).astype(np.float32)
    rows = np.concatenate([steps[:n] + half, steps[n:] - half])   # rounds to `steps`: dot too small / too large by sum |q_j| / 2
    rows[:, 0] = 127.0
    rows += (3.0 * q)[None, :]                # lean towards the query so that these rows are the neighbourhood
    rows = rows[rng.permutation(2 * n)].astype(np.float32)
    t.insert(rows)
    for k in (1, 10, 64, 1000):
        _same(*_both(t, q, k))
        cand, fell_back = t.prefilter_stats()
        assert not fell_back and cand >= k
    t.close()


def test_bytes_odd_rows_duplicates_and_appends(built, bytes_mode):
    rng = np.random.default_rng(17)
    t = EmbeddingTable(DIM, 0)
    t.insert_synthetic(22, 0, N)
    q = rng.standard_normal(DIM).astype(np.float32)
    t.set_option("prefilter", 2)
    t.knn(q, 5)                               # builds the mirror; what follows is caught up by later searches
    odd = rng.standard_normal((7, DIM)).astype(np.float32)
    odd[0, 5] = np.nan
    odd[1, 9] = np.inf
    odd[2] *= np.float32(3.2e38) / np.abs(odd[2]).max()
    odd[3] = 0.0
    odd[4] = q * np.float32(1e-17)
    odd[5] = q * np.float32(1e19)
    odd[6] = q
    odd[6, 3] = 1e6                           # one dominant element: every other byte of the row is 128 +- 0
    t.insert(odd)
    t.insert(np.repeat(q[None, :] * np.float32(0.5), 3000, 0))  # 3 000 exact duplicates at distance ~0: ties by id
    for k in (3, 64, 1000):
        _same(*_both(t, q, k))
    t.close()


def test_bytes_narrow_rows_are_served_by_the_single_pass(built, bytes_mode):
    t = EmbeddingTable(128, 0)                # 128 bytes per mirror row: not a whole 256-byte chunk
    t.insert_synthetic(1, 0, 300_000)
    q = np.random.default_rng(0).standard_normal(128).astype(np.float32)
    _same(*_both(t, q, 10))
    assert t.prefilter_stats() == (0, False)
    t.close()


def test_repeated_fallbacks_switch_stage_one_off_for_a_while_and_probe_again(built):
    """A corpus with > 2^22 rows inside the band used to pay stage 1 plus the single pass on EVERY query.  The search now
    reads its own candidate count / fallback flag back asynchronously; after two fallbacks in a row the next 64 queries
    run the single pass alone, then stage 1 is probed again.  Results are the single pass's throughout."""
    import torch
    gen = torch.Generator(device="cuda"); gen.manual_seed(12)
    base = torch.randn((DIM,), device="cuda", generator=gen)
    t = EmbeddingTable(DIM, 0)
    t.reserve(4_400_000)
    x = base[None, :].repeat(100_000, 1).contiguous()
    torch.cuda.synchronize()
    for i in range(44):
        t.insert_device(x.data_ptr(), 100_000, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    q = (base + 0.01 * torch.randn((DIM,), device="cuda", generator=gen)).cpu().numpy()
    want = t.knn(q, 10)
    t.set_option("prefilter", 1)
    seen = []
    for j in range(70):
        _same(t.knn(q, 10), want)
        seen.append(t.prefilter_stats())
    # queries 0 and 1 fall back; 2 .. 65 skip stage 1 (no stats); 66 and 67 probe (and fall back) again; then skipping resumes
    assert all(fb and c > (1 << 22) for c, fb in seen[:2]), seen[:3]
    assert seen[2:66] == [(0, False)] * 64, seen[2:8]
    assert all(fb for _, fb in seen[66:68]) and seen[68:] == [(0, False)] * 2, seen[64:]
    st = t.prefilter_state()
    assert st["searches_skipped"] == 66 and st["skips_left"] == 62, st
    t.set_option("prefilter_adaptive", 0)                      # opt out: every query tries stage 1 again
    for j in range(4):
        _same(t.knn(q, 10), want)
        assert t.prefilter_stats()[1] is True
    t.close()


def test_a_growing_table_keeps_its_mirror_and_refreshes_the_channel_scales_at_4x(built):
    """any growth of the table's allocation used to throw the mirror away (a full fp32 pass inside the next
    query).  Now the mirrored rows move to the larger buffers and only the new rows are converted; the byte mirror's
    channel scales (taken from a sample spread over the table) are taken again once the table has grown 4x."""
    t = EmbeddingTable(DIM, 0)
    n0 = 300_000
    t.insert_synthetic(21, 0, n0)                              # no reserve: every append below reallocates
    rng = np.random.default_rng(5)
    q = rng.standard_normal(DIM).astype(np.float32)
    t.set_option("prefilter", 2)
    t.knn(q, 10)
    assert t.prefilter_state()["scales_taken_at_rows"] == n0
    rows = n0
    for step in range(4):
        extra = 250_000 if step < 3 else 500_000
        t.insert_synthetic(22 + step, 0, extra)
        rows += extra
        got = t.knn(q, 100)
        cand, fell_back = t.prefilter_stats()
        assert not fell_back and cand >= 100
        t.set_option("prefilter_adaptive", 1)                  # (no-op toggle: does not touch the mirror)
        taken = t.prefilter_state()["scales_taken_at_rows"]
        assert taken == (n0 if rows < 4 * n0 else rows), (rows, taken)
        ref = EmbeddingTable(DIM, 0)                           # the same rows, single pass
        ref.insert_synthetic(21, 0, n0)
        for s2 in range(step + 1):
            ref.insert_synthetic(22 + s2, 0, 250_000 if s2 < 3 else 500_000)
        _same(got, ref.knn(q, 100))
        ref.close()
    t.close()


@pytest.mark.parametrize("k", [1, 10, 64, 1000])
def test_batched_queries_share_one_pass_over_the_byte_mirror(built, k):
    """mi_knn_search with nq > 1 and mi_knn_search_batched_device used to bypass the prefilter.  With the
    byte mirror a group of queries shares ONE stage-1 pass — round 3: 8, 4 or 2 queries on the vector ALU
    (knn_scan_coarse8_batched_kernel: a query per 16-lane group); round 4: any group size up to 16 on the matrix pipe
    (knn_scan_coarse8_mfma_kernel: the queries as int8 digits, exact integer dot products) —
    and, since round 4, ONE launch of every later kernel — selects, collect, stage 2, select over the candidates, sort,
    the gated single pass, finalize — with the query as grid.y on per-query workspaces (QGroup): ids and distance bits of nq
    single searches."""
    import torch
    t = EmbeddingTable(DIM, 0)
    t.insert_synthetic(31, 0, N + 1234)                         # a ragged last tile
    rng = np.random.default_rng(100 + k)
    qs = rng.standard_normal((16, DIM)).astype(np.float32)
    qs[3] = t.rows(777, 1)[0]                                   # a stored row as a query
    qs[9] = 0.0                                                 # a zero query: every distance NaN
    want = t.knn(qs, k)                                         # prefilter off: the single pass (batched fp32 kernel for k <= 64)
    t.set_option("prefilter", 2)
    singles = [t.knn(q, k) for q in qs]                         # two-stage, one query at a time
    for u in range(16):
        assert np.array_equal(singles[u][0], want[0][u]) and np.array_equal(singles[u][1].view(np.uint32), want[1][u].view(np.uint32))
    got = t.knn(qs, k)                                          # 16 queries: ONE group on the matrix pipe (8 + 8 on the vector ALU)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    cand, fell_back = t.prefilter_stats()
    assert not fell_back and cand >= min(k, 1)
    d_q = torch.from_numpy(qs).cuda()
    d_i = torch.empty((16, k), dtype=torch.int64, device="cuda")
    d_d = torch.empty((16, k), dtype=torch.float32, device="cuda")
    for stage1 in (1, 0):                                       # the group's stage 1 on the matrix pipe (default), then on the vector ALU
        t.set_option("batch_stage1", stage1)
        got = t.knn(qs, k)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), stage1
    t.set_option("batch_stage1", 1)
    for nq in (2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 15, 16):    # every group size: 1 to 4 column blocks of digit columns
        d_i.zero_(); d_d.zero_()
        t.knn_device(d_q.data_ptr(), nq, k, d_i.data_ptr(), d_d.data_ptr(), torch.cuda.current_stream().cuda_stream, batched=True)
        torch.cuda.synchronize()
        assert np.array_equal(d_i.cpu().numpy()[:nq].view(np.uint64), want[0][:nq])
        assert np.array_equal(d_d.cpu().numpy()[:nq].view(np.uint32), want[1][:nq].view(np.uint32))
    t.close()


@pytest.mark.parametrize("k", [3, 64, 1000])
def test_a_group_with_degenerate_queries_over_a_corpus_with_odd_rows(built, k):
    """What the single-query tests cover, inside GROUPS: rows with a NaN, an infinity, elements beyond bf16's range, a zero
    row, tiny and huge multiples of a query, a dominant element, 3 000 exact duplicates (ties by id) in the corpus; a zero
    query, a NaN query, an infinite query, a stored row and a huge-magnitude query among the group's queries.  Groups of 16,
    8, 5 and 2, stage 1 on the matrix pipe and on the vector ALU: ids and distance bits of the single pass."""
    rng = np.random.default_rng(170 + k)
    t = EmbeddingTable(DIM, 0)
    t.insert_synthetic(22, 0, N)
    qs = rng.standard_normal((16, DIM)).astype(np.float32)
    q = qs[0]
    odd = rng.standard_normal((7, DIM)).astype(np.float32)
    odd[0, 5] = np.nan
    odd[1, 9] = np.inf
    odd[2] *= np.float32(3.2e38) / np.abs(odd[2]).max()
    odd[3] = 0.0
    odd[4] = q * np.float32(1e-17)
    odd[5] = q * np.float32(1e19)
    odd[6] = q
    odd[6, 3] = 1e6
    t.insert(odd)
    t.insert(np.repeat(q[None, :] * np.float32(0.5), 3000, 0))
    qs[1] = 0.0
    qs[2, 17] = np.nan
    qs[3, 3] = np.inf
    qs[4] = t.rows(4321, 1)[0]
    qs[5] *= np.float32(1e30)                                   # finite, squares overflow fp32
    qs[6] *= np.float32(1e-30)
    want = t.knn(qs, k)                                         # prefilter off: the single pass
    t.set_option("prefilter", 2)
    for stage1 in (1, 0):
        t.set_option("batch_stage1", stage1)
        for lo, nq in ((0, 16), (0, 8), (1, 5), (2, 2), (5, 2)):
            got = t.knn(qs[lo:lo + nq], k)
            assert np.array_equal(got[0], want[0][lo:lo + nq]), (stage1, lo, nq)
            assert np.array_equal(got[1].view(np.uint32), want[1][lo:lo + nq].view(np.uint32)), (stage1, lo, nq)
    t.close()


@pytest.mark.parametrize("k", [10, 1000])
def test_a_group_in_which_some_queries_fall_back_and_others_do_not(built, k):
    """The queries of a group share every launch, the gated single pass included: query y of the group runs it iff ITS OWN
    fallback word is set.  4.3 M duplicates of one row + 300 k random rows; queries beside the duplicated row collect more
    than 2^22 candidates (fallback), random queries a few hundred (no fallback) — mixed inside one group of 8, a group of
    4 and a group of 2, every answer equal to the single pass bit for bit."""
    import torch
    gen = torch.Generator(device="cuda"); gen.manual_seed(12)
    base = torch.randn((DIM,), device="cuda", generator=gen)
    t = EmbeddingTable(DIM, 0)
    t.reserve(4_600_000)
    t.insert_synthetic(33, 0, N)
    x = base[None, :].repeat(100_000, 1).contiguous()
    torch.cuda.synchronize()   # (insert_device below runs on the handle's own stream: x must be complete first)
    for i in range(43):
        t.insert_device(x.data_ptr(), 100_000, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    near = lambda: (base + 0.01 * torch.randn((DIM,), device="cuda", generator=gen)).cpu().numpy()
    rng = np.random.default_rng(5)
    far = lambda: rng.standard_normal(DIM).astype(np.float32)
    qs = np.stack([far(), near(), far(), far(), near(), near(), far(), near(),      # a group of 8
                   near(), far(), far(), near(),                                    # a group of 4
                   far(), near()])                                                  # a group of 2
    t.set_option("prefilter", 0)
    want = [t.knn(q, k) for q in qs]
    t.set_option("prefilter", 2)
    t.set_option("prefilter_adaptive", 0)            # every query tries stage 1 (the adaptive rule would switch it off here)
    fell = []
    for q in qs:                                     # which queries fall back, one at a time
        _same(t.knn(q, k), want[len(fell)])
        fell.append(bool(t.prefilter_stats()[1]))
    assert fell == [False, True, False, False, True, True, False, True, True, False, False, True, False, True], fell
    got = t.knn(qs, k)                               # 14 = 8 + 4 + 2: three groups
    for u in range(len(qs)):
        assert np.array_equal(got[0][u], want[u][0]), u
        assert np.array_equal(got[1][u].view(np.uint32), want[u][1].view(np.uint32)), u
    t.close()


@pytest.mark.parametrize("k", [1, 10, 64])
def test_the_sampled_threshold_gives_the_same_answers_even_when_the_sample_misleads(built, k):
    """"prefilter_sample" (round 4): for k <= 64 the collect threshold is the k-th smallest upper bound of every 8th tile's keys
    — valid because the k-th smallest of a subset is never below the k-th smallest of the whole, looser when the sample is
    unrepresentative.  Here it is as unrepresentative as it gets: every row of a sampled tile (64-row tiles for one query,
    16-row tiles for a group: rows whose index is 0..63 mod 512 cover both) points AWAY from the queries, every near row sits in
    a tile the sample never sees.  Same ids and distance bits as the single pass, for one query and for a group, with the
    option on and off; with it on, stage 2 re-evaluates more rows."""
    rng = np.random.default_rng(77 + k)
    n = N + 777
    qs = rng.standard_normal((5, DIM)).astype(np.float32)
    rows = rng.standard_normal((n, DIM)).astype(np.float32)
    idx = np.arange(n)
    sampled = (idx % 512) < 64
    rows[sampled] = -qs[0] + 0.3 * rows[sampled]                     # far from query 0 (and from its neighbours)
    near = np.flatnonzero(~sampled)[rng.permutation((~sampled).sum())[:200]]
    rows[near] = qs[0] + 0.2 * rows[near]                             # the true neighbours of query 0: never in the sample
    t = EmbeddingTable(DIM, 0)
    t.insert(rows)
    want = t.knn(qs, k)                                               # prefilter off
    t.set_option("prefilter", 2)
    cands = {}
    for flag in (0, 2):                                               # 2: single queries sample too (1, the default: groups only)
        t.set_option("prefilter_sample", flag)
        got1 = t.knn(qs[0], k)
        cands[flag], fell_back = t.prefilter_stats()
        assert not fell_back
        assert np.array_equal(got1[0], want[0][0]) and np.array_equal(got1[1].view(np.uint32), want[1][0].view(np.uint32)), flag
        got = t.knn(qs, k)                                            # a group of 5
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32)), flag
    assert set(want[0][0].tolist()) <= set(near.tolist())             # the answer is made of rows the sample never saw
    assert cands[2] >= cands[0] >= k
    t.close()
