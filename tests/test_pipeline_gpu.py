"""BASELINE config 4 as written, on a real MI355X and through the C ABI: batch = 256 bf16 ViT-L/14
embed -> rows appended to the table on the device -> top-k over table + appended rows, fused on HIP
streams (mi_pipeline_*; reference flow server/src/clip.rs:107-137 + server/src/search.rs:70-86).

Checked against the oracle (oracle/vit_numpy.py, oracle/oracle.c) at the benchmarked shape:
  images {0, 127, 128, 255} of the batch (both edges of the two 128-image half-chunk streams)
    bf16 rows in the table : |y - ref| <= 3e-2 * rms(ref)   (the stated bf16 bound, tests/test_vit_gpu.py)
    fp32 HIP path          : |y - ref| <= 1e-4 * (|ref| + rms(ref))
  appended rows            : what mi_knn_get_rows returns == what mi_clip_embed returns, bit for bit
  query                    : ids and distance bits == the oracle on (table rows ++ appended rows)
and the cross-stream ordering the handles promise (embed_device -> append_device -> search on
different streams without a host synchronisation in between).
"""
import os

import numpy as np
import pytest

from image_search_amd import synth
from image_search_amd.clip import PRECISION_BF16, PRECISION_F32, Model
from image_search_amd.search import EmbeddingTable, PinnedBuffer, Pipeline
from oracle import vit_numpy
from oracle.binding import orc_knn

pytestmark = pytest.mark.gpu

SEL = [0, 127, 128, 255]
N0 = 300_000  # rows already in the table: oracle-sized, and above the 2^18 rows from which the two-stage search applies
              # (the 10 M case is in test_knn_gpu.py::test_full_size_10m_matches_oracle_and_properties)


def close(out, ref, tol):
    rms = float(np.sqrt((np.asarray(ref, np.float64) ** 2).mean()))
    return np.allclose(out, ref, rtol=tol, atol=tol * rms), float(np.abs(out - ref).max() / rms)


@pytest.fixture(scope="module")
def l14_batch(built, tmp_path_factory):
    cfg = synth.VitConfig.vit_l14()
    w = synth.vit_weights(cfg, 0)
    path = str(tmp_path_factory.mktemp("w") / "l14.safetensors")
    synth.save_safetensors(w, path, {"num_attention_heads": cfg.heads})
    px = synth.preprocess_rgb8(synth.images_u8(4242, 256, cfg.image))
    ref = vit_numpy.vit_forward(w, cfg, px[SEL], np.float32)
    return cfg, path, px, ref


@pytest.mark.parametrize("prefilter", [0, 2])
def test_config4_batch256_bf16_embed_append_query(l14_batch, orc, prefilter):
    """prefilter = 2 is the query mode bench.py runs (the two-stage exact search over the byte mirror): the same
    oracle answers, and the two stages — not the fallback — must have produced them."""
    cfg, path, px, ref = l14_batch
    # the fp32 HIP path on the same four images: the parity path at 1e-4
    m32 = Model.from_file(path, 0, PRECISION_F32)
    out32 = m32.forward(px[SEL])
    m32.close()
    ok, err = close(out32, ref, 1e-4)
    assert ok, err

    m = Model.from_file(path, 0, PRECISION_BF16)
    t = EmbeddingTable(768, 0)
    t.reserve(N0 + 3 * 256)
    t.insert_synthetic(7, 0, N0)
    if prefilter:
        t.set_option("prefilter", prefilter)
    base_rows = synth.corpus_rows(7, 0, N0)
    pipe = Pipeline(m, t)
    pin = [PinnedBuffer(px.shape), PinnedBuffer(px.shape)]
    pin[0].array[:] = px
    pin[1].array[:40] = px[:40]

    q_rand = synth.corpus_rows(8, 0, 1)[0]
    q_img = out32[2]                                  # image 128's fp32 embedding: nearest row must be its bf16 twin
    before = pipe.query(q_rand, 10)                   # enqueued before any ingest: sees the N0 base rows only
    assert pipe.ingest(pin[0].array) == N0            # chunk 1: 256 images, two half-chunk streams
    after1 = pipe.query(q_rand, 10)
    twin = pipe.query(q_img, 10)
    assert pipe.ingest(pin[1].array[:40]) == N0 + 256  # chunk 2: other buffer, other size
    assert pipe.ingest(pin[0].array) == N0 + 296      # chunk 3: buffer 0 again (its upload waited for chunk 1's forward)
    after3 = pipe.query(q_img, 1000)                  # the reference's K
    assert pipe.ingest(pin[0].array[:0]) == N0 + 552  # n = 0: no-op (clip.rs:112-118)
    pipe.sync()
    assert len(t) == N0 + 552
    cand, fell_back = t.prefilter_stats()             # of the last query (k = 1000)
    if prefilter:
        assert cand >= 1000 and not fell_back, (cand, fell_back)
    else:
        assert cand == 0 and not fell_back

    rows = t.rows(N0, 552)
    ok, err = close(rows[SEL], ref, 3e-2)             # bf16 bound at the benchmarked shape, vs the oracle
    assert ok, err
    direct = m.forward(px)                            # the host entry point: same kernels, same half-chunks
    assert np.array_equal(rows[:256].view(np.uint32), direct.view(np.uint32))
    assert np.array_equal(rows[296:552].view(np.uint32), direct.view(np.uint32))
    assert np.array_equal(rows[256:296].view(np.uint32), m.forward(px[:40]).view(np.uint32))
    err16 = float(np.abs(rows[SEL] - ref).max() / np.sqrt((ref.astype(np.float64) ** 2).mean()))
    print(f"config 4, b=256 bf16 vs oracle on images {SEL}: max|err|/rms = {err16:.2e}")

    def same(got, q, table_rows, k):
        oi, od = orc_knn(orc, q, table_rows, k)
        assert np.array_equal(got[0], oi), (got[0][:5], oi[:5])
        assert np.array_equal(got[1].view(np.uint32), od.view(np.uint32))

    same(before, q_rand, base_rows, 10)
    all1 = np.concatenate([base_rows, rows[:256]])
    same(after1, q_rand, all1, 10)
    same(twin, q_img, all1, 10)
    assert int(twin[0][0]) == N0 + 128 and twin[1][0] < 1e-3
    same(after3, q_img, np.concatenate([base_rows, rows]), 1000)
    assert {int(i) for i in after3[0][:2]} == {N0 + 128, N0 + 296 + 128}   # the image was ingested twice
    pipe.close()
    for b in pin:
        b.close()
    t.close()
    m.close()


@pytest.mark.parametrize("prec", [PRECISION_BF16, PRECISION_F32])
def test_the_reference_default_chunk_of_500_images(l14_batch, prec):
    """`-c/--chunk-size` defaults to 500 (server/src/server_arguments.rs:12-13) and a chunk is ONE forward call
    (server/src/clip.rs:72-73, :112-118); the library's passes hold at most max_batch = 256 images, so the default call runs
    as 256 + 244 (a ragged second pass of the full geometry), and 257 images as 256 + 1 (a second pass of one image, below
    the two-stream threshold).  Every row must carry the bits of the pass it would get alone — through the host entry
    point, through Pipeline.ingest with a 500-image pinned buffer, and with a forward's front overlapped with the previous
    forward (option "front_overlap")."""
    cfg, path, px, ref = l14_batch
    px500 = np.concatenate([px, px[:244][::-1]])
    m = Model.from_file(path, 0, prec)
    first, second = m.forward(px500[:256]), m.forward(px500[256:])
    want = np.concatenate([first, second])
    assert np.isfinite(want).all()
    ok, err = close(first[SEL], ref, 1e-4 if prec == PRECISION_F32 else 3e-2)
    assert ok, err
    whole = m.forward(px500)
    assert np.array_equal(whole.view(np.uint32), want.view(np.uint32))
    one_more = m.forward(px500[:257])
    assert np.array_equal(one_more.view(np.uint32), np.concatenate([first, m.forward(px500[256:257])]).view(np.uint32))
    t = EmbeddingTable(768, 0)
    t.reserve(2000)
    pipe = Pipeline(m, t)
    pin = PinnedBuffer(px500.shape)
    pin.array[:] = px500
    for overlap in (0, 1):
        m.set_option("front_overlap", overlap)
        at = len(t)
        assert pipe.ingest(pin.array) == at               # 256 + 244 inside one call
        assert pipe.ingest(pin.array[:257]) == at + 500   # 256 + 1
        pipe.sync()
        rows = t.rows(at, 757)
        assert np.array_equal(rows[:500].view(np.uint32), want.view(np.uint32)), overlap
        assert np.array_equal(rows[500:].view(np.uint32), one_more.view(np.uint32)), overlap
        assert np.array_equal(m.forward(px500).view(np.uint32), want.view(np.uint32)), overlap   # mi_clip_embed: two chunks, its own copy stream
    pipe.close()
    pin.close()
    t.close()
    m.close()


def test_handles_order_work_across_streams(l14_batch, orc):
    """embed_device(stream A) -> append_device(stream A) -> search on ANOTHER stream / the handle's own,
    no host synchronisation in between: the search must scan the appended rows (the handle
    keeps an event per kind of work and every entry point waits for it on its own stream)."""
    import torch
    cfg, path, px, ref = l14_batch
    m = Model.from_file(path, 0, PRECISION_BF16)
    t = EmbeddingTable(768, 0)
    t.reserve(60_000)
    t.insert_synthetic(9, 0, 50_000)
    n = 64
    d_img = torch.from_numpy(px[:n]).cuda()
    d_emb = torch.empty((n, 768), dtype=torch.float32, device="cuda")
    d_q = torch.empty((768,), dtype=torch.float32, device="cuda")
    d_idx = torch.empty((10,), dtype=torch.int64, device="cuda")
    d_dist = torch.empty((10,), dtype=torch.float32, device="cuda")
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    m.forward_device(d_img.data_ptr(), n, d_emb.data_ptr(), a.cuda_stream)
    t.insert_device(d_emb.data_ptr(), n, a.cuda_stream)
    with torch.cuda.stream(a):
        d_q.copy_(d_emb[17])                          # the query is one of the rows being appended
        ev = torch.cuda.Event()
        ev.record(a)
    b.wait_event(ev)                                  # the caller orders its OWN buffer (d_q); the table orders itself
    t.knn_device(d_q.data_ptr(), 1, 10, d_idx.data_ptr(), d_dist.data_ptr(), b.cuda_stream)
    second = m.forward_device(d_img.data_ptr(), n, d_emb.data_ptr(), b.cuda_stream)  # same workspace, other stream
    torch.cuda.synchronize()
    assert second is None
    emb = d_emb.cpu().numpy()
    idx = d_idx.cpu().numpy().view(np.uint64)
    assert int(idx[0]) == 50_000 + 17 and float(d_dist[0]) <= 1e-6
    rows = t.rows(50_000, n)
    assert np.array_equal(rows, emb)                  # both forwards give the same bits; the append saw the first
    oi, od = orc_knn(orc, emb[17], np.concatenate([synth.corpus_rows(9, 0, 50_000), rows]), 10)
    assert np.array_equal(idx, oi)
    # the host entry point on the handle's own stream right after an async append on stream a
    t.insert_device(d_emb.data_ptr(), n, a.cuda_stream)
    gi, gd = t.knn(emb[5], 3)
    assert {int(gi[0]), int(gi[1])} == {50_005, 50_000 + n + 5}
    t.close()
    m.close()


def test_pipeline_argument_errors(built, tmp_path):
    import ctypes
    from image_search_amd._lib import MiError, c_vp, lib
    cfg = synth.VitConfig.tiny()
    path = str(tmp_path / "tiny.safetensors")
    synth.save_safetensors(synth.vit_weights(cfg, 1), path, {"num_attention_heads": cfg.heads})
    m = Model.from_file(path, 0, PRECISION_F32)
    t768 = EmbeddingTable(768, 0)
    with pytest.raises(MiError) as e:
        Pipeline(m, t768)                             # tiny model embeds into 64 dims
    assert e.value.code == -1
    t = EmbeddingTable(64, 0)
    pipe = Pipeline(m, t)
    px = synth.preprocess_rgb8(synth.images_u8(3, 5, cfg.image))
    assert pipe.ingest(px) == 0                       # pageable memory works too
    i, d = pipe.query(np.ones(64, np.float32), 8)
    pipe.sync()
    assert np.array_equal(t.rows(0, 5), m.forward(px))
    assert list(i[5:]) == [0xFFFFFFFFFFFFFFFF] * 3 and np.isinf(d[5:]).all()   # fewer than k rows
    h = c_vp()
    assert lib().mi_pipeline_create(None, t._h, ctypes.byref(h)) == -1
    assert lib().mi_pipeline_query(pipe._h, None, 3, i.ctypes.data, d.ctypes.data) == -1
    pipe.close()
    t.close()
    t768.close()
    m.close()


def test_bf16_retrieval_agrees_with_fp32_and_outlier_channels_do_not_matter(built, tmp_path):
    """What the bf16 tower costs downstream (profiles/r02_bf16_acceptance.json holds the 4096-image run of
    tools/bf16_acceptance.py): the same structured images indexed from fp32 and from bf16 embeddings, held-out
    queries embedded in the index's own precision.  Pinned here on 1024 + 200 images (50 queries made the top-1 line a
    coin: one flipped near-tie is 2 %; the 1000-query runs of profiles/r06_bf16_acceptance.json read 99.1-99.6 %):
      * seeded weights: top-10 / top-100 id sets agree >= 99 %, the nearest image is the same for >= 97 %;
      * channels 50x above the rest planted WITHOUT changing the function (LayerNorm gain x50, the matching
        q/k/v/fc1 input columns /50): nothing changes — a float keeps its relative precision at any magnitude;
      * the same channels planted so that the function changes (attention logits ~10x: sharply peaked softmax):
        the bf16 error doubles; MI_PRECISION_BF16_SPLIT (LayerNorm outputs as hi + lo halves) recovers part of it."""
    from image_search_amd.clip import PRECISION_BF16_SPLIT
    cfg = synth.VitConfig.vit_l14()
    w = synth.vit_weights(cfg, 0)
    px_i = synth.preprocess_rgb8(synth.scenes_u8(5, 1024, cfg.image))
    px_q = synth.preprocess_rgb8(synth.scenes_u8(6, 200, cfg.image))

    def run(weights, prec):
        path = str(tmp_path / "w.safetensors")
        synth.save_safetensors(weights, path, {"num_attention_heads": cfg.heads})
        m = Model.from_file(path, 0, prec)
        out = [np.concatenate([m.forward(x[i:i + 256]) for i in range(0, len(x), 256)]) for x in (px_i, px_q)]
        m.close()
        return out

    def agreement(ref, got, ks=(1, 10, 100)):
        res = {}
        tabs = []
        for e, q in (ref, got):
            t = EmbeddingTable(768, 0)
            t.insert(e)
            tabs.append(t.knn(q, max(ks))[0])
            t.close()
        for k in ks:
            res[k] = float(np.mean([len(set(tabs[0][u, :k].tolist()) & set(tabs[1][u, :k].tolist())) / k for u in range(len(ref[1]))]))
        rms = float(np.sqrt((ref[0].astype(np.float64) ** 2).mean()))
        return float(np.abs(got[0] - ref[0]).max() / rms), res

    report = {}
    for name, weights in (("generated", w), ("outliers, function preserved", synth.plant_outlier_channels(w, compensate=True)),
                          ("outliers, function changed", synth.plant_outlier_channels(w))):
        ref = run(weights, PRECISION_F32)
        report[name] = {"bf16": agreement(ref, run(weights, PRECISION_BF16))}
        if name != "generated":
            report[name]["bf16_split"] = agreement(ref, run(weights, PRECISION_BF16_SPLIT))
    print(report)
    for name in ("generated", "outliers, function preserved"):
        err, agree = report[name]["bf16"]
        assert err <= 3e-2, (name, err)
        assert agree[1] >= 0.97 and agree[10] >= 0.99 and agree[100] >= 0.99, (name, agree)
    err16, _ = report["outliers, function changed"]["bf16"]
    errsp, agree_sp = report["outliers, function changed"]["bf16_split"]
    assert err16 <= 6e-2 and errsp <= err16 * 1.05 and agree_sp[10] >= 0.97, report["outliers, function changed"]


def test_sharded_pipeline_one_process_replicas_feed_their_own_shards(built, tmp_path, orc):
    """mi_pipeline_create_sharded (BASELINE config 5 as the reference's ONE server process would run it,
    server/src/main.rs:30-35 + server/src/clip.rs:112-137): a tower replica per shard, every run of a chunk embedded on
    the GPU that owns its block and written straight into that shard, queries over all shards with the device-side
    exchange.  One GPU here, so the three shards (and replicas) share device 0; rows, ids and results must be those of
    the one-GPU pipeline over one table, and of the oracle."""
    from image_search_amd.search import ShardedTable
    cfg = synth.VitConfig.tiny()
    path = str(tmp_path / "tiny.safetensors")
    synth.save_safetensors(synth.vit_weights(cfg, 1), path, {"num_attention_heads": cfg.heads})
    models = [Model.from_file(path, 0, PRECISION_F32) for _ in range(2)]
    px = synth.preprocess_rgb8(synth.images_u8(21, 300, cfg.image))
    want_rows = models[0].forward(px)

    base = synth.corpus_rows(23, 0, 1000, 64)
    st = ShardedTable(64, [0, 0, 0], 64)                       # blocks of 64 rows: a 300-image chunk spans all replicas
    st.insert(base)
    pipe = Pipeline([models[0], models[1], models[0]], st)    # a replica may serve several shards of its GPU
    one = EmbeddingTable(64, 0)
    one.insert(base)
    ref = Pipeline(models[1], one)

    qs = synth.corpus_rows(24, 0, 4, 64)
    got, want = [], []
    for p_, out in ((pipe, got), (ref, want)):
        out.append(p_.query(qs[0], 10))                        # before any ingest: the base rows only
        assert p_.ingest(px[:100]) == 1000
        out.append(p_.query(qs[1], 10))
        assert p_.ingest(px[100:101]) == 1100                  # a single image, mid-block
        assert p_.ingest(px[101:]) == 1101                     # 199 images: crosses three block boundaries
        out.append(p_.query(want_rows[150], 5))                # an ingested row itself: distance ~0 at its own id
        out.append(p_.query(qs[2], 1000))
        assert p_.ingest(px[:0]) == 1300                       # n = 0 (clip.rs:112-118)
        p_.sync()
    assert len(st) == 1300 and sum(st.shard_rows(s) for s in range(3)) == 1300
    assert np.array_equal(st.rows(1000, 300).view(np.uint32), want_rows.view(np.uint32))    # same kernels, same bits, in order
    all_rows = np.concatenate([base, want_rows])
    for (gi, gd), (wi, wd) in zip(got, want):
        gi, gd = gi.reshape(-1), gd.reshape(-1)
        assert np.array_equal(gi, wi) and np.array_equal(gd.view(np.uint32), wd.view(np.uint32))
    assert int(got[2][0].reshape(-1)[0]) == 1150
    oi, od = orc_knn(orc, qs[2], all_rows, 1000)
    assert np.array_equal(got[3][0].reshape(-1), oi) and np.array_equal(got[3][1].reshape(-1).view(np.uint32), od.view(np.uint32))
    n_f, ms_f, _, _ = pipe.stats()
    assert n_f >= 6 and ms_f > 0                               # forwards of every lane are timed
    pipe.close(); ref.close(); st.close(); one.close()
    for m in models:
        m.close()
