"""The reference server runs its scan task and its search handlers side by side in one process, each behind a mutex
(server/src/main.rs:30-35, server/src/search.rs:26, :109-118).  A drop-in is called the same way: several threads on the
same handles, at once.  Every handle serialises internally (include/mi355clip.h, Conventions) and the lock order is fixed
(pipeline -> sharded table -> model, shard), so nothing may deadlock, fail or return a torn result — whatever interleaving
the threads produce, every answer must be the exact answer for SOME prefix of the ingested chunks."""
import threading
import time

import numpy as np
import pytest

from image_search_amd import synth
from image_search_amd.clip import PRECISION_F32, Model
from image_search_amd.search import EmbeddingTable, Pipeline, ShardedTable
from oracle.binding import orc_knn

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _setup(tmp_path):
    cfg = synth.VitConfig.tiny()
    path = str(tmp_path / "tiny.safetensors")
    synth.save_safetensors(synth.vit_weights(cfg, 1), path, {"num_attention_heads": cfg.heads})
    m = Model.from_file(path, 0, PRECISION_F32)
    px = synth.preprocess_rgb8(synth.images_u8(31, 40, cfg.image))
    return cfg, path, m, px


def _run(threads):
    errors = []

    def guard(fn):
        def wrapped():
            try:
                fn()
            except Exception as e:  # noqa: BLE001 — collected and re-raised by the test thread
                errors.append(repr(e))
        return wrapped

    ts = [threading.Thread(target=guard(fn)) for fn in threads]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in ts), "a thread is stuck: deadlock between handles"
    assert not errors, errors


def _check_answers(answers, q, base, emb, chunk, n_chunks, k, orc):
    """every (ids, dists) seen during the run is the oracle's answer for the table after j chunks, for some j"""
    valid = {}
    for j in range(n_chunks + 1):
        rows = np.concatenate([base] + [emb] * j) if j else base
        oi, od = orc_knn(orc, q, rows, k)
        valid[(oi.tobytes(), od.tobytes())] = j
    seen = set()
    for gi, gd in answers:
        key = (np.ascontiguousarray(gi).tobytes(), np.ascontiguousarray(gd).tobytes())
        assert key in valid, "an answer that matches no prefix of the ingested chunks"
        seen.add(valid[key])
    return seen


def test_scan_task_and_search_handlers_on_one_gpu(built, tmp_path, orc):
    cfg, path, m, px = _setup(tmp_path)
    emb = m.forward(px)
    base = synth.corpus_rows(33, 0, 5000, 64)
    t = EmbeddingTable(64, 0)
    t.reserve(5000 + 40 * 25)
    t.insert(base)
    pipe = Pipeline(m, t)
    q = (emb[7] + 0.01 * synth.corpus_rows(34, 0, 1, 64)[0]).astype(np.float32)   # next to an ingested row: the answer changes with every chunk
    n_chunks, k = 25, 10
    answers, lock, stop = [], threading.Lock(), threading.Event()

    def scan_task():
        for _ in range(n_chunks):
            pipe.ingest(px)
            time.sleep(0.002)
        pipe.sync()
        stop.set()

    def handler_host():                                   # mi_knn_search on the table the pipeline is feeding
        while not stop.is_set():
            r = t.knn(q, k)
            with lock:
                answers.append(r)

    def handler_pipeline():                               # queries through the pipeline itself, delivered by drain
        while not stop.is_set():
            r = pipe.query(q, k)
            pipe.drain(0)
            with lock:
                answers.append((r[0].copy(), r[1].copy()))

    def reader():                                         # the refine step's row fetch (search.rs:43-58)
        while not stop.is_set():
            assert np.array_equal(t.rows(100, 50), base[100:150])

    _run([scan_task, handler_host, handler_host, handler_pipeline, reader])
    assert len(t) == 5000 + 40 * n_chunks
    assert np.array_equal(t.rows(5000 + 40 * (n_chunks - 1), 40).view(np.uint32), emb.view(np.uint32))
    seen = _check_answers(answers, q, base, emb, 40, n_chunks, k, orc)
    assert len(seen) >= 2, "the handlers never saw the table change: the test did not overlap anything"
    print(f"{len(answers)} concurrent answers, tables after {sorted(seen)[:3]}..{sorted(seen)[-3:]} chunks seen")
    pipe.close(); t.close(); m.close()


def test_scan_task_and_search_handlers_over_a_sharded_table(built, tmp_path, orc):
    cfg, path, m, px = _setup(tmp_path)
    emb = m.forward(px)
    base = synth.corpus_rows(35, 0, 6000, 64)
    st = ShardedTable(64, [0, 0, 0], 64)
    st.reserve(6000 + 40 * 20)
    st.insert(base)
    pipe = Pipeline([m, m, m], st)
    q = (emb[11] + 0.01 * synth.corpus_rows(36, 0, 1, 64)[0]).astype(np.float32)
    n_chunks, k = 20, 10
    answers, lock, stop = [], threading.Lock(), threading.Event()

    def scan_task():
        for _ in range(n_chunks):
            pipe.ingest(px)
            time.sleep(0.002)
        pipe.sync()
        stop.set()

    def handler_sync():
        while not stop.is_set():
            r = st.knn(q, k)
            with lock:
                answers.append(r)

    def handler_async():                                  # several searches in flight, delivered together
        while not stop.is_set():
            pend = [st.knn_async(q, k) for _ in range(3)]
            st.sync()
            with lock:
                answers.extend((i[0].copy(), d[0].copy()) for i, d in pend)

    def handler_pipeline():
        while not stop.is_set():
            r = pipe.query(q, k)
            pipe.drain(0)
            with lock:
                answers.append((r[0].copy(), r[1].copy()))

    _run([scan_task, handler_sync, handler_async, handler_pipeline])
    assert len(st) == 6000 + 40 * n_chunks
    assert np.array_equal(st.rows(6000, 40).view(np.uint32), emb.view(np.uint32))
    _check_answers(answers, q, base, emb, 40, n_chunks, k, orc)
    pipe.close(); st.close(); m.close()
