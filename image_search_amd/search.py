"""Host-side mirror of the reference's query/refine loop and kNN statement
(/root/reference/server/src/search.rs) on top of the C ABI.

`average_slices` and `refine_query` keep the reference's names, argument meaning
and error behaviour (search.rs:127-150, :60-67); `EmbeddingTable` stands where the
SurrealDB table `image` + index `mt_pts` stood (clip.rs:135-143) and its
`knn()` is the `embedding <|K|> $reference` statement (search.rs:70-86).
`ShardedTable` is the multi-GPU form: one process per GPU, rows split
contiguously, per-shard top-k all-gathered (RCCL over xGMI via torch.distributed)
and merged identically on every rank.
"""
from __future__ import annotations

import ctypes
from typing import Sequence

import numpy as np

from ._lib import c_f, c_vp, check, lib

K_REFERENCE = 1000  # `<|1000|>` in server/src/search.rs:76
NO_ID = np.uint64(0xFFFFFFFFFFFFFFFF)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptrs(vecs):
    arr = (c_f * len(vecs))()
    for i, v in enumerate(vecs):
        arr[i] = v.ctypes.data_as(c_f)
    return arr


def average_slices(vectors: Sequence[np.ndarray]) -> np.ndarray:
    """fn average_slices(vectors: &Vec<&Vec<f32>>) -> Vec<f32>  (search.rs:127-150).
    Panics in the reference on empty input / ragged lengths; raises here."""
    if len(vectors) == 0:
        raise AssertionError("Input must not be empty")
    vecs = [_f32(v).reshape(-1) for v in vectors]
    n = vecs[0].size
    if any(v.size != n for v in vecs):
        raise AssertionError("All vectors must have the same length")
    out = np.empty(n, np.float32)
    check(lib().mi_average_slices(_ptrs(vecs), len(vecs), n, out.ctypes.data_as(c_f)))
    return out


def refine_query(text_embedding: np.ndarray, selected: Sequence[np.ndarray]) -> np.ndarray:
    """search.rs:28,:60-67 — no marked image found: the text vector itself; else
    average_slices([average_slices(selected), text])."""
    text = _f32(text_embedding).reshape(-1)
    sel = [_f32(v).reshape(-1) for v in selected]
    if any(v.size != text.size for v in sel):
        raise AssertionError("All vectors must have the same length")
    out = np.empty_like(text)
    check(lib().mi_refine(text.ctypes.data_as(c_f), _ptrs(sel), len(sel), text.size, out.ctypes.data_as(c_f)))
    return out


class EmbeddingTable:
    """One row-shard of `image.embedding` resident in HBM (mi_knn)."""

    def __init__(self, dim: int = 768, device: int = 0, base: int = 0):
        self._h = c_vp()
        self.dim, self.device = dim, device
        check(lib().mi_knn_create(dim, device, ctypes.byref(self._h)))
        if base:
            self.set_base(base)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mi_knn_free(self._h)
            self._h = c_vp()

    __del__ = close

    def __len__(self) -> int:
        n = ctypes.c_uint64()
        check(lib().mi_knn_size(self._h, ctypes.byref(n)))
        return n.value

    def set_base(self, base: int):
        check(lib().mi_knn_set_base(self._h, base))

    def reserve(self, rows: int):
        check(lib().mi_knn_reserve(self._h, rows))

    def set_option(self, key: str, value: int):
        """mi_knn_set_option: "prefilter" = 1 turns on the two-stage exact search (bf16 mirror, + 50 % memory)."""
        check(lib().mi_knn_set_option(self._h, key.encode(), int(value)))

    def prefilter_stats(self):
        """(rows re-evaluated by stage 2, fell back to the single pass) of the most recent single-query search"""
        c, f = ctypes.c_uint32(), ctypes.c_uint32()
        check(lib().mi_knn_prefilter_stats(self._h, ctypes.byref(c), ctypes.byref(f)))
        return c.value, bool(f.value)

    def prefilter_state(self):
        """mi_knn_prefilter_state: the two-stage search's view of itself"""
        out = (ctypes.c_uint32 * 4)()
        check(lib().mi_knn_prefilter_state(self._h, out))
        return {"skips_left": out[0], "consecutive_fallbacks": out[1], "searches_skipped": out[2], "scales_taken_at_rows": out[3]}

    def insert(self, embeddings: np.ndarray):
        """db.insert("image").content(rows) (clip.rs:125-137): ids are insertion ordinals."""
        e = _f32(embeddings).reshape(-1, self.dim)
        check(lib().mi_knn_append(self._h, e.ctypes.data, e.shape[0]))

    def insert_device(self, d_ptr: int, n: int, stream: int = 0):
        check(lib().mi_knn_append_device(self._h, d_ptr, n, stream))

    def insert_synthetic(self, seed: int, first_row: int, n: int):
        check(lib().mi_knn_append_synthetic(self._h, seed, first_row, n))

    def rows(self, first: int, n: int) -> np.ndarray:
        out = np.empty((n, self.dim), np.float32)
        check(lib().mi_knn_get_rows(self._h, first, n, out.ctypes.data))
        return out

    def knn(self, reference: np.ndarray, k: int = K_REFERENCE):
        """`WHERE embedding <|k|> $reference` (search.rs:70-77): (ids, cosine distances),
        ascending distance then id.  reference: [dim] or [nq,dim]."""
        q = _f32(reference)
        single = q.ndim == 1
        q = q.reshape(-1, self.dim)
        idx = np.empty((q.shape[0], k), np.uint64)
        dist = np.empty((q.shape[0], k), np.float32)
        check(lib().mi_knn_search(self._h, q.ctypes.data, q.shape[0], k, idx.ctypes.data, dist.ctypes.data))
        return (idx[0], dist[0]) if single else (idx, dist)

    def knn_device(self, d_q: int, nq: int, k: int, d_idx: int, d_dist: int, stream: int = 0, batched: bool = False):
        fn = lib().mi_knn_search_batched_device if batched else lib().mi_knn_search_device
        check(fn(self._h, d_q, nq, k, d_idx, d_dist, stream))


class PinnedBuffer:
    """Page-locked host memory (mi_host_alloc) viewed as a numpy array: upload buffers of the
    fused pipeline (asynchronous H2D needs pinned memory)."""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(x) for x in shape)
        self.dtype = np.dtype(dtype)
        nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self._p = c_vp()
        check(lib().mi_host_alloc(nbytes, ctypes.byref(self._p)))
        buf = (ctypes.c_char * nbytes).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=self.dtype).reshape(self.shape)

    def close(self):
        if getattr(self, "_p", None) and self._p.value:
            self.array = None
            lib().mi_host_free(self._p)
            self._p = c_vp()

    __del__ = close


class Pipeline:
    """The scan-loop body (clip.rs:107-137) and the query (search.rs:70-86) fused on HIP streams
    (mi_pipeline_*): `ingest` uploads, embeds and inserts a chunk without a readback; `query`
    scans on a second stream under the next chunk's forward; `sync` delivers the results."""

    def __init__(self, model, table):
        """model: one image tower and table: an EmbeddingTable on its GPU — or, for ONE process over several GPUs
        (mi_pipeline_create_sharded), a list of towers, models[s] on the device of shard s of a ShardedTable."""
        self._h = c_vp()
        self.model, self.table = model, table  # borrowed: keep them alive
        self._pending = []
        if isinstance(table, ShardedTable):
            models = list(model) if isinstance(model, (list, tuple)) else [model] * table.info()["shards"]
            arr = (c_vp * len(models))(*[m._h for m in models])
            check(lib().mi_pipeline_create_sharded(arr, len(models), table._h, ctypes.byref(self._h)))
        else:
            check(lib().mi_pipeline_create(model._h, table._h, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mi_pipeline_free(self._h)
            self._h = c_vp()
            self._pending = []

    __del__ = close

    def ingest(self, nchw: np.ndarray) -> int:
        """nchw: [n,3,H,W] f32 (a PinnedBuffer.array makes the upload asynchronous; the array must stay
        untouched until the next ingest / sync returns).  Returns the id of the chunk's first row."""
        x = nchw if (nchw.dtype == np.float32 and nchw.flags.c_contiguous) else np.ascontiguousarray(nchw, np.float32)
        first = ctypes.c_uint64()
        check(lib().mi_pipeline_ingest(self._h, x.ctypes.data, x.shape[0], ctypes.byref(first)))
        return first.value

    def query(self, reference: np.ndarray, k: int = K_REFERENCE):
        """Enqueue `embedding <|k|> $reference`; returns (ids, distances) arrays that are filled by sync()."""
        q = _f32(reference).reshape(-1)
        idx = np.full(k, NO_ID, np.uint64)
        dist = np.full(k, np.inf, np.float32)
        check(lib().mi_pipeline_query(self._h, q.ctypes.data, k, idx.ctypes.data, dist.ctypes.data))
        self._pending.append((idx, dist))  # the library writes into them at sync: keep them alive
        return idx, dist

    def query_device(self, reference: np.ndarray, k: int, d_idx: int, d_dist: int, consumer_stream: int = 0):
        """The same query, its k results left on the device at d_idx [k] uint64 / d_dist [k] f32 (mi_pipeline_query_device);
        `consumer_stream` (a raw hipStream_t) is made to wait for them: the list a sharded search feeds to the all-gather."""
        q = _f32(reference).reshape(-1)
        check(lib().mi_pipeline_query_device(self._h, q.ctypes.data, k, d_idx, d_dist, consumer_stream))

    def sync(self):
        check(lib().mi_pipeline_sync(self._h))
        self._pending = []

    def drain(self, leave_pending: int = 0):
        """Deliver finished queries (oldest first) until at most `leave_pending` remain pending."""
        check(lib().mi_pipeline_drain(self._h, leave_pending))
        if leave_pending == 0:
            self._pending = []
        else:
            self._pending = self._pending[-leave_pending:]

    def stats(self, reset: bool = False):
        """(forwards, ms in forwards, scans, ms in scans) measured by events on the pipeline's streams."""
        out = (ctypes.c_double * 4)()
        check(lib().mi_pipeline_stats(self._h, out, 1 if reset else 0))
        return tuple(out)


def merge_candidates(idx_lists: np.ndarray, dist_lists: np.ndarray, k: int):
    """Global top-k of `lists` per-shard candidate lists (same ordering rule)."""
    i = np.ascontiguousarray(idx_lists, np.uint64).reshape(-1)
    d = np.ascontiguousarray(dist_lists, np.float32).reshape(-1)
    assert i.size == d.size and i.size % k == 0
    idx = np.empty(k, np.uint64)
    dist = np.empty(k, np.float32)
    check(lib().mi_knn_merge(i.ctypes.data, d.ctypes.data, i.size // k, k, idx.ctypes.data, dist.ctypes.data))
    return idx, dist


def merge_candidates_device(device: int, d_idx_in: int, d_dist_in: int, lists: int, nq: int, k: int, d_idx: int, d_dist: int,
                            stream: int = 0):
    """mi_knn_merge_device: the merge of the all-gathered [lists][nq][k] lists on the device, asynchronous on `stream`."""
    check(lib().mi_knn_merge_device(device, d_idx_in, d_dist_in, lists, nq, k, d_idx, d_dist, stream))


class ShardedTable:
    """`image.embedding` row-sharded over the GPUs of one node inside ONE process (mi_knn_sharded_*): the
    reference's one-handle, one-search-at-a-time shape (main.rs:30-35, search.rs:26) for BASELINE config 5.
    Same `insert` / `knn` surface as EmbeddingTable; ids are global insertion ordinals."""

    TRANSPORTS = {0: "single shard", 1: "device copies", 2: "rccl all-gather"}

    def __init__(self, dim: int = 768, devices: Sequence[int] = (0,), block_rows: int = 0):
        self._h = c_vp()
        self.dim = dim
        devs = (ctypes.c_int * len(devices))(*devices)
        check(lib().mi_knn_sharded_create(dim, devs, len(devices), block_rows, ctypes.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mi_knn_sharded_free(self._h)
            self._h = c_vp()

    __del__ = close

    def info(self):
        rows, n, blk, tr = ctypes.c_uint64(), ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_int()
        check(lib().mi_knn_sharded_info(self._h, ctypes.byref(rows), ctypes.byref(n), ctypes.byref(blk), ctypes.byref(tr)))
        return {"rows": rows.value, "shards": n.value, "block_rows": blk.value, "transport": self.TRANSPORTS[tr.value]}

    def stats(self):
        """what the exchange step has executed so far (mi_knn_sharded_stats)"""
        out = (ctypes.c_uint64 * 4)()
        check(lib().mi_knn_sharded_stats(self._h, out))
        return dict(zip(("searches", "collectives", "copies", "merges"), (int(v) for v in out)))

    def __len__(self) -> int:
        return self.info()["rows"]

    def set_option(self, key: str, value: int):
        check(lib().mi_knn_sharded_set_option(self._h, key.encode(), int(value)))

    def reserve(self, rows: int):
        check(lib().mi_knn_sharded_reserve(self._h, rows))

    def insert(self, embeddings: np.ndarray) -> int:
        e = _f32(embeddings).reshape(-1, self.dim)
        first = ctypes.c_uint64()
        check(lib().mi_knn_sharded_append(self._h, e.ctypes.data, e.shape[0], ctypes.byref(first)))
        return first.value

    def insert_synthetic(self, seed: int, first_row: int, n: int):
        check(lib().mi_knn_sharded_append_synthetic(self._h, seed, first_row, n))

    def insert_device(self, d_ptr: int, n: int, src_device: int = 0, stream: int = 0) -> int:
        """rows already in device memory on `src_device`: routed to their shards device to device (mi_knn_sharded_append_device)"""
        first = ctypes.c_uint64()
        check(lib().mi_knn_sharded_append_device(self._h, d_ptr, n, src_device, stream, ctypes.byref(first)))
        return first.value

    def shard_rows(self, s: int) -> int:
        h = lib().mi_knn_sharded_shard(self._h, s)
        n = ctypes.c_uint64()
        check(lib().mi_knn_size(c_vp(h), ctypes.byref(n)))
        return n.value

    def rows(self, first: int, n: int) -> np.ndarray:
        out = np.empty((n, self.dim), np.float32)
        check(lib().mi_knn_sharded_get_rows(self._h, first, n, out.ctypes.data))
        return out

    def knn_async(self, reference: np.ndarray, k: int = K_REFERENCE):
        """mi_knn_sharded_search_async: returns (ids, distances) arrays that are filled by sync()."""
        q = _f32(reference).reshape(-1, self.dim)
        idx = np.full((q.shape[0], k), NO_ID, np.uint64)
        dist = np.full((q.shape[0], k), np.inf, np.float32)
        check(lib().mi_knn_sharded_search_async(self._h, q.ctypes.data, q.shape[0], k, idx.ctypes.data, dist.ctypes.data))
        self._pending = getattr(self, "_pending", []) + [(idx, dist)]
        return idx, dist

    def sync(self):
        check(lib().mi_knn_sharded_sync(self._h))
        self._pending = []

    def rebalance_from(self, src: "ShardedTable"):
        """every row of `src` into this empty table, device to device (mi_knn_sharded_rebalance)"""
        check(lib().mi_knn_sharded_rebalance(self._h, src._h))

    def knn(self, reference: np.ndarray, k: int = K_REFERENCE):
        q = _f32(reference)
        single = q.ndim == 1
        q = q.reshape(-1, self.dim)
        idx = np.empty((q.shape[0], k), np.uint64)
        dist = np.empty((q.shape[0], k), np.float32)
        check(lib().mi_knn_sharded_search(self._h, q.ctypes.data, q.shape[0], k, idx.ctypes.data, dist.ctypes.data))
        return (idx[0], dist[0]) if single else (idx, dist)

    def save(self, prefix: str):
        check(lib().mi_knn_sharded_save(self._h, prefix.encode()))

    def load(self, prefix: str):
        check(lib().mi_knn_sharded_load(self._h, prefix.encode()))


def shard_bounds(n_rows: int, world: int, rank: int):
    """Contiguous row split: rank r owns [r*N/W, (r+1)*N/W)."""
    return (n_rows * rank) // world, (n_rows * (rank + 1)) // world


def gather_and_merge(local_idx: np.ndarray, local_dist: np.ndarray, k: int, group=None):
    """The one exchange step of the sharded search: all-gather every rank's k
    candidates (12*k bytes per rank and query) and merge them identically on
    every rank.  Works on any torch.distributed backend (nccl == RCCL on ROCm,
    gloo on CPU); tensors live where the backend needs them."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    nq = local_idx.shape[0] if local_idx.ndim == 2 else 1
    li = torch.from_numpy(np.ascontiguousarray(local_idx, np.uint64).view(np.int64).reshape(nq, k))
    ld = torch.from_numpy(np.ascontiguousarray(local_dist, np.float32).reshape(nq, k))
    if dist.get_backend(group) == "nccl":
        li, ld = li.cuda(), ld.cuda()
    gi = torch.empty((world * nq, k), dtype=li.dtype, device=li.device)  # rank-major concatenation
    gd = torch.empty((world * nq, k), dtype=ld.dtype, device=ld.device)
    dist.all_gather_into_tensor(gi, li, group=group)
    dist.all_gather_into_tensor(gd, ld, group=group)
    gi = gi.cpu().numpy().view(np.uint64).reshape(world, nq, k)
    gd = gd.cpu().numpy().reshape(world, nq, k)
    out_i = np.empty((nq, k), np.uint64)
    out_d = np.empty((nq, k), np.float32)
    for u in range(nq):
        out_i[u], out_d[u] = merge_candidates(gi[:, u, :], gd[:, u, :], k)
    return out_i, out_d


class ShardExchange:
    """The one exchange step of the row-sharded search, one process per GPU: every rank's k results
    — packed as k uint64 ids followed by k f32 distances, 12 k bytes — all-gathered in ONE collective and merged
    identically on every rank (mi_knn_merge).
      nccl (= RCCL over xGMI): the scan writes its list into a device buffer (Pipeline.query_device), the collective and the
          readback of the gathered lists are enqueued behind it on torch's current stream; the host never touches the
          per-rank list and blocks only in `collect`, for the oldest exchange.
      gloo (CPU rehearsals and tests): the lists come from the host results of Pipeline.query / EmbeddingTable.knn.
    `depth` exchanges may be in flight (a ring of buffers)."""

    def __init__(self, k: int, device=None, group=None, depth: int = 4):
        import torch
        import torch.distributed as dist
        self.k, self.group, self.depth = k, group, depth
        self.world = dist.get_world_size(group)
        self.on_device = dist.get_backend(group) == "nccl"
        self.nbytes = 12 * k
        dev = (device if device is not None else torch.device("cuda", torch.cuda.current_device())) if self.on_device else "cpu"
        self.ring = []
        for _ in range(depth):
            slot = {"loc": torch.empty((1, self.nbytes), dtype=torch.uint8, device=dev),
                    "all": torch.empty((self.world, self.nbytes), dtype=torch.uint8, device=dev), "busy": False}
            if self.on_device:
                slot["host"] = torch.empty((self.world, self.nbytes), dtype=torch.uint8).pin_memory()
                slot["ev"] = torch.cuda.Event()
            self.ring.append(slot)
        self.n = 0
        self.pending = []

    def _slot(self):
        slot = self.ring[self.n % self.depth]
        self.n += 1
        if slot["busy"]:
            raise RuntimeError(f"more than {self.depth} exchanges in flight: collect() first")
        slot["busy"] = True
        self.pending.append(slot)
        return slot

    def _gather(self, slot):
        import torch.distributed as dist
        dist.all_gather_into_tensor(slot["all"], slot["loc"], group=self.group)

    def query(self, pipeline: "Pipeline", reference: np.ndarray):
        """nccl only: enqueue the scan on the pipeline's search stream, the all-gather and the readback behind it."""
        import torch
        assert self.on_device, "ShardExchange.query needs the nccl backend; use submit() with host lists on gloo"
        slot = self._slot()
        p = slot["loc"].data_ptr()
        pipeline.query_device(reference, self.k, p, p + 8 * self.k, torch.cuda.current_stream().cuda_stream)
        self._gather(slot)
        slot["host"].copy_(slot["all"], non_blocking=True)
        slot["ev"].record()

    def submit(self, local_idx: np.ndarray, local_dist: np.ndarray):
        """A rank's list already on the host (gloo; or nccl after a host search): pack, all-gather."""
        import torch
        slot = self._slot()
        packed = np.concatenate([np.ascontiguousarray(local_idx, np.uint64).reshape(self.k).view(np.uint8),
                                 np.ascontiguousarray(local_dist, np.float32).reshape(self.k).view(np.uint8)])
        slot["loc"].copy_(torch.from_numpy(packed).reshape(1, -1))
        self._gather(slot)
        if self.on_device:
            slot["host"].copy_(slot["all"], non_blocking=True)
            slot["ev"].record()

    def collect(self):
        """Oldest exchange in flight -> (ids [k], distances [k]) of the whole table; blocks for that one only."""
        slot = self.pending.pop(0)
        if self.on_device:
            slot["ev"].synchronize()
            h = slot["host"].numpy()
        else:
            h = slot["all"].numpy()
        gi = np.ascontiguousarray(h[:, :8 * self.k]).view(np.uint64)
        gd = np.ascontiguousarray(h[:, 8 * self.k:]).view(np.float32)
        slot["busy"] = False
        return merge_candidates(gi, gd, self.k)

    def in_flight(self) -> int:
        return len(self.pending)


def _cstrs(strings):
    arr = (ctypes.c_char_p * max(len(strings), 1))()
    for i, p in enumerate(strings):
        arr[i] = p.encode()
    return arr


class ImageIndex:
    """Table `image` {id, image_path, embedding} (search.rs:13-18; rows inserted at clip.rs:125-137): the C ABI's
    mi_index_* — one HBM shard plus the image_path column, both inside the library.  Row id = insertion ordinal.
      `SELECT image_path FROM image WHERE image_path IN $paths`              -> existing()
      `db.insert("image").content(rows)`                                     -> insert()
      `SELECT id, image_path, embedding FROM image WHERE image_path IN $p`   -> embeddings_of()
      `SELECT id, image_path, knn() FROM image WHERE embedding <|K|> $ref`   -> web_search_text()
    kept across restarts by save / load (what the database did for the reference)."""

    def __init__(self, dim: int = 768, device: int = 0, media_dir: str = ""):
        self._h = c_vp()
        self.dim, self.device, self.media_dir = dim, device, media_dir
        check(lib().mi_index_create(dim, device, media_dir.encode(), ctypes.byref(self._h)))
        self.table = EmbeddingTable.__new__(EmbeddingTable)          # the shard inside the index, borrowed
        self.table._h, self.table.dim, self.table.device = c_vp(lib().mi_index_table(self._h)), dim, device
        self.table.close = lambda: None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.table._h = c_vp()
            lib().mi_index_free(self._h)
            self._h = c_vp()

    __del__ = close

    def __len__(self) -> int:
        n = ctypes.c_uint64()
        check(lib().mi_index_size(self._h, ctypes.byref(n)))
        return n.value

    def path(self, row: int, web: bool = False) -> str:
        need = ctypes.c_size_t()
        check(lib().mi_index_path(self._h, row, int(web), None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        check(lib().mi_index_path(self._h, row, int(web), buf, need.value, None))
        return buf.value.decode()

    @property
    def paths(self) -> list:
        return [self.path(i) for i in range(len(self))]

    def existing(self, paths: Sequence[str]) -> set:
        """clip.rs:74-83: which of `paths` already have a row."""
        paths = list(paths)
        flags = (ctypes.c_uint8 * max(len(paths), 1))()
        check(lib().mi_index_existing(self._h, _cstrs(paths), len(paths), flags))
        return {p for p, f in zip(paths, flags) if f}

    def insert(self, paths: Sequence[str], embeddings: np.ndarray) -> int:
        """clip.rs:125-137: one row per (image_path, embedding) pair, ids in insertion order.  Like the
        reference's table there is no uniqueness constraint; the scan loop filters first."""
        paths = list(paths)
        e = _f32(embeddings).reshape(-1, self.dim)
        if e.shape[0] != len(paths):
            raise ValueError(f"{len(paths)} paths for {e.shape[0]} embeddings")
        first = ctypes.c_uint64()
        check(lib().mi_index_insert(self._h, _cstrs(paths), e.ctypes.data, len(paths), ctypes.byref(first)))
        return first.value

    def adopt(self, paths: Sequence[str]):
        """Paths for rows the fused pipeline has just written into `self.table` (mi_index_adopt)."""
        paths = list(paths)
        check(lib().mi_index_adopt(self._h, _cstrs(paths), len(paths)))

    def embeddings_of(self, paths: Sequence[str]):
        """search.rs:43-58.  Rows come back in table (id) order, whatever the request order — the
        order matters: average_slices adds in input order (search.rs:139-143)."""
        paths = list(paths)
        cnt = ctypes.c_size_t()
        check(lib().mi_index_rows_of(self._h, _cstrs(paths), len(paths), None, 0, ctypes.byref(cnt)))
        ids = np.empty(cnt.value, np.uint64)
        check(lib().mi_index_rows_of(self._h, _cstrs(paths), len(paths), ids.ctypes.data, cnt.value, ctypes.byref(cnt)))
        rows = [int(i) for i in ids]
        return rows, [self.table.rows(r, 1)[0] for r in rows]

    def web_search_text(self, text_embedding: np.ndarray, referenced_images: Sequence[str] = (), k: int = K_REFERENCE):
        """search.rs:20-110 after the text tower: refine with the marked images that are in the
        table, K nearest by cosine distance, paths mapped back under `media/`.
        Returns [(id, image_path, similarity)] with similarity = vector::distance::knn()."""
        q = _f32(text_embedding).reshape(-1)
        refs = list(referenced_images)
        idx = np.empty(k, np.uint64)
        dist = np.empty(k, np.float32)
        n = ctypes.c_uint32()
        check(lib().mi_index_search(self._h, q.ctypes.data, _cstrs(refs), len(refs), k, idx.ctypes.data, dist.ctypes.data,
                                    ctypes.byref(n)))
        return [(int(idx[i]), self.path(int(idx[i]), web=True), float(dist[i])) for i in range(n.value)]

    def save(self, directory: str):
        check(lib().mi_index_save(self._h, directory.encode()))

    @classmethod
    def load(cls, directory: str, device: int = 0, dim: int = 768) -> "ImageIndex":
        ix = cls(dim, device, "")
        check(lib().mi_index_load(ix._h, directory.encode()))
        need = ctypes.c_size_t()
        check(lib().mi_index_media_dir(ix._h, None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        check(lib().mi_index_media_dir(ix._h, buf, need.value, None))
        ix.media_dir = buf.value.decode()
        return ix


def decode_rgb8(path: str) -> np.ndarray:
    """`image::open(path)...to_rgb8()` with Pillow as the decoder: RGB8 [H,W,3].  High-bit-depth images (16-bit PNG /
    TIFF, modes I;16 / I / F) are SCALED to 8 bits the way the image crate converts sample types (>> 8), not clipped at
    255 as Pillow's convert("RGB") does — those photos would otherwise embed as almost white."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode in ("I;16", "I;16B", "I;16L", "I"):
            a = np.asarray(im).astype(np.uint32)
            a = (a >> 8).clip(0, 255).astype(np.uint8) if a.max(initial=0) > 255 else a.astype(np.uint8)
            return np.repeat(a[..., None], 3, axis=2)
        if im.mode == "F":
            a = np.asarray(im, np.float32)
            return np.repeat((np.clip(a, 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)[..., None], 3, axis=2)
        return np.asarray(im.convert("RGB"), np.uint8)


def embed_all_images_in_dir(model, index: ImageIndex, media_dir: str, image_chunk_size: int = 500, decode=None,
                            shuffle_seed=None) -> int:
    """clip.rs:42-151: walk `media_dir` (following links), keep the allow-listed extensions, shuffle,
    and per chunk: skip paths that already have a row, decode, embed (resize + normalise + tower on
    the device: Model.forward_images), insert.  A crash loses at most one chunk; a rerun resumes.
    `decode(path) -> RGB8 [H,W,3]` defaults to Pillow; files that fail to decode are logged and
    skipped TOGETHER WITH their path (the reference zips the unfiltered path list with the
    surviving embeddings, clip.rs:125-134, which shifts paths after a failure).  Returns rows added."""
    import logging
    import os
    import random
    from .clip import is_image_path
    if decode is None:
        decode = decode_rgb8
    paths = []
    seen_dirs = set()
    for root, dirs, files in os.walk(media_dir, followlinks=True):
        # WalkDir::follow_links detects cycles; os.walk does not: never descend into a directory twice
        st = os.stat(root)
        seen_dirs.add((st.st_dev, st.st_ino))
        keep = []
        for d in dirs:
            try:
                sd = os.stat(os.path.join(root, d))
            except OSError:
                continue
            if (sd.st_dev, sd.st_ino) not in seen_dirs:
                seen_dirs.add((sd.st_dev, sd.st_ino))
                keep.append(d)
        dirs[:] = keep
        for name in files:
            p = os.path.join(root, name)
            if os.path.isfile(p) and is_image_path(p):
                paths.append(p)
    random.Random(shuffle_seed).shuffle(paths)
    added = 0
    for c0 in range(0, len(paths), image_chunk_size):
        chunk = paths[c0:c0 + image_chunk_size]
        have = index.existing(chunk)
        new_paths, images = [], []
        for p in chunk:
            if p in have:
                continue
            try:
                images.append(decode(p))
                new_paths.append(p)
            except Exception as err:  # noqa: BLE001 — mirrors `Failed to open image` (clip.rs:98-101)
                logging.getLogger(__name__).error("Failed to open image %s: %s", p, err)
        if new_paths:
            index.insert(new_paths, model.forward_images(images))
            added += len(new_paths)
    return added
