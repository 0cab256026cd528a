// sharded.hip — the embedding table row-sharded over the GPUs of one node, inside ONE process
// (mi_knn_sharded_*, include/mi355clip.h).
//
// The reference server is one process with one database handle (server/src/main.rs:30-35) and serves one
// search at a time under a mutex (server/src/search.rs:26); a drop-in for `embedding <|K|> $reference`
// (server/src/search.rs:70-86) over more rows than one GPU holds must therefore shard BEHIND the handle:
//   * rows: block-cyclic — global row r (the insertion ordinal = the id) lives in block r / B, blocks are
//     dealt round-robin to the n shards, each shard stores its blocks back to back.  Appending in order keeps
//     every shard contiguous, ids stay global ordinals, and a shard's local order is its id order, so the
//     per-shard top-k is already sorted by (distance, id) (knn_kernels.h: IdMap).
//   * search: the query goes to every device, every shard scans on its own stream (the scans run
//     concurrently), the n lists of k (id, distance) — 12 k bytes per shard and query — meet by an RCCL
//     all-gather over xGMI (ncclAllGather under ncclGroupStart/End, one communicator per device from
//     ncclCommInitAll: single-process multi-GPU) and are merged ON THE DEVICE (knn_merge_lists_kernel on the
//     first shard's stream), then read back once.  Nothing in a search blocks the host until its results are
//     asked for (mi_knn_sharded_search_async / _sync; mi_knn_sharded_search = both).
//     RCCL is bound at run time (dlopen): a one-GPU table never needs it, and a process that already holds
//     PyTorch's bundled copy must not map a second one.  Where RCCL cannot run — the same device listed
//     twice (how the one-GPU test box exercises n > 1 shards), or no librccl — each shard's list is copied
//     into the first shard's gather buffer (device-to-device / peer copy on the shard's stream, an event per
//     shard); both transports feed the same merge kernel.
//   * ingest on the device: mi_knn_sharded_append_device routes runs of rows produced on ANY device to their
//     shard (same device: a device-to-device copy; another device: hipMemcpyPeerAsync over xGMI), on the
//     producer's stream, no host hop.  mi_knn_sharded_rebalance re-deals a live table into another shard
//     count / device set the same way.
//   * persistence: one file per shard per GENERATION plus a manifest naming the generation, written last; a
//     crash or an I/O error at any point leaves the previous generation complete on disk.
#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "handles.h"

using namespace mi;

namespace {

// ---- RCCL, bound at run time -------------------------------------------------------------------
enum { kNcclUint8 = 1 };  // ncclDataType_t: the packed records travel as bytes
struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.lib, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.lib, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.lib, "ncclGroupEnd");
        r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.AllGather && r.GetErrorString;
    });
    return r;
}
#define RCCL_CHECK(x)                                                                                         \
    do {                                                                                                      \
        const int rc_ = (x);                                                                                  \
        if (rc_ != 0) fail(MI_ERR_HIP, "%s failed: %s", #x, rccl().GetErrorString ? rccl().GetErrorString(rc_) : "?"); \
    } while (0)

void free_sharded(mi_knn_sharded* t) {
    if (!t) return;
    for (size_t s = 0; s < t->shard.size(); ++s) {
        (void)hipSetDevice(t->devices[s]);
        (void)hipDeviceSynchronize();
    }
    for (auto& sl : t->slots) {
        for (size_t s = 0; s < t->shard.size(); ++s) {
            (void)hipSetDevice(t->devices[s]);
            for (std::vector<DevBuf>* v : {&sl.d_q, &sl.d_rec, &sl.g_rec})
                if (s < v->size() && (*v)[s].p) (void)hipFree((*v)[s].p);
            if (s < sl.ev.size() && sl.ev[s]) (void)hipEventDestroy(sl.ev[s]);
        }
        if (!t->devices.empty()) (void)hipSetDevice(t->devices[0]);
        for (DevBuf* b : {&sl.m_rec})
            if (b->p) (void)hipFree(b->p);
        for (PinnedBuf2* b : {&sl.h_q, &sl.h_rec})
            if (b->p) (void)hipHostFree(b->p);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    for (hipEvent_t e : t->ev_src)
        if (e) (void)hipEventDestroy(e);
    for (size_t s = 0; s < t->shard.size(); ++s) {
        (void)hipSetDevice(t->devices[s]);
        if (s < t->comms.size() && t->comms[s] && rccl().ok) (void)rccl().CommDestroy(t->comms[s]);
        mi_knn_free(t->shard[s]);
    }
    delete t;
}

// run fn(lo, n, shard, local) over the maximal runs of [first, first + n) that are contiguous inside one shard
template <class F>
void for_runs(const mi_knn_sharded* t, uint64_t first, uint64_t n, F&& fn) {
    uint64_t r = first;
    const uint64_t end = first + n;
    while (r < end) {
        const uint64_t in_block = t->block - r % t->block;
        const uint64_t len = std::min<uint64_t>(in_block, end - r);
        uint32_t s; uint64_t local;
        sharded_place(t, r, &s, &local);
        fn(r, len, s, local);
        r += len;
    }
}

// a failed multi-shard append / load leaves some shards ahead of the table: forget those rows
void roll_back(mi_knn_sharded* t) {
    for (uint32_t s = 0; s < t->n(); ++s) knn_truncate(t->shard[s], sharded_rows_of(t, t->rows, s));
}

void grow_buf(DevBuf& b, size_t bytes, hipStream_t in_flight_on) {
    if (bytes <= b.cap) return;
    if (b.p && in_flight_on) HIP_CHECK(hipStreamSynchronize(in_flight_on));  // an older search of this slot may still use it
    b.reserve(bytes);
}

void deliver(ShardedSlot& sl) {
    if (!sl.busy) return;
    HIP_CHECK(hipEventSynchronize(sl.done));
    const size_t per = (size_t)sl.nq * sl.k;
    std::memcpy(sl.user_idx, sl.h_rec.p, per * 8);
    std::memcpy(sl.user_dist, (const char*)sl.h_rec.p + per * 8, per * 4);
    sl.busy = false;
}

// rows [t->rows, t->rows + n) from device memory on `src_device` (t->mu held); copies enqueued on `st`, a stream of that device
void append_device_locked(mi_knn_sharded* t, const float* d_rows, uint64_t n, int src_device, hipStream_t st) {
    std::vector<char> touched(t->n(), 0);
    try {
        // Every shard that receives rows is taken to its FINAL size before any copy is enqueued: a reallocation in the
        // middle of the call would move (and free) a table that this call's earlier copies, still queued on the producer's
        // stream, are writing into — grow() only waits for the work recorded by EARLIER calls.
        for (uint32_t s = 0; s < t->n(); ++s) {
            mi_knn* sh = t->shard[s];
            const uint64_t want = sharded_rows_of(t, t->rows + n, s);
            if (want == sh->rows) continue;
            std::lock_guard<std::mutex> ls(sh->mu);
            DeviceGuard gs(sh->device);
            knn_grow(sh, want);  // a reallocation waits for the shard's work in flight (reserve ahead to avoid it)
        }
        for_runs(t, t->rows, n, [&](uint64_t r, uint64_t len, uint32_t s, uint64_t local) {
            mi_knn* sh = t->shard[s];
            std::lock_guard<std::mutex> ls(sh->mu);
            if (local != sh->rows) fail(MI_ERR_INVALID, "shard %u out of step (%llu rows, expected %llu)", s,
                                        (unsigned long long)sh->rows, (unsigned long long)local);
            if (sh->rows + len > sh->cap) fail(MI_ERR_INVALID, "shard %u was not grown for this append", s);
            DeviceGuard g(src_device);
            sh->writes.begin(st);
            const float* src = d_rows + (r - t->rows) * t->dim;
            float* dst = sh->table + sh->rows * t->dim;
            const size_t bytes = (size_t)len * t->dim * sizeof(float);
            if (sh->device == src_device) HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
            else HIP_CHECK(hipMemcpyPeerAsync(dst, sh->device, src, src_device, bytes, st));
            sh->rows += len;
            touched[s] = 1;
        });
        // a search of a shard must wait for these copies: same device -> the shard's write event is recorded on the
        // producer's stream; another device -> an event of the source device, waited for on the shard's own stream, which
        // then carries the shard's write event (events are recorded only on streams of their own device)
        hipEvent_t& es = t->ev_src[(size_t)src_device];
        bool recorded = false;
        for (uint32_t s = 0; s < t->n(); ++s) {
            if (!touched[s]) continue;
            mi_knn* sh = t->shard[s];
            std::lock_guard<std::mutex> ls(sh->mu);
            if (sh->device == src_device) {
                DeviceGuard g(src_device);
                sh->writes.end(st);
                continue;
            }
            if (!recorded) {
                DeviceGuard g(src_device);
                if (!es) HIP_CHECK(hipEventCreateWithFlags(&es, hipEventDisableTiming));
                HIP_CHECK(hipEventRecord(es, st));
                recorded = true;
            }
            DeviceGuard g(sh->device);
            hipStream_t ss = knn_own_stream(sh);
            HIP_CHECK(hipStreamWaitEvent(ss, es, 0));
            sh->writes.end(ss);
        }
    } catch (...) {
        roll_back(t);
        throw;
    }
    t->rows += n;
}

std::string shard_file(const std::string& prefix, uint64_t gen, unsigned s, unsigned n) {
    // generation 0 = the names written before generations existed
    return prefix + (gen ? ".g" + std::to_string(gen) : std::string()) + "." + std::to_string(s) + "of" + std::to_string(n) + ".miknn";
}

struct Manifest { unsigned n = 0, block = 0, dim = 0; unsigned long long rows = 0, gen = 0; bool ok = false; };
Manifest read_manifest(const std::string& prefix) {
    Manifest m;
    FILE* f = std::fopen((prefix + ".shards").c_str(), "r");
    if (!f) return m;
    const int got = std::fscanf(f, "%u %u %llu %u %llu", &m.n, &m.block, &m.rows, &m.dim, &m.gen);
    std::fclose(f);
    if (got == 4) m.gen = 0;
    m.ok = got >= 4 && m.n != 0 && m.block != 0;
    return m;
}

void fsync_dir_of(const std::string& path) {
    const size_t slash = path.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "." : (slash == 0 ? "/" : path.substr(0, slash));
    FILE* d = std::fopen(dir.c_str(), "r");
    if (d) { (void)fsync(fileno(d)); std::fclose(d); }
}

}  // namespace

namespace mi {

void sharded_place(const mi_knn_sharded* t, uint64_t r, uint32_t* s, uint64_t* local) {
    const uint64_t blk = r / t->block;
    *s = (uint32_t)(blk % t->n());
    *local = (blk / t->n()) * t->block + r % t->block;
}

uint64_t sharded_rows_of(const mi_knn_sharded* t, uint64_t total, uint32_t s) {
    const uint64_t full = total / t->block, rem = total % t->block, n = t->n();
    return (full / n + (s < full % n ? 1 : 0)) * t->block + (s == full % n ? rem : 0);
}

void sharded_deliver_all(mi_knn_sharded* t) {
    for (int i = 0; i < mi_knn_sharded::N_SLOTS; ++i) deliver(t->slots[(t->next_slot + i) % mi_knn_sharded::N_SLOTS]);  // oldest first
}

// Replaces `SELECT id, image_path, vector::distance::knn() FROM image WHERE embedding <|K|> $reference`
// (server/src/search.rs:70-86) over all shards; same results and ordering as ONE mi_knn holding every row.
// Enqueues everything and returns; the results reach idx / dist when the slot is delivered.
//
// A shard's answer is ONE packed record [nq*k x u64 id | nq*k x f32 distance], padded to 16 bytes: the scan writes both
// halves in place, the exchange moves it as one piece — one ncclAllGather of bytes per shard (or one copy) — the merge
// kernel reads the gathered records where they lie, and the merged record goes to the host in one copy.
void sharded_search_enqueue(mi_knn_sharded* t, const float* q, uint32_t nq, uint32_t k, uint64_t* idx, float* dist) {
    const uint32_t n = t->n();
    const size_t per = (size_t)nq * k;                        // results per shard
    const size_t rec = (per * 12 + 15) / 16 * 16;             // bytes of one shard's packed record
    auto rec_idx = [&](void* r) { return (uint64_t*)r; };
    auto rec_dist = [&](void* r) { return (float*)((char*)r + per * 8); };
    ShardedSlot& sl = t->slots[t->next_slot];
    t->next_slot = (t->next_slot + 1) % mi_knn_sharded::N_SLOTS;
    deliver(sl);  // the ring is full only when N_SLOTS searches are pending: finish the oldest
    if (sl.d_q.empty()) {
        sl.d_q.resize(n); sl.d_rec.resize(n); sl.g_rec.resize(n);
        sl.ev.assign(n, nullptr);
    }
    sl.nq = nq; sl.k = k; sl.user_idx = idx; sl.user_dist = dist;
    sl.h_q.reserve((size_t)nq * t->dim * 4);
    std::memcpy(sl.h_q.p, q, (size_t)nq * t->dim * 4);
    sl.h_rec.reserve(rec);
    // 1. every shard: query up, scan on the shard's own stream — the n scans run side by side
    for (uint32_t s = 0; s < n; ++s) {
        mi_knn* sh = t->shard[s];
        std::lock_guard<std::mutex> ls(sh->mu);
        DeviceGuard g(sh->device);
        hipStream_t st = knn_own_stream(sh);
        grow_buf(sl.d_q[s], (size_t)nq * t->dim * 4, st);
        grow_buf(sl.d_rec[s], rec, st);
        HIP_CHECK(hipMemcpyAsync(sl.d_q[s].p, sl.h_q.p, (size_t)nq * t->dim * 4, hipMemcpyHostToDevice, st));
        sh->writes.begin(st);
        sh->reads.begin(st);
        // several queries per call share their passes over the shard (groups of up to 16 through the two-stage search)
        knn_search_many(sh, (const float*)sl.d_q[s].p, nq, k, rec_idx(sl.d_rec[s].p), rec_dist(sl.d_rec[s].p), st);
        sh->reads.end(st);
    }
    mi_knn* first = t->shard[0];
    void* result = sl.d_rec[0].p;
    ++t->stats.searches;
    // A one-shard table gathers nothing — unless it was made with the RCCL transport (MI_KNN_SHARDED_TRANSPORT=rccl): then
    // its one-rank communicator runs the same collective and merge as n ranks do (how a one-GPU box exercises that code).
    if (n > 1 || t->use_rccl) {
        // 2. the one exchange step: 12 k bytes per shard and query, [shard] records on the first shard's device
        if (t->use_rccl) {
            for (uint32_t s = 0; s < n; ++s) {
                DeviceGuard g(t->shard[s]->device);
                grow_buf(sl.g_rec[s], rec * n, t->shard[s]->stream);
            }
            RCCL_CHECK(rccl().GroupStart());
            for (uint32_t s = 0; s < n; ++s) {
                RCCL_CHECK(rccl().AllGather(sl.d_rec[s].p, sl.g_rec[s].p, rec, kNcclUint8, t->comms[s], t->shard[s]->stream));
                ++t->stats.collectives;
            }
            RCCL_CHECK(rccl().GroupEnd());
        } else {
            {
                DeviceGuard g(first->device);
                grow_buf(sl.g_rec[0], rec * n, first->stream);
            }
            for (uint32_t s = 0; s < n; ++s) {
                mi_knn* sh = t->shard[s];
                DeviceGuard g(sh->device);
                char* dst = (char*)sl.g_rec[0].p + rec * s;
                if (sh->device == first->device) HIP_CHECK(hipMemcpyAsync(dst, sl.d_rec[s].p, rec, hipMemcpyDeviceToDevice, sh->stream));
                else HIP_CHECK(hipMemcpyPeerAsync(dst, first->device, sl.d_rec[s].p, sh->device, rec, sh->stream));
                ++t->stats.copies;
                if (s == 0) continue;
                if (!sl.ev[s]) HIP_CHECK(hipEventCreateWithFlags(&sl.ev[s], hipEventDisableTiming));
                HIP_CHECK(hipEventRecord(sl.ev[s], sh->stream));
            }
            DeviceGuard g(first->device);
            for (uint32_t s = 1; s < n; ++s) HIP_CHECK(hipStreamWaitEvent(first->stream, sl.ev[s], 0));
        }
        // 3. one merge per query over the n lists, on the first shard's device, reading the records where they lie
        DeviceGuard g(first->device);
        grow_buf(sl.m_rec, rec, first->stream);
        knn_merge_lists_device(rec_idx(sl.g_rec[0].p), rec_dist(sl.g_rec[0].p), n, nq, k, rec / 8, rec / 4, rec_idx(sl.m_rec.p),
                               rec_dist(sl.m_rec.p), first->stream);
        ++t->stats.merges;
        result = sl.m_rec.p;
    }
    DeviceGuard g(first->device);
    HIP_CHECK(hipMemcpyAsync(sl.h_rec.p, result, per * 12, hipMemcpyDeviceToHost, first->stream));
    if (!sl.done) HIP_CHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(sl.done, first->stream));
    sl.busy = true;
}

}  // namespace mi

extern "C" {

int mi_knn_sharded_create(uint32_t dim, const int* devices, int n_dev, uint32_t block_rows, mi_knn_sharded** out) {
    mi_knn_sharded* t = nullptr;
    const int rc = guarded([&] {
        if (!out) fail(MI_ERR_INVALID, "out is null");
        *out = nullptr;
        if (!devices || n_dev < 1 || n_dev > 64) fail(MI_ERR_INVALID, "n_dev must be 1..64 with a device list");
        if (block_rows == 0) block_rows = 4096;
        if (block_rows % 64 != 0) fail(MI_ERR_INVALID, "block_rows must be a multiple of 64 (a scan tile)");
        t = new mi_knn_sharded();
        t->dim = dim; t->block = block_rows;
        t->devices.assign(devices, devices + n_dev);
        for (int s = 0; s < n_dev; ++s) {
            mi_knn* h = nullptr;
            const int e = mi_knn_create(dim, devices[s], &h);
            if (e != MI_OK) fail(e, "%s", mi_last_error());
            h->cyc_block = block_rows; h->cyc_n = (uint32_t)n_dev; h->cyc_rank = (uint32_t)s;
            t->shard.push_back(h);
        }
        t->ev_src.assign((size_t)std::max(1, mi_device_count()), nullptr);
        // RCCL only between distinct devices (a communicator refuses one GPU twice); n_dev == 1 gathers nothing
        bool distinct = true;
        for (int a = 0; a < n_dev; ++a)
            for (int b = a + 1; b < n_dev; ++b) distinct = distinct && devices[a] != devices[b];
        // direct peer copies (append_device, rebalance, the copy transport) where the hardware allows; errors here only mean
        // the runtime stages such copies itself
        for (int a = 0; a < n_dev; ++a)
            for (int b = 0; b < n_dev; ++b) {
                if (devices[a] == devices[b]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                    DeviceGuard g(devices[a]);
                    (void)hipDeviceEnablePeerAccess(devices[b], 0);
                    (void)hipGetLastError();  // "already enabled" is not an error worth keeping
                }
            }
        const char* force = std::getenv("MI_KNN_SHARDED_TRANSPORT");  // "copy" (or "host", its old name) | "rccl": read once, at creation (tests)
        const std::string f = force ? force : "";
        const bool no_rccl = f == "copy" || f == "host";
        const bool want = f == "rccl" || (!no_rccl && n_dev > 1 && distinct);
        if (want) {
            if (!rccl().ok) {
                if (f == "rccl") fail(MI_ERR_UNSUPPORTED, "librccl.so not found: the all-gather transport is unavailable");
            } else if (distinct) {
                t->comms.assign(n_dev, nullptr);
                RCCL_CHECK(rccl().CommInitAll(t->comms.data(), n_dev, devices));
                t->use_rccl = true;
            } else if (f == "rccl") {
                fail(MI_ERR_INVALID, "RCCL needs distinct devices");
            }
        }
        *out = t;
    });
    if (rc != MI_OK && t) free_sharded(t);
    return rc;
}

void mi_knn_sharded_free(mi_knn_sharded* t) { free_sharded(t); }

int mi_knn_sharded_info(const mi_knn_sharded* t, uint64_t* rows, uint32_t* n_shards, uint32_t* block_rows, int* transport) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        if (rows) *rows = t->rows;
        if (n_shards) *n_shards = t->n();
        if (block_rows) *block_rows = t->block;
        if (transport) *transport = t->use_rccl ? 2 : (t->n() == 1 ? 0 : 1);
    });
}

// {searches enqueued, ncclAllGather calls issued (one per shard and search), transport copies, device merges}: lets a
// caller (and the tests) see which exchange code a search really ran
int mi_knn_sharded_stats(const mi_knn_sharded* t, uint64_t out[4]) {
    return guarded([&] {
        if (!t || !out) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(const_cast<mi_knn_sharded*>(t)->mu);
        out[0] = t->stats.searches; out[1] = t->stats.collectives; out[2] = t->stats.copies; out[3] = t->stats.merges;
    });
}

mi_knn* mi_knn_sharded_shard(mi_knn_sharded* t, uint32_t s) { return (t && s < t->n()) ? t->shard[s] : nullptr; }

int mi_knn_sharded_set_option(mi_knn_sharded* t, const char* key, int value) {
    return guarded([&] {
        if (!t || !key) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(t->mu);
        for (mi_knn* s : t->shard) {  // the options of mi_knn_set_option, applied to every shard
            const int e = mi_knn_set_option(s, key, value);
            if (e != MI_OK) fail(e, "%s", mi_last_error());
        }
    });
}

int mi_knn_sharded_reserve(mi_knn_sharded* t, uint64_t rows) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        std::lock_guard<std::mutex> l(t->mu);
        for (uint32_t s = 0; s < t->n(); ++s) {
            // whole blocks: the shard that owns the last, partial block must be able to fill it
            const uint64_t mine = (sharded_rows_of(t, rows, s) + t->block - 1) / t->block * t->block;
            const int e = mi_knn_reserve(t->shard[s], mine);
            if (e != MI_OK) fail(e, "%s", mi_last_error());
        }
    });
}

int mi_knn_sharded_append(mi_knn_sharded* t, const float* rows, uint64_t n, uint64_t* first_id) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        std::lock_guard<std::mutex> l(t->mu);
        if (first_id) *first_id = t->rows;
        if (n == 0) return;
        if (!rows) fail(MI_ERR_INVALID, "rows is null");
        try {
            for_runs(t, t->rows, n, [&](uint64_t r, uint64_t len, uint32_t s, uint64_t local) {
                if (local != t->shard[s]->rows) fail(MI_ERR_INVALID, "shard %u out of step (%llu rows, expected %llu)", s,
                                                     (unsigned long long)t->shard[s]->rows, (unsigned long long)local);
                const int e = mi_knn_append(t->shard[s], rows + (r - t->rows) * t->dim, len);
                if (e != MI_OK) fail(e, "%s", mi_last_error());
            });
        } catch (...) {
            roll_back(t);  // the shards that took their runs give them back: the table stays usable
            throw;
        }
        t->rows += n;
    });
}

// The insert of server/src/clip.rs:125-137 for embeddings that are already in device memory — on ANY device of the
// process: each run goes to its shard by a device-to-device copy (same GPU) or a peer copy over xGMI, enqueued on `stream`
// (a stream of src_device; NULL = that device's null stream), so the rows need no trip through the host and the call
// does not wait for the producer.  Searches enqueued afterwards see the rows (events, no host block).
int mi_knn_sharded_append_device(mi_knn_sharded* t, const float* d_rows, uint64_t n, int src_device, void* stream, uint64_t* first_id) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        std::lock_guard<std::mutex> l(t->mu);
        if (first_id) *first_id = t->rows;
        if (n == 0) return;
        if (!d_rows) fail(MI_ERR_INVALID, "d_rows is null");
        if (src_device < 0 || (size_t)src_device >= t->ev_src.size()) fail(MI_ERR_NO_DEVICE, "source device %d out of range", src_device);
        append_device_locked(t, d_rows, n, src_device, (hipStream_t)stream);
    });
}

int mi_knn_sharded_append_synthetic(mi_knn_sharded* t, uint64_t seed, uint64_t first_row, uint64_t n) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        std::lock_guard<std::mutex> l(t->mu);
        try {
            for_runs(t, t->rows, n, [&](uint64_t r, uint64_t len, uint32_t s, uint64_t) {
                const int e = mi_knn_append_synthetic(t->shard[s], seed, first_row + (r - t->rows), len);
                if (e != MI_OK) fail(e, "%s", mi_last_error());
            });
        } catch (...) {
            roll_back(t);
            throw;
        }
        t->rows += n;
    });
}

int mi_knn_sharded_get_rows(mi_knn_sharded* t, uint64_t first, uint64_t n, float* out) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        if (n == 0) return;
        if (!out) fail(MI_ERR_INVALID, "out is null");
        std::lock_guard<std::mutex> l(t->mu);
        if (first + n > t->rows) fail(MI_ERR_INVALID, "rows [%llu,%llu) out of range (size %llu)", (unsigned long long)first,
                                      (unsigned long long)(first + n), (unsigned long long)t->rows);
        for_runs(t, first, n, [&](uint64_t r, uint64_t len, uint32_t s, uint64_t local) {
            const int e = mi_knn_get_rows(t->shard[s], local, len, out + (r - first) * t->dim);
            if (e != MI_OK) fail(e, "%s", mi_last_error());
        });
    });
}

static void check_sharded_search(const mi_knn_sharded* t, const float* q, uint32_t nq, uint32_t k, const void* idx, const void* dist) {
    if (!t) fail(MI_ERR_INVALID, "null table handle");
    if (nq && (!q || !idx || !dist)) fail(MI_ERR_INVALID, "null query/result pointer");
    if (k == 0) fail(MI_ERR_INVALID, "k must be >= 1");
    if ((uint64_t)t->n() * k > 0xFFFFFFFFull) fail(MI_ERR_UNSUPPORTED, "shards * k too large");
}

int mi_knn_sharded_search(mi_knn_sharded* t, const float* q, uint32_t nq, uint32_t k, uint64_t* idx, float* dist) {
    return guarded([&] {
        check_sharded_search(t, q, nq, k, idx, dist);
        if (nq == 0) return;
        std::lock_guard<std::mutex> l(t->mu);
        sharded_search_enqueue(t, q, nq, k, idx, dist);
        sharded_deliver_all(t);
    });
}

int mi_knn_sharded_search_async(mi_knn_sharded* t, const float* q, uint32_t nq, uint32_t k, uint64_t* idx, float* dist) {
    return guarded([&] {
        check_sharded_search(t, q, nq, k, idx, dist);
        if (nq == 0) return;
        std::lock_guard<std::mutex> l(t->mu);
        sharded_search_enqueue(t, q, nq, k, idx, dist);
    });
}

int mi_knn_sharded_sync(mi_knn_sharded* t) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        std::lock_guard<std::mutex> l(t->mu);
        sharded_deliver_all(t);
    });
}

// Every row of `src` into the EMPTY `dst` (another shard count, device set or block size) without leaving the devices:
// src's blocks are walked in global order and each contiguous run is routed by append_device_locked — a device-to-device
// copy when source and destination shard share a GPU, hipMemcpyPeerAsync over xGMI otherwise.  src is unchanged.
int mi_knn_sharded_rebalance(mi_knn_sharded* dst, mi_knn_sharded* src) {
    return guarded([&] {
        if (!dst || !src || dst == src) fail(MI_ERR_INVALID, "two distinct table handles are needed");
        if (dst->dim != src->dim) fail(MI_ERR_INVALID, "dim %u into dim %u", src->dim, dst->dim);
        std::scoped_lock l(dst->mu, src->mu);
        if (dst->rows != 0) fail(MI_ERR_INVALID, "mi_knn_sharded_rebalance needs an empty destination");
        for (uint32_t s = 0; s < dst->n(); ++s) {
            const uint64_t mine = (sharded_rows_of(dst, src->rows, s) + dst->block - 1) / dst->block * dst->block;
            const int e = mi_knn_reserve(dst->shard[s], mine);  // one allocation per shard, nothing moves during the copy
            if (e != MI_OK) fail(e, "%s", mi_last_error());
        }
        try {
            for_runs(src, 0, src->rows, [&](uint64_t, uint64_t len, uint32_t s, uint64_t local) {
                mi_knn* from = src->shard[s];
                hipStream_t st;
                {
                    std::lock_guard<std::mutex> ls(from->mu);
                    DeviceGuard g(from->device);
                    st = knn_own_stream(from);
                    from->writes.begin(st);  // the rows being read have landed
                }
                append_device_locked(dst, from->table + local * src->dim, len, from->device, st);
            });
            for (mi_knn* from : src->shard) {  // the source may be freed as soon as this returns
                DeviceGuard g(from->device);
                if (from->stream) HIP_CHECK(hipStreamSynchronize(from->stream));
            }
        } catch (...) {
            dst->rows = 0;
            roll_back(dst);
            throw;
        }
    });
}

// Persistence: one MIKNNv01 file per shard and GENERATION, `<prefix>.g<gen>.<s>of<n>.miknn`, plus the manifest
// `<prefix>.shards` (text: n, block, rows, dim, gen) written last through a temporary + fsync + rename.  Until that rename
// the previous generation's files are untouched and still named by the previous manifest; afterwards they are deleted.
int mi_knn_sharded_save(mi_knn_sharded* t, const char* prefix) {
    return guarded([&] {
        if (!t || !prefix) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(t->mu);
        const std::string pre(prefix);
        const Manifest old = read_manifest(pre);
        const uint64_t gen = std::max<uint64_t>(t->generation, old.ok ? old.gen : 0) + 1;
        const char* stop = std::getenv("MI_KNN_SHARDED_SAVE_FAIL_AFTER");  // fault injection for the tests: fail after this many shard files
        const long fail_after = stop ? std::atol(stop) : -1;
        uint32_t written = 0;
        try {
            for (uint32_t s = 0; s < t->n(); ++s) {
                if (fail_after >= 0 && (long)s == fail_after) fail(MI_ERR_IO, "injected failure after %u shard files", s);
                const int e = mi_knn_save(t->shard[s], shard_file(pre, gen, s, t->n()).c_str());
                if (e != MI_OK) fail(e, "%s", mi_last_error());
                ++written;
            }
            const std::string meta = pre + ".shards", tmp = meta + ".tmp";
            FILE* f = std::fopen(tmp.c_str(), "w");
            if (!f) fail(MI_ERR_IO, "cannot create %s", tmp.c_str());
            const bool ok = std::fprintf(f, "%u %u %llu %u %llu\n", t->n(), t->block, (unsigned long long)t->rows, t->dim,
                                         (unsigned long long)gen) > 0 && std::fflush(f) == 0 && fsync(fileno(f)) == 0;
            if (std::fclose(f) != 0 || !ok) fail(MI_ERR_IO, "write to %s failed", tmp.c_str());
            if (std::rename(tmp.c_str(), meta.c_str()) != 0) fail(MI_ERR_IO, "cannot rename %s", tmp.c_str());  // the manifest last
            fsync_dir_of(meta);
        } catch (...) {
            for (uint32_t s = 0; s < written; ++s) (void)std::remove(shard_file(pre, gen, s, t->n()).c_str());  // the unfinished generation goes
            throw;
        }
        t->generation = gen;
        if (old.ok)
            for (unsigned s = 0; s < old.n; ++s) (void)std::remove(shard_file(pre, old.gen, s, old.n).c_str());
    });
}

// Load into an EMPTY table.  Same shard count and block size: every shard reads its own file.  Otherwise the blocks
// are re-dealt: read in global order from the old files and appended (disk -> pinned host -> device; a LIVE table changes
// its layout without the host: mi_knn_sharded_rebalance).  A failure leaves the table empty.
int mi_knn_sharded_load(mi_knn_sharded* t, const char* prefix) {
    return guarded([&] {
        if (!t || !prefix) fail(MI_ERR_INVALID, "null argument");
        const std::string pre(prefix);
        const Manifest m = read_manifest(pre);
        if (!m.ok) fail(MI_ERR_IO, "%s.shards is missing or not a shard list", prefix);
        if (m.dim != t->dim) fail(MI_ERR_INVALID, "%s holds dim %u rows, the table has dim %u", prefix, m.dim, t->dim);
        auto file_of = [&](unsigned s) { return shard_file(pre, m.gen, s, m.n); };
        const unsigned on = m.n, ob = m.block;
        const unsigned long long orows = m.rows;
        {
            std::lock_guard<std::mutex> l(t->mu);
            if (t->rows != 0) fail(MI_ERR_INVALID, "mi_knn_sharded_load needs an empty table");
            for (mi_knn* sh : t->shard)
                if (sh->rows != 0) fail(MI_ERR_INVALID, "mi_knn_sharded_load needs an empty table");
            if (on == t->n() && ob == t->block) {
                try {
                    for (uint32_t s = 0; s < t->n(); ++s) {
                        const int e = mi_knn_load(t->shard[s], file_of(s).c_str());
                        if (e != MI_OK) fail(e, "%s", mi_last_error());
                        t->shard[s]->base = 0;  // ids come from the block-cyclic map, not from the file's base
                        const uint64_t want = sharded_rows_of(t, orows, s);
                        if (t->shard[s]->rows != want)
                            fail(MI_ERR_IO, "%s holds %llu rows, the shard list implies %llu", file_of(s).c_str(),
                                 (unsigned long long)t->shard[s]->rows, (unsigned long long)want);
                    }
                } catch (...) {
                    t->rows = 0;
                    roll_back(t);
                    throw;
                }
                t->rows = orows;
                t->generation = m.gen;
                return;
            }
        }
        // another layout: stream block by block in global order
        std::vector<FILE*> fs(on, nullptr);
        struct Closer { std::vector<FILE*>& v; ~Closer() { for (FILE* f : v) if (f) std::fclose(f); } } closer{fs};
        for (unsigned s = 0; s < on; ++s) {
            fs[s] = std::fopen(file_of(s).c_str(), "rb");
            if (!fs[s]) fail(MI_ERR_IO, "cannot open %s", file_of(s).c_str());
        }
        std::vector<float> buf((size_t)ob * t->dim);
        try {
            for (uint64_t r = 0; r < orows; r += ob) {
                const uint64_t len = std::min<uint64_t>(ob, orows - r), blk = r / ob;
                const unsigned s = (unsigned)(blk % on);
                const uint64_t local = (blk / on) * ob;
                if (fseeko(fs[s], (off_t)(32 + local * t->dim * 4), SEEK_SET) != 0 ||
                    std::fread(buf.data(), 4, (size_t)len * t->dim, fs[s]) != (size_t)len * t->dim)
                    fail(MI_ERR_IO, "%s is truncated", file_of(s).c_str());
                const int e = mi_knn_sharded_append(t, buf.data(), len, nullptr);
                if (e != MI_OK) fail(e, "%s", mi_last_error());
            }
        } catch (...) {
            std::lock_guard<std::mutex> l(t->mu);
            t->rows = 0;
            roll_back(t);
            throw;
        }
        t->generation = m.gen;
    });
}

}  // extern "C"
