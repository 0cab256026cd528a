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
//     ncclCommInitAll: single-process multi-GPU) and are merged once under the same ordering.
//     RCCL is bound at run time (dlopen): a one-GPU table never needs it, and a process that already holds
//     PyTorch's bundled copy must not map a second one.  Where RCCL cannot run — the same device listed
//     twice (how the one-GPU test box exercises n > 1 shards), or no librccl — the lists are gathered
//     through pinned host memory instead; both transports feed the same merge.
//   * load with another shard count re-deals the blocks (rebalancing through the host; a direct
//     device-to-device move over xGMI is not built).
#include <dlfcn.h>

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

namespace mi {
// core.hip: the one merge both the host entry point and the sharded search use
void merge_lists(const uint64_t* idx_in, const float* dist_in, uint32_t lists, uint32_t k, uint64_t* idx, float* dist);
}  // namespace mi

namespace {

// ---- RCCL, bound at run time -------------------------------------------------------------------
typedef struct ncclComm* ncclComm_t;
enum { kNcclUint64 = 5, kNcclFloat32 = 7 };
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

struct Pinned {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t b) {
        if (b <= cap) return;
        if (p) HIP_CHECK(hipHostFree(p));
        p = nullptr; cap = 0;
        HIP_CHECK(hipHostMalloc(&p, b, hipHostMallocDefault));
        cap = b;
    }
    ~Pinned() { if (p) (void)hipHostFree(p); }
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t b) {  // caller has the device selected
        if (b <= cap) return;
        if (p) HIP_CHECK(hipFree(p));
        p = nullptr; cap = 0;
        HIP_CHECK(hipMalloc(&p, b));
        cap = b;
    }
};

}  // namespace

struct mi_knn_sharded {
    uint32_t dim = 0, block = 0;
    uint64_t rows = 0;
    std::vector<int> devices;
    std::vector<mi_knn*> shard;
    // per shard: query, local results, gathered results (RCCL receive side)
    std::vector<DevBuf> d_q, d_idx, d_dist, g_idx, g_dist;
    Pinned h_q, h_idx, h_dist;
    std::vector<ncclComm_t> comms;
    bool use_rccl = false;
    std::mutex mu;
    uint32_t n() const { return (uint32_t)shard.size(); }
};

namespace {

// global row -> (shard, local row)
inline void place(const mi_knn_sharded* t, uint64_t r, uint32_t* s, uint64_t* local) {
    const uint64_t blk = r / t->block;
    *s = (uint32_t)(blk % t->n());
    *local = (blk / t->n()) * t->block + r % t->block;
}

void free_sharded(mi_knn_sharded* t) {
    if (!t) return;
    for (size_t s = 0; s < t->shard.size(); ++s) {
        (void)hipSetDevice(t->devices[s]);
        (void)hipDeviceSynchronize();
        for (DevBuf* b : {&t->d_q[s], &t->d_idx[s], &t->d_dist[s], &t->g_idx[s], &t->g_dist[s]})
            if (b->p) (void)hipFree(b->p);
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
        place(t, r, &s, &local);
        fn(r, len, s, local);
        r += len;
    }
}

}  // namespace

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
        t->d_q.resize(n_dev); t->d_idx.resize(n_dev); t->d_dist.resize(n_dev); t->g_idx.resize(n_dev); t->g_dist.resize(n_dev);
        for (int s = 0; s < n_dev; ++s) {
            mi_knn* h = nullptr;
            const int e = mi_knn_create(dim, devices[s], &h);
            if (e != MI_OK) fail(e, "%s", mi_last_error());
            h->cyc_block = block_rows; h->cyc_n = (uint32_t)n_dev; h->cyc_rank = (uint32_t)s;
            t->shard.push_back(h);
        }
        // RCCL only between distinct devices (a communicator refuses one GPU twice); n_dev == 1 gathers nothing
        bool distinct = true;
        for (int a = 0; a < n_dev; ++a)
            for (int b = a + 1; b < n_dev; ++b) distinct = distinct && devices[a] != devices[b];
        const char* force = std::getenv("MI_KNN_SHARDED_TRANSPORT");  // "host" | "rccl": read once, at creation (tests)
        const bool want = force ? std::string(force) == "rccl" : (n_dev > 1 && distinct);
        if (want && !(force && std::string(force) == "host")) {
            if (!rccl().ok) {
                if (force) fail(MI_ERR_UNSUPPORTED, "librccl.so not found: the all-gather transport is unavailable");
            } else if (distinct) {
                t->comms.assign(n_dev, nullptr);
                RCCL_CHECK(rccl().CommInitAll(t->comms.data(), n_dev, devices));
                t->use_rccl = true;
            } else if (force) {
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
        if (transport) *transport = t->n() == 1 ? 0 : (t->use_rccl ? 2 : 1);
    });
}

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
        const uint64_t blocks = (rows + t->block - 1) / t->block;
        for (uint32_t s = 0; s < t->n(); ++s) {
            const uint64_t mine = blocks / t->n() + (s < blocks % t->n() ? 1 : 0);
            const int e = mi_knn_reserve(t->shard[s], mine * t->block);
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
        for_runs(t, t->rows, n, [&](uint64_t r, uint64_t len, uint32_t s, uint64_t local) {
            if (local != t->shard[s]->rows) fail(MI_ERR_INVALID, "shard %u out of step (%llu rows, expected %llu)", s,
                                                 (unsigned long long)t->shard[s]->rows, (unsigned long long)local);
            const int e = mi_knn_append(t->shard[s], rows + (r - t->rows) * t->dim, len);
            if (e != MI_OK) fail(e, "%s", mi_last_error());
        });
        t->rows += n;
    });
}

int mi_knn_sharded_append_synthetic(mi_knn_sharded* t, uint64_t seed, uint64_t first_row, uint64_t n) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        std::lock_guard<std::mutex> l(t->mu);
        for_runs(t, t->rows, n, [&](uint64_t r, uint64_t len, uint32_t s, uint64_t) {
            const int e = mi_knn_append_synthetic(t->shard[s], seed, first_row + (r - t->rows), len);
            if (e != MI_OK) fail(e, "%s", mi_last_error());
        });
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

// Replaces `SELECT id, image_path, vector::distance::knn() FROM image WHERE embedding <|K|> $reference`
// (server/src/search.rs:70-86) over all shards; same results and ordering as ONE mi_knn holding every row.
int mi_knn_sharded_search(mi_knn_sharded* t, const float* q, uint32_t nq, uint32_t k, uint64_t* idx, float* dist) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        if (nq && (!q || !idx || !dist)) fail(MI_ERR_INVALID, "null query/result pointer");
        if (k == 0) fail(MI_ERR_INVALID, "k must be >= 1");
        if (nq == 0) return;
        std::lock_guard<std::mutex> l(t->mu);
        const uint32_t n = t->n();
        const size_t per = (size_t)nq * k;  // results per shard
        t->h_q.reserve((size_t)nq * t->dim * 4);
        std::memcpy(t->h_q.p, q, (size_t)nq * t->dim * 4);
        t->h_idx.reserve(per * n * 8);
        t->h_dist.reserve(per * n * 4);
        // 1. every shard: query up, scan on the shard's own stream — the n scans run side by side
        for (uint32_t s = 0; s < n; ++s) {
            mi_knn* sh = t->shard[s];
            std::lock_guard<std::mutex> ls(sh->mu);
            DeviceGuard g(sh->device);
            hipStream_t st = knn_own_stream(sh);
            t->d_q[s].reserve((size_t)nq * t->dim * 4);
            t->d_idx[s].reserve(per * 8);
            t->d_dist[s].reserve(per * 4);
            HIP_CHECK(hipMemcpyAsync(t->d_q[s].p, t->h_q.p, (size_t)nq * t->dim * 4, hipMemcpyHostToDevice, st));
            sh->writes.begin(st);
            sh->reads.begin(st);
            for (uint32_t u = 0; u < nq; ++u)
                knn_search_one(sh, (const float*)t->d_q[s].p + (size_t)u * t->dim, k, (uint64_t*)t->d_idx[s].p + (size_t)u * k,
                               (float*)t->d_dist[s].p + (size_t)u * k, st);
            sh->reads.end(st);
        }
        // 2. the one exchange step: 12 k bytes per shard and query
        if (t->use_rccl) {
            for (uint32_t s = 0; s < n; ++s) {
                DeviceGuard g(t->shard[s]->device);
                t->g_idx[s].reserve(per * n * 8);
                t->g_dist[s].reserve(per * n * 4);
            }
            RCCL_CHECK(rccl().GroupStart());
            for (uint32_t s = 0; s < n; ++s) {
                hipStream_t st = t->shard[s]->stream;
                RCCL_CHECK(rccl().AllGather(t->d_idx[s].p, t->g_idx[s].p, per, kNcclUint64, t->comms[s], st));
                RCCL_CHECK(rccl().AllGather(t->d_dist[s].p, t->g_dist[s].p, per, kNcclFloat32, t->comms[s], st));
            }
            RCCL_CHECK(rccl().GroupEnd());
            {   // every device now holds all lists, rank-major; the host reads them from the first
                mi_knn* sh = t->shard[0];
                DeviceGuard g(sh->device);
                HIP_CHECK(hipMemcpyAsync(t->h_idx.p, t->g_idx[0].p, per * n * 8, hipMemcpyDeviceToHost, sh->stream));
                HIP_CHECK(hipMemcpyAsync(t->h_dist.p, t->g_dist[0].p, per * n * 4, hipMemcpyDeviceToHost, sh->stream));
            }
            for (uint32_t s = 0; s < n; ++s) {  // the collective is complete on a device when its stream is
                DeviceGuard g(t->shard[s]->device);
                HIP_CHECK(hipStreamSynchronize(t->shard[s]->stream));
            }
        } else {
            for (uint32_t s = 0; s < n; ++s) {
                mi_knn* sh = t->shard[s];
                DeviceGuard g(sh->device);
                HIP_CHECK(hipMemcpyAsync((uint64_t*)t->h_idx.p + per * s, t->d_idx[s].p, per * 8, hipMemcpyDeviceToHost, sh->stream));
                HIP_CHECK(hipMemcpyAsync((float*)t->h_dist.p + per * s, t->d_dist[s].p, per * 4, hipMemcpyDeviceToHost, sh->stream));
            }
            for (uint32_t s = 0; s < n; ++s) {
                DeviceGuard g(t->shard[s]->device);
                HIP_CHECK(hipStreamSynchronize(t->shard[s]->stream));
            }
        }
        // 3. one merge per query over the n lists (h_*: [shard][query][k])
        if (n == 1) {
            std::memcpy(idx, t->h_idx.p, per * 8);
            std::memcpy(dist, t->h_dist.p, per * 4);
            return;
        }
        std::vector<uint64_t> li((size_t)n * k);
        std::vector<float> ld((size_t)n * k);
        for (uint32_t u = 0; u < nq; ++u) {
            for (uint32_t s = 0; s < n; ++s) {
                std::memcpy(&li[(size_t)s * k], (const uint64_t*)t->h_idx.p + per * s + (size_t)u * k, (size_t)k * 8);
                std::memcpy(&ld[(size_t)s * k], (const float*)t->h_dist.p + per * s + (size_t)u * k, (size_t)k * 4);
            }
            merge_lists(li.data(), ld.data(), n, k, idx + (size_t)u * k, dist + (size_t)u * k);
        }
    });
}

// Persistence: one MIKNNv01 file per shard, `<prefix>.<s>of<n>.miknn`, plus `<prefix>.shards` (text: n, block, rows, dim).
int mi_knn_sharded_save(mi_knn_sharded* t, const char* prefix) {
    return guarded([&] {
        if (!t || !prefix) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(t->mu);
        for (uint32_t s = 0; s < t->n(); ++s) {
            const std::string f = std::string(prefix) + "." + std::to_string(s) + "of" + std::to_string(t->n()) + ".miknn";
            const int e = mi_knn_save(t->shard[s], f.c_str());
            if (e != MI_OK) fail(e, "%s", mi_last_error());
        }
        const std::string meta = std::string(prefix) + ".shards", tmp = meta + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "w");
        if (!f) fail(MI_ERR_IO, "cannot create %s", tmp.c_str());
        const bool ok = std::fprintf(f, "%u %u %llu %u\n", t->n(), t->block, (unsigned long long)t->rows, t->dim) > 0;
        if (std::fclose(f) != 0 || !ok) fail(MI_ERR_IO, "write to %s failed", tmp.c_str());
        if (std::rename(tmp.c_str(), meta.c_str()) != 0) fail(MI_ERR_IO, "cannot rename %s", tmp.c_str());  // the meta file last
    });
}

// Load into an EMPTY table.  Same shard count and block size: every shard reads its own file.  Otherwise the blocks
// are re-dealt: read in global order from the old files and appended (rebalancing through the host).
int mi_knn_sharded_load(mi_knn_sharded* t, const char* prefix) {
    return guarded([&] {
        if (!t || !prefix) fail(MI_ERR_INVALID, "null argument");
        unsigned on = 0, ob = 0, od = 0;
        unsigned long long orows = 0;
        {
            const std::string meta = std::string(prefix) + ".shards";
            FILE* f = std::fopen(meta.c_str(), "r");
            if (!f) fail(MI_ERR_IO, "cannot open %s", meta.c_str());
            const int got = std::fscanf(f, "%u %u %llu %u", &on, &ob, &orows, &od);
            std::fclose(f);
            if (got != 4 || on == 0 || ob == 0) fail(MI_ERR_IO, "%s is not a shard list", meta.c_str());
        }
        if (od != t->dim) fail(MI_ERR_INVALID, "%s holds dim %u rows, the table has dim %u", prefix, od, t->dim);
        auto file_of = [&](unsigned s) { return std::string(prefix) + "." + std::to_string(s) + "of" + std::to_string(on) + ".miknn"; };
        {
            std::lock_guard<std::mutex> l(t->mu);
            if (t->rows != 0) fail(MI_ERR_INVALID, "mi_knn_sharded_load needs an empty table");
            if (on == t->n() && ob == t->block) {
                uint64_t total = 0;
                for (uint32_t s = 0; s < t->n(); ++s) {
                    const int e = mi_knn_load(t->shard[s], file_of(s).c_str());
                    if (e != MI_OK) fail(e, "%s", mi_last_error());
                    t->shard[s]->base = 0;  // ids come from the block-cyclic map, not from the file's base
                    total += t->shard[s]->rows;
                }
                if (total != orows) fail(MI_ERR_IO, "%s: shard files hold %llu rows, the list says %llu", prefix,
                                         (unsigned long long)total, orows);
                t->rows = total;
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
    });
}

}  // extern "C"
