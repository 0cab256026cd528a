// pipeline.hip — the reference's scan-loop body and its query as ONE stream pipeline
// (BASELINE config 4: "batch=256 embed (bf16 ViT) + top-10 over 10M x 768, fused on HIP streams"), on one GPU
// (mi_pipeline_create) or over the GPUs of a row-sharded table inside one process (mi_pipeline_create_sharded).
//
//   reference, per chunk (server/src/clip.rs:107-137):   flatten -> upload -> forward -> blocking readback
//                                                        -> split per 768 -> db.insert(rows)
//   reference, per request (server/src/search.rs:70-86): SELECT ... WHERE embedding <|K|> $reference
//
//   here, per lane (= one tower replica + the table shard on its GPU):
//          copy stream    H2D chunk i+1                      | under the forward of chunk i
//          ingest stream  forward(chunk i): its last kernel writes the n embeddings straight into rows
//                         [size, size+n) of the shard — no readback, no re-upload, no extra copy
//   one GPU:      search stream  scan of the query enqueued after chunk i (sees its rows), D2H of the k results
//   sharded:      the shards' own streams scan side by side, the lists are all-gathered and merged on the device
//                 (sharded.hip); a chunk is cut at the table's block boundaries and every run is embedded by the replica
//                 on the GPU that owns the block, so an embedding is born in its shard ("each replica
//                 appends its embeddings to its local shard") and the replicas of consecutive blocks work concurrently.
//
// Ordering is carried by events only (handles.h: mi_clip::order, mi_knn::writes / reads); the host
// blocks only to keep at most two chunks in flight per lane (so that a caller alternating two upload buffers
// may refill a buffer as soon as the NEXT ingest call has returned).
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"
#include "handles.h"

using namespace mi;

namespace {
constexpr int N_SLOTS = 16;
constexpr int N_SPANS = 32;
struct QuerySlot {
    float* h_q = nullptr;        // pinned staging of the query
    float* d_q = nullptr;
    uint64_t *d_idx = nullptr, *h_idx = nullptr;
    float *d_dist = nullptr, *h_dist = nullptr;
    uint32_t cap_k = 0, k = 0;
    uint64_t* user_idx = nullptr;
    float* user_dist = nullptr;
    hipEvent_t done = nullptr;
    bool busy = false;
};
struct Span { hipEvent_t a = nullptr, b = nullptr; bool used = false; };
// one tower replica feeding one shard, both on `device`
struct Lane {
    mi_clip* m = nullptr;
    mi_knn* t = nullptr;
    int device = 0;
    hipStream_t ingest = nullptr, copy = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
    float* d_in[2] = {nullptr, nullptr};
    size_t in_cap = 0;   // images per upload buffer
    uint64_t seq = 0;    // chunks enqueued on this lane
    Span fwd[N_SPANS];
    uint64_t n_fwd = 0;
};
}  // namespace

struct mi_pipeline {
    std::vector<Lane> lanes;
    mi_knn_sharded* st = nullptr;   // sharded form: lane s feeds st->shard[s]; null: one lane, lanes[0].t is the table
    int device = 0;                 // of the search stream (one-GPU form)
    hipStream_t search = nullptr;
    QuerySlot slots[N_SLOTS];
    int next_slot = 0;
    // device time of the forwards / scans, from timing events on their own streams (mi_pipeline_stats)
    Span scan[N_SPANS];
    uint64_t n_scan = 0;
    double st_n[2] = {0, 0}, st_ms[2] = {0, 0};
    // uploads of the previous ingest call that may still be reading the caller's buffer: (lane, buffer)
    std::vector<std::pair<int, int>> prev_uploads;
    std::mutex mu;
};

namespace {

void deliver(QuerySlot& s) {
    if (!s.busy) return;
    HIP_CHECK(hipEventSynchronize(s.done));
    if (s.user_idx) {  // a query whose results stay on the device (mi_pipeline_query_device) has nothing to hand over
        std::memcpy(s.user_idx, s.h_idx, (size_t)s.k * sizeof(uint64_t));
        std::memcpy(s.user_dist, s.h_dist, (size_t)s.k * sizeof(float));
    }
    s.busy = false;
}

void slot_reserve(QuerySlot& s, uint32_t dim, uint32_t k) {
    if (!s.done) HIP_CHECK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    if (!s.h_q) {
        HIP_CHECK(hipHostMalloc((void**)&s.h_q, (size_t)dim * 4, hipHostMallocDefault));
        HIP_CHECK(hipMalloc((void**)&s.d_q, (size_t)dim * 4));
    }
    if (k <= s.cap_k) return;
    if (s.d_idx) { HIP_CHECK(hipFree(s.d_idx)); HIP_CHECK(hipFree(s.d_dist)); HIP_CHECK(hipHostFree(s.h_idx)); HIP_CHECK(hipHostFree(s.h_dist)); }
    s.d_idx = nullptr; s.d_dist = nullptr; s.h_idx = nullptr; s.h_dist = nullptr; s.cap_k = 0;
    const uint32_t cap = (k + 63) / 64 * 64;
    HIP_CHECK(hipMalloc((void**)&s.d_idx, (size_t)cap * 8));
    HIP_CHECK(hipMalloc((void**)&s.d_dist, (size_t)cap * 4));
    HIP_CHECK(hipHostMalloc((void**)&s.h_idx, (size_t)cap * 8, hipHostMallocDefault));
    HIP_CHECK(hipHostMalloc((void**)&s.h_dist, (size_t)cap * 4, hipHostMallocDefault));
    s.cap_k = cap;
}

// fold a finished span into the totals (blocks until its end event has happened)
void collect(mi_pipeline* p, Span& sp, int kind) {
    if (!sp.used) return;
    HIP_CHECK(hipEventSynchronize(sp.b));
    float ms = 0.0f;
    HIP_CHECK(hipEventElapsedTime(&ms, sp.a, sp.b));
    p->st_n[kind] += 1;
    p->st_ms[kind] += ms;
    sp.used = false;
}

Span& span_begin(mi_pipeline* p, Span* ring, uint64_t* count, int kind, hipStream_t s) {
    Span& sp = ring[(*count)++ % N_SPANS];
    collect(p, sp, kind);  // ring full: the oldest span finished long ago
    if (!sp.a) { HIP_CHECK(hipEventCreate(&sp.a)); HIP_CHECK(hipEventCreate(&sp.b)); }
    HIP_CHECK(hipEventRecord(sp.a, s));
    return sp;
}

void span_end(Span& sp, hipStream_t s) {
    HIP_CHECK(hipEventRecord(sp.b, s));
    sp.used = true;
}

void free_spans(Span* ring) {
    for (int i = 0; i < N_SPANS; ++i) {
        if (ring[i].a) (void)hipEventDestroy(ring[i].a);
        if (ring[i].b) (void)hipEventDestroy(ring[i].b);
    }
}

void free_pipeline(mi_pipeline* p) {
    if (!p) return;
    for (Lane& ln : p->lanes) {
        (void)hipSetDevice(ln.device);
        for (hipStream_t s : {ln.ingest, ln.copy})
            if (s) (void)hipStreamSynchronize(s);
    }
    if (p->st) (void)mi_knn_sharded_sync(p->st);
    (void)hipSetDevice(p->device);
    if (p->search) (void)hipStreamSynchronize(p->search);
    free_spans(p->scan);
    for (auto& s : p->slots) {
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.h_q) (void)hipHostFree(s.h_q);
        if (s.d_q) (void)hipFree(s.d_q);
        if (s.d_idx) (void)hipFree(s.d_idx);
        if (s.d_dist) (void)hipFree(s.d_dist);
        if (s.h_idx) (void)hipHostFree(s.h_idx);
        if (s.h_dist) (void)hipHostFree(s.h_dist);
    }
    if (p->search) (void)hipStreamDestroy(p->search);
    for (Lane& ln : p->lanes) {
        (void)hipSetDevice(ln.device);
        free_spans(ln.fwd);
        for (int b = 0; b < 2; ++b) {
            if (ln.ev_up[b]) (void)hipEventDestroy(ln.ev_up[b]);
            if (ln.ev_used[b]) (void)hipEventDestroy(ln.ev_used[b]);
            if (ln.d_in[b]) (void)hipFree(ln.d_in[b]);
        }
        for (hipStream_t s : {ln.ingest, ln.copy})
            if (s) (void)hipStreamDestroy(s);
    }
    delete p;
}

void lane_init(Lane& ln, mi_clip* model, mi_knn* table) {
    if (!model || !table) fail(MI_ERR_INVALID, "null handle");
    if (model->text) fail(MI_ERR_INVALID, "the pipeline ingests images: pass the image tower");
    if (model->device != table->device)
        fail(MI_ERR_INVALID, "model on device %d, table shard on device %d: a tower replica feeds the shard on its own GPU", model->device,
             table->device);
    if ((uint32_t)model->E != table->dim)
        fail(MI_ERR_INVALID, "the model embeds into %d dimensions, the table holds %u", model->E, table->dim);
    ln.m = model; ln.t = table; ln.device = model->device;
    DeviceGuard g(ln.device);
    HIP_CHECK(hipStreamCreateWithFlags(&ln.ingest, hipStreamNonBlocking));
    HIP_CHECK(hipStreamCreateWithFlags(&ln.copy, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
        HIP_CHECK(hipEventCreateWithFlags(&ln.ev_up[b], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ln.ev_used[b], hipEventDisableTiming));
    }
}

// n images at nchw into rows [t->rows, t->rows + n) of the lane's shard, in passes of at most max_batch
void lane_ingest(mi_pipeline* p, int li, const float* nchw, size_t n) {
    Lane& ln = p->lanes[(size_t)li];
    mi_clip* m = ln.m;
    mi_knn* t = ln.t;
    std::scoped_lock l(m->mu, t->mu);
    DeviceGuard g(ln.device);
    const size_t px = (size_t)m->image * m->image * 3;
    const size_t chunk = std::min(n, m->max_batch);
    clip_ensure_workspace(m, chunk);
    if (chunk > ln.in_cap) {
        for (hipStream_t s : {ln.ingest, ln.copy}) HIP_CHECK(hipStreamSynchronize(s));
        for (int b = 0; b < 2; ++b) {
            if (ln.d_in[b]) HIP_CHECK(hipFree(ln.d_in[b]));
            ln.d_in[b] = nullptr;
            HIP_CHECK(hipMalloc((void**)&ln.d_in[b], chunk * px * 4));
        }
        ln.in_cap = chunk;
    }
    knn_grow(t, t->rows + n);  // a reallocation waits for everything in flight (reserve ahead to avoid it)
    for (size_t i = 0; i < n; i += chunk) {
        const size_t c = std::min(chunk, n - i);
        const int b = (int)(ln.seq & 1);
        // the caller's previous buffer is free once its upload has finished: wait for it here, so that
        // "reuse a host buffer after the following call has returned" holds and at most two chunks queue up
        if (ln.seq >= 1) HIP_CHECK(hipEventSynchronize(ln.ev_up[b ^ 1]));
        if (ln.seq >= 2) HIP_CHECK(hipStreamWaitEvent(ln.copy, ln.ev_used[b], 0));  // forward(seq-2) consumed d_in[b]
        HIP_CHECK(hipMemcpyAsync(ln.d_in[b], nchw + i * px, c * px * 4, hipMemcpyHostToDevice, ln.copy));
        HIP_CHECK(hipEventRecord(ln.ev_up[b], ln.copy));
        p->prev_uploads.emplace_back(li, b);
        m->order.begin(ln.ingest);
        t->writes.begin(ln.ingest);
        HIP_CHECK(hipStreamWaitEvent(ln.ingest, ln.ev_up[b], 0));
        auto& sp = span_begin(p, ln.fwd, &ln.n_fwd, 0, ln.ingest);
        // option "front_overlap": the forward's front (the only reader of d_in) on the copy stream, right behind its upload
        clip_forward(m, ln.d_in[b], c, t->table + t->rows * t->dim, ln.ingest, m->front_overlap ? ln.copy : nullptr);
        span_end(sp, ln.ingest);
        HIP_CHECK(hipEventRecord(ln.ev_used[b], ln.ingest));
        m->order.end(ln.ingest);
        t->writes.end(ln.ingest);
        t->rows += c;
        ++ln.seq;
    }
}

}  // namespace

extern "C" {

int mi_pipeline_create(mi_clip* model, mi_knn* table, mi_pipeline** out) {
    mi_pipeline* p = nullptr;
    const int rc = guarded([&] {
        if (!out) fail(MI_ERR_INVALID, "out is null");
        *out = nullptr;
        p = new mi_pipeline();
        p->lanes.resize(1);
        lane_init(p->lanes[0], model, table);
        p->device = model->device;
        DeviceGuard g(p->device);
        HIP_CHECK(hipStreamCreateWithFlags(&p->search, hipStreamNonBlocking));
        *out = p;
    });
    if (rc != MI_OK && p) free_pipeline(p);
    return rc;
}

// One server process over the GPUs of one node (BASELINE config 5 as the reference would run it: one AppState, one
// table, scan task and search handler side by side — server/src/main.rs:30-35, server/src/clip.rs:112-137): models[s] is
// the tower replica on the device of the table's shard s (a handle may be listed for several shards of its GPU).
int mi_pipeline_create_sharded(mi_clip* const* models, int n_models, mi_knn_sharded* table, mi_pipeline** out) {
    mi_pipeline* p = nullptr;
    const int rc = guarded([&] {
        if (!out) fail(MI_ERR_INVALID, "out is null");
        *out = nullptr;
        if (!models || !table) fail(MI_ERR_INVALID, "null handle");
        if (n_models != (int)table->n()) fail(MI_ERR_INVALID, "%d models for %u shards: one tower replica per shard", n_models, table->n());
        p = new mi_pipeline();
        p->st = table;
        p->lanes.resize((size_t)n_models);
        for (int s = 0; s < n_models; ++s) lane_init(p->lanes[(size_t)s], models[s], table->shard[(size_t)s]);
        p->device = p->lanes[0].device;
        *out = p;
    });
    if (rc != MI_OK && p) free_pipeline(p);
    return rc;
}

void mi_pipeline_free(mi_pipeline* p) { free_pipeline(p); }

int mi_pipeline_ingest(mi_pipeline* p, const float* nchw, size_t n, uint64_t* first_id) {
    return guarded([&] {
        if (!p) fail(MI_ERR_INVALID, "null pipeline handle");
        std::lock_guard<std::mutex> lp(p->mu);
        // the buffer of the PREVIOUS call may be refilled once this call returns: its uploads must have finished
        for (auto [li, b] : p->prev_uploads) HIP_CHECK(hipEventSynchronize(p->lanes[(size_t)li].ev_up[b]));
        p->prev_uploads.clear();
        if (!p->st) {
            mi_knn* t = p->lanes[0].t;
            if (first_id) *first_id = t->base + t->rows;
            if (n == 0) return;  // the reference forwards and inserts an empty chunk (server/src/clip.rs:112-137)
            if (!nchw) fail(MI_ERR_INVALID, "nchw is null");
            lane_ingest(p, 0, nchw, n);
            return;
        }
        mi_knn_sharded* st = p->st;
        std::lock_guard<std::mutex> lt(st->mu);
        if (first_id) *first_id = st->rows;
        if (n == 0) return;
        if (!nchw) fail(MI_ERR_INVALID, "nchw is null");
        const size_t px = (size_t)p->lanes[0].m->image * p->lanes[0].m->image * 3;
        // cut at the block boundaries: every run is embedded on the GPU that owns its block and lands there
        uint64_t r = st->rows;
        const uint64_t end = st->rows + n;
        try {
            while (r < end) {
                const uint64_t len = std::min<uint64_t>(st->block - r % st->block, end - r);
                uint32_t s; uint64_t local;
                sharded_place(st, r, &s, &local);
                if (local != st->shard[s]->rows) fail(MI_ERR_INVALID, "shard %u out of step (%llu rows, expected %llu)", s,
                                                      (unsigned long long)st->shard[s]->rows, (unsigned long long)local);
                lane_ingest(p, (int)s, nchw + (r - st->rows) * px, (size_t)len);
                r += len;
            }
        } catch (...) {
            for (uint32_t s = 0; s < st->n(); ++s) knn_truncate(st->shard[s], sharded_rows_of(st, st->rows, s));
            throw;
        }
        st->rows = end;
    });
}

int mi_pipeline_query(mi_pipeline* p, const float* q, uint32_t k, uint64_t* idx, float* dist) {
    return guarded([&] {
        if (!p) fail(MI_ERR_INVALID, "null pipeline handle");
        if (!q || !idx || !dist) fail(MI_ERR_INVALID, "null query/result pointer");
        if (k == 0) fail(MI_ERR_INVALID, "k must be >= 1");
        std::lock_guard<std::mutex> lp(p->mu);
        if (p->st) {  // every shard scans on its own stream, lists all-gathered and merged on the device; results at sync / drain
            std::lock_guard<std::mutex> lt(p->st->mu);
            sharded_search_enqueue(p->st, q, 1, k, idx, dist);
            return;
        }
        mi_knn* t = p->lanes[0].t;
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(p->device);
        QuerySlot& s = p->slots[p->next_slot];
        p->next_slot = (p->next_slot + 1) % N_SLOTS;
        deliver(s);  // the ring is full only when 16 queries are pending: finish the oldest
        slot_reserve(s, t->dim, k);
        std::memcpy(s.h_q, q, (size_t)t->dim * 4);
        s.k = k; s.user_idx = idx; s.user_dist = dist;
        HIP_CHECK(hipMemcpyAsync(s.d_q, s.h_q, (size_t)t->dim * 4, hipMemcpyHostToDevice, p->search));
        t->writes.begin(p->search);  // every row counted in t->rows has landed before the scan reads it
        t->reads.begin(p->search);   // searches share the candidate workspace
        auto& sp = span_begin(p, p->scan, &p->n_scan, 1, p->search);
        knn_search_one(t, s.d_q, k, s.d_idx, s.d_dist, p->search);
        span_end(sp, p->search);
        t->reads.end(p->search);
        HIP_CHECK(hipMemcpyAsync(s.h_idx, s.d_idx, (size_t)k * 8, hipMemcpyDeviceToHost, p->search));
        HIP_CHECK(hipMemcpyAsync(s.h_dist, s.d_dist, (size_t)k * 4, hipMemcpyDeviceToHost, p->search));
        HIP_CHECK(hipEventRecord(s.done, p->search));
        s.busy = true;
    });
}

// The same query with its k results left on the device: the per-shard list a multi-GPU caller hands to the
// all-gather (one process per GPU: torch.distributed / RCCL on `consumer_stream`) without a trip through the host.
int mi_pipeline_query_device(mi_pipeline* p, const float* q, uint32_t k, uint64_t* d_idx, float* d_dist, void* consumer_stream) {
    return guarded([&] {
        if (!p) fail(MI_ERR_INVALID, "null pipeline handle");
        if (!q || !d_idx || !d_dist) fail(MI_ERR_INVALID, "null query/result pointer");
        if (k == 0) fail(MI_ERR_INVALID, "k must be >= 1");
        std::lock_guard<std::mutex> lp(p->mu);
        if (p->st) fail(MI_ERR_UNSUPPORTED, "a sharded pipeline gathers and merges its shards' lists itself: use mi_pipeline_query");
        mi_knn* t = p->lanes[0].t;
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(p->device);
        QuerySlot& s = p->slots[p->next_slot];
        p->next_slot = (p->next_slot + 1) % N_SLOTS;
        deliver(s);
        slot_reserve(s, t->dim, 0);  // only the query's staging buffers
        std::memcpy(s.h_q, q, (size_t)t->dim * 4);
        s.k = k; s.user_idx = nullptr; s.user_dist = nullptr;
        HIP_CHECK(hipMemcpyAsync(s.d_q, s.h_q, (size_t)t->dim * 4, hipMemcpyHostToDevice, p->search));
        t->writes.begin(p->search);
        t->reads.begin(p->search);
        auto& sp = span_begin(p, p->scan, &p->n_scan, 1, p->search);
        knn_search_one(t, s.d_q, k, d_idx, d_dist, p->search);
        span_end(sp, p->search);
        t->reads.end(p->search);
        HIP_CHECK(hipEventRecord(s.done, p->search));
        // NULL is a stream too — the device's default stream, which is what torch.cuda.current_stream() is unless the
        // caller changed it; it does not synchronise with the (non-blocking) search stream by itself
        HIP_CHECK(hipStreamWaitEvent((hipStream_t)consumer_stream, s.done, 0));
        s.busy = true;
    });
}

int mi_pipeline_sync(mi_pipeline* p) {
    return guarded([&] {
        if (!p) fail(MI_ERR_INVALID, "null pipeline handle");
        std::lock_guard<std::mutex> lp(p->mu);
        for (Lane& ln : p->lanes) {
            DeviceGuard g(ln.device);
            HIP_CHECK(hipStreamSynchronize(ln.copy));
            HIP_CHECK(hipStreamSynchronize(ln.ingest));
            for (auto& sp : ln.fwd) collect(p, sp, 0);
        }
        p->prev_uploads.clear();
        if (p->st) {
            std::lock_guard<std::mutex> lt(p->st->mu);
            sharded_deliver_all(p->st);
            return;
        }
        DeviceGuard g(p->device);
        HIP_CHECK(hipStreamSynchronize(p->search));
        for (auto& s : p->slots) deliver(s);
        for (auto& sp : p->scan) collect(p, sp, 1);
    });
}

int mi_pipeline_drain(mi_pipeline* p, uint32_t leave_pending) {
    return guarded([&] {
        if (!p) fail(MI_ERR_INVALID, "null pipeline handle");
        std::lock_guard<std::mutex> lp(p->mu);
        if (p->st) {  // the sharded table keeps its own ring: everything pending is delivered
            std::lock_guard<std::mutex> lt(p->st->mu);
            sharded_deliver_all(p->st);
            return;
        }
        DeviceGuard g(p->device);
        // slots are used round-robin: walk from the oldest, stop when only `leave_pending` busy ones remain
        uint32_t busy = 0;
        for (auto& s : p->slots) busy += s.busy ? 1 : 0;
        for (int i = 0; i < N_SLOTS && busy > leave_pending; ++i) {
            QuerySlot& s = p->slots[(p->next_slot + i) % N_SLOTS];
            if (s.busy) { deliver(s); --busy; }
        }
    });
}

int mi_pipeline_stats(mi_pipeline* p, double out[4], int reset) {
    return guarded([&] {
        if (!p || !out) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> lp(p->mu);
        out[0] = p->st_n[0]; out[1] = p->st_ms[0]; out[2] = p->st_n[1]; out[3] = p->st_ms[1];
        if (reset) { p->st_n[0] = p->st_n[1] = 0; p->st_ms[0] = p->st_ms[1] = 0; }
    });
}

int mi_host_alloc(size_t bytes, void** out) {
    return guarded([&] {
        if (!out) fail(MI_ERR_INVALID, "out is null");
        *out = nullptr;
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(MI_ERR_NO_DEVICE, "no HIP device visible: pinned memory needs the runtime");
        // portable: a chunk in this memory is uploaded to several GPUs by a sharded pipeline
        HIP_CHECK(hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocPortable));
    });
}

void mi_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

}  // extern "C"
