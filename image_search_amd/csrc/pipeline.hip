// pipeline.hip — the reference's scan-loop body and its query as ONE stream pipeline on one GPU
// (BASELINE config 4: "batch=256 embed (bf16 ViT) + top-10 over 10M x 768, fused on HIP streams").
//
//   reference, per chunk (server/src/clip.rs:107-137):   flatten -> upload -> forward -> blocking readback
//                                                        -> split per 768 -> db.insert(rows)
//   reference, per request (server/src/search.rs:70-86): SELECT ... WHERE embedding <|K|> $reference
//
//   here:  copy stream    H2D chunk i+1                      | under the forward of chunk i
//          ingest stream  forward(chunk i): its last kernel writes the n embeddings straight into rows
//                         [size, size+n) of the table — no readback, no re-upload, no extra copy
//          search stream  scan of the query enqueued after chunk i (sees its rows), D2H of the k results
//                         | under the forward of chunk i+1 (HBM-bound scan beside the MFMA-bound tower)
//
// Ordering is carried by events only (handles.h: mi_clip::order, mi_knn::writes / reads); the host
// blocks only to keep at most two chunks in flight (so that a caller alternating two upload buffers
// may refill a buffer as soon as the NEXT ingest call has returned).
#include <cstdlib>
#include <cstring>
#include <mutex>

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
}  // namespace

struct mi_pipeline {
    mi_clip* m = nullptr;
    mi_knn* t = nullptr;
    int device = 0;
    hipStream_t ingest = nullptr, copy = nullptr, search = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
    float* d_in[2] = {nullptr, nullptr};
    size_t in_cap = 0;  // images per upload buffer
    uint64_t seq = 0;   // chunks enqueued
    QuerySlot slots[N_SLOTS];
    int next_slot = 0;
    // device time of the forwards / scans, from timing events on their own streams (mi_pipeline_stats)
    struct Span { hipEvent_t a = nullptr, b = nullptr; bool used = false; };
    Span fwd[N_SPANS], scan[N_SPANS];
    uint64_t n_fwd = 0, n_scan = 0;
    double st_n[2] = {0, 0}, st_ms[2] = {0, 0};
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
void collect(mi_pipeline* p, mi_pipeline::Span& sp, int kind) {
    if (!sp.used) return;
    HIP_CHECK(hipEventSynchronize(sp.b));
    float ms = 0.0f;
    HIP_CHECK(hipEventElapsedTime(&ms, sp.a, sp.b));
    p->st_n[kind] += 1;
    p->st_ms[kind] += ms;
    sp.used = false;
}

mi_pipeline::Span& span_begin(mi_pipeline* p, int kind, hipStream_t s) {
    mi_pipeline::Span& sp = kind == 0 ? p->fwd[p->n_fwd++ % N_SPANS] : p->scan[p->n_scan++ % N_SPANS];
    collect(p, sp, kind);  // ring full: the oldest span finished long ago
    if (!sp.a) { HIP_CHECK(hipEventCreate(&sp.a)); HIP_CHECK(hipEventCreate(&sp.b)); }
    HIP_CHECK(hipEventRecord(sp.a, s));
    return sp;
}

void span_end(mi_pipeline::Span& sp, hipStream_t s) {
    HIP_CHECK(hipEventRecord(sp.b, s));
    sp.used = true;
}

void free_pipeline(mi_pipeline* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (hipStream_t s : {p->ingest, p->copy, p->search})
        if (s) (void)hipStreamSynchronize(s);
    for (auto* ring : {p->fwd, p->scan})
        for (int i = 0; i < N_SPANS; ++i) {
            if (ring[i].a) (void)hipEventDestroy(ring[i].a);
            if (ring[i].b) (void)hipEventDestroy(ring[i].b);
        }
    for (auto& s : p->slots) {
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.h_q) (void)hipHostFree(s.h_q);
        if (s.d_q) (void)hipFree(s.d_q);
        if (s.d_idx) (void)hipFree(s.d_idx);
        if (s.d_dist) (void)hipFree(s.d_dist);
        if (s.h_idx) (void)hipHostFree(s.h_idx);
        if (s.h_dist) (void)hipHostFree(s.h_dist);
    }
    for (int b = 0; b < 2; ++b) {
        if (p->ev_up[b]) (void)hipEventDestroy(p->ev_up[b]);
        if (p->ev_used[b]) (void)hipEventDestroy(p->ev_used[b]);
        if (p->d_in[b]) (void)hipFree(p->d_in[b]);
    }
    for (hipStream_t s : {p->ingest, p->copy, p->search})
        if (s) (void)hipStreamDestroy(s);
    delete p;
}

}  // namespace

extern "C" {

int mi_pipeline_create(mi_clip* model, mi_knn* table, mi_pipeline** out) {
    mi_pipeline* p = nullptr;
    const int rc = guarded([&] {
        if (!out) fail(MI_ERR_INVALID, "out is null");
        *out = nullptr;
        if (!model || !table) fail(MI_ERR_INVALID, "null handle");
        if (model->text) fail(MI_ERR_INVALID, "the pipeline ingests images: pass the image tower");
        if (model->device != table->device)
            fail(MI_ERR_INVALID, "model on device %d, table on device %d: one pipeline per GPU", model->device, table->device);
        if ((uint32_t)model->E != table->dim)
            fail(MI_ERR_INVALID, "the model embeds into %d dimensions, the table holds %u", model->E, table->dim);
        DeviceGuard g(model->device);
        p = new mi_pipeline();
        p->m = model; p->t = table; p->device = model->device;
        HIP_CHECK(hipStreamCreateWithFlags(&p->ingest, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&p->copy, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&p->search, hipStreamNonBlocking));
        for (int b = 0; b < 2; ++b) {
            HIP_CHECK(hipEventCreateWithFlags(&p->ev_up[b], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&p->ev_used[b], hipEventDisableTiming));
        }
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
        mi_clip* m = p->m;
        mi_knn* t = p->t;
        if (first_id) *first_id = t->base + t->rows;
        if (n == 0) return;  // the reference forwards and inserts an empty chunk (server/src/clip.rs:112-137)
        if (!nchw) fail(MI_ERR_INVALID, "nchw is null");
        std::scoped_lock l(m->mu, t->mu);
        DeviceGuard g(p->device);
        const size_t px = (size_t)m->image * m->image * 3;
        const size_t chunk = std::min(n, m->max_batch);
        clip_ensure_workspace(m, chunk);
        if (chunk > p->in_cap) {
            for (hipStream_t s : {p->ingest, p->copy}) HIP_CHECK(hipStreamSynchronize(s));
            for (int b = 0; b < 2; ++b) {
                if (p->d_in[b]) HIP_CHECK(hipFree(p->d_in[b]));
                p->d_in[b] = nullptr;
                HIP_CHECK(hipMalloc((void**)&p->d_in[b], chunk * px * 4));
            }
            p->in_cap = chunk;
        }
        knn_grow(t, t->rows + n);  // a reallocation waits for everything in flight (reserve ahead to avoid it)
        for (size_t i = 0; i < n; i += chunk) {
            const size_t c = std::min(chunk, n - i);
            const int b = (int)(p->seq & 1);
            // the caller's previous buffer is free once its upload has finished: wait for it here, so that
            // "reuse a host buffer after the following call has returned" holds and at most two chunks queue up
            if (p->seq >= 1) HIP_CHECK(hipEventSynchronize(p->ev_up[b ^ 1]));
            if (p->seq >= 2) HIP_CHECK(hipStreamWaitEvent(p->copy, p->ev_used[b], 0));  // forward(seq-2) consumed d_in[b]
            HIP_CHECK(hipMemcpyAsync(p->d_in[b], nchw + i * px, c * px * 4, hipMemcpyHostToDevice, p->copy));
            HIP_CHECK(hipEventRecord(p->ev_up[b], p->copy));
            m->order.begin(p->ingest);
            t->writes.begin(p->ingest);
            HIP_CHECK(hipStreamWaitEvent(p->ingest, p->ev_up[b], 0));
            auto& sp = span_begin(p, 0, p->ingest);
            clip_forward(m, p->d_in[b], c, t->table + t->rows * t->dim, p->ingest);
            span_end(sp, p->ingest);
            HIP_CHECK(hipEventRecord(p->ev_used[b], p->ingest));
            m->order.end(p->ingest);
            t->writes.end(p->ingest);
            t->rows += c;
            ++p->seq;
        }
    });
}

int mi_pipeline_query(mi_pipeline* p, const float* q, uint32_t k, uint64_t* idx, float* dist) {
    return guarded([&] {
        if (!p) fail(MI_ERR_INVALID, "null pipeline handle");
        if (!q || !idx || !dist) fail(MI_ERR_INVALID, "null query/result pointer");
        if (k == 0) fail(MI_ERR_INVALID, "k must be >= 1");
        std::lock_guard<std::mutex> lp(p->mu);
        mi_knn* t = p->t;
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
        auto& sp = span_begin(p, 1, p->search);
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
        mi_knn* t = p->t;
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
        auto& sp = span_begin(p, 1, p->search);
        knn_search_one(t, s.d_q, k, d_idx, d_dist, p->search);
        span_end(sp, p->search);
        t->reads.end(p->search);
        HIP_CHECK(hipEventRecord(s.done, p->search));
        if (consumer_stream) HIP_CHECK(hipStreamWaitEvent((hipStream_t)consumer_stream, s.done, 0));
        s.busy = true;
    });
}

int mi_pipeline_sync(mi_pipeline* p) {
    return guarded([&] {
        if (!p) fail(MI_ERR_INVALID, "null pipeline handle");
        std::lock_guard<std::mutex> lp(p->mu);
        DeviceGuard g(p->device);
        HIP_CHECK(hipStreamSynchronize(p->copy));
        HIP_CHECK(hipStreamSynchronize(p->ingest));
        HIP_CHECK(hipStreamSynchronize(p->search));
        for (auto& s : p->slots) deliver(s);
        for (auto& sp : p->fwd) collect(p, sp, 0);
        for (auto& sp : p->scan) collect(p, sp, 1);
    });
}

int mi_pipeline_drain(mi_pipeline* p, uint32_t leave_pending) {
    return guarded([&] {
        if (!p) fail(MI_ERR_INVALID, "null pipeline handle");
        std::lock_guard<std::mutex> lp(p->mu);
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
        HIP_CHECK(hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault));
    });
}

void mi_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

}  // extern "C"
