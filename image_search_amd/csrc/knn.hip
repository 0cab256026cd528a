// knn.hip — host side of Seam B behind the C ABI (include/mi355clip.h).
// One mi_knn = one row-shard of the reference's `image.embedding` column
// (server/src/search.rs:13-18) resident in HBM on one GPU.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <unistd.h>

#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "handles.h"
#include "knn_kernels.h"

using namespace mi;

namespace {

constexpr int MAX_BATCH_Q = 8;

IdMap id_map(const mi_knn* t) { return IdMap{t->base, t->cyc_block, t->cyc_n, t->cyc_rank}; }

void ensure(mi_knn* t, void** p, size_t* have, size_t want, size_t elem) {
    if (*have >= want) return;
    t->reads.sync();  // a search still in flight uses the buffer freed here
    if (*p) HIP_CHECK(hipFree(*p));
    *p = nullptr; *have = 0;
    HIP_CHECK(hipMalloc(p, want * elem));
    *have = want;
}

// as ensure(), keeping the first `keep` elements (a mirror that grows with the table must not be rebuilt from the fp32 rows)
void ensure_keep(mi_knn* t, void** p, size_t* have, size_t want, size_t elem, size_t keep) {
    if (*have >= want) return;
    t->reads.sync();
    void* np_ = nullptr;
    HIP_CHECK(hipMalloc(&np_, want * elem));
    if (*p && keep) HIP_CHECK(hipMemcpy(np_, *p, std::min(keep, *have) * elem, hipMemcpyDeviceToDevice));
    if (*p) HIP_CHECK(hipFree(*p));
    *p = np_;
    *have = want;
}

// the handle's own stream, created on first use (an idle stream still takes one of the four
// hardware queues HIP multiplexes streams onto)
hipStream_t own_stream(mi_knn* t) {
    if (!t->stream) HIP_CHECK(hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking));
    return t->stream;
}

void grow(mi_knn* t, uint64_t want_rows) {
    if (want_rows <= t->cap) return;
    uint64_t ncap = std::max<uint64_t>(want_rows, t->cap + t->cap / 2);
    ncap = (ncap + 63) & ~63ull;
    if (ncap > 0xFFFFFFFFull) fail(MI_ERR_UNSUPPORTED, "a shard holds at most 2^32-1 rows (asked %llu)",
                                    (unsigned long long)want_rows);
    // the table moves: everything enqueued against the old allocation must have finished
    t->writes.sync();
    t->reads.sync();
    float* nt = nullptr;
    HIP_CHECK(hipMalloc((void**)&nt, ncap * t->dim * sizeof(float)));
    if (t->rows) {
        HIP_CHECK(hipMemcpyAsync(nt, t->table, t->rows * t->dim * sizeof(float), hipMemcpyDeviceToDevice, own_stream(t)));
        HIP_CHECK(hipStreamSynchronize(t->stream));
    }
    if (t->table) HIP_CHECK(hipFree(t->table));
    t->table = nt;
    t->cap = ncap;
}

template <class K>
void allow_lds(K kernel, size_t bytes) {
    if (bytes > 48 * 1024)
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

// gy > 1: a group of queries in one launch (grid.y = query, knn_kernels.h QGroup)
template <class Top>
void launch_scan(mi_knn* t, const float* d_q, uint32_t k, const uint64_t* lo, uint64_t* cand, uint32_t blocks,
                 hipStream_t s, const uint32_t* run_if = nullptr, uint32_t gy = 1, QGroup qg = QGroup{}) {
    const size_t lds = (size_t)4 * Top::LDS_KEYS * sizeof(uint64_t);
    switch (t->dim / 64) {
#define MI_CASE(NCH)                                                                                   \
    case NCH:                                                                                          \
        allow_lds(knn_scan_kernel<NCH, Top>, lds);                                                     \
        hipLaunchKernelGGL((knn_scan_kernel<NCH, Top>), dim3(blocks, gy), dim3(256), lds, s, t->table, \
                           t->rows, d_q, k, lo, cand, (uint32_t*)nullptr, run_if, qg);                 \
        break;
        MI_CASE(1) MI_CASE(2) MI_CASE(4) MI_CASE(8) MI_CASE(12) MI_CASE(16)
#undef MI_CASE
        default: fail(MI_ERR_UNSUPPORTED, "dim %u: built for dim/64 in {1,2,4,8,12,16}", t->dim);
    }
    HIP_CHECK(hipGetLastError());
}

template <class Top>
void launch_merge(const uint64_t* in, uint32_t n_lists, uint32_t k, uint32_t lpb, uint64_t* out, uint32_t nq,
                  size_t in_stride, size_t out_stride, hipStream_t s, const uint32_t* run_if = nullptr, uint32_t run_stride = 0) {
    const size_t lds = (size_t)4 * Top::LDS_KEYS * sizeof(uint64_t);
    allow_lds(knn_merge_kernel<Top>, lds + 4096);
    const uint32_t blocks = (n_lists + lpb - 1) / lpb;
    hipLaunchKernelGGL((knn_merge_kernel<Top>), dim3(blocks, nq), dim3(256), lds, s, in, n_lists, k, lpb, out,
                       in_stride, out_stride, run_if, run_stride);
    HIP_CHECK(hipGetLastError());
}

// One pass: the kp <= 1024 smallest keys (> *lo if lo) of the shard, ascending, into keys_out[0..kp).
// gy > 1 (register lists only): the same pass for a group of gy queries (contiguous at d_q) in the same launches, query y
// gated by run_if[y * qg.flags], its keys to keys_out + y * qg.out
template <class Top>
void one_pass(mi_knn* t, const float* d_q, uint32_t kp, const uint64_t* lo, uint64_t* keys_out, hipStream_t s,
              const uint32_t* run_if = nullptr, uint32_t gy = 1, QGroup qg = QGroup{}) {
    const uint32_t bpc = Top::LDS_KEYS == 0 ? 4 : (Top::KP <= 256 ? 4 : 2);
    const uint64_t n_tiles = (t->rows + 63) / 64;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((uint64_t)t->n_cu * bpc, (n_tiles + 3) / 4);
    blocks = std::max(blocks, 1u);
    const uint32_t lists = blocks * 4;
    ensure(t, (void**)&t->d_cand, &t->cand_keys, (size_t)lists * kp * gy, sizeof(uint64_t));
    qg.lists = (uint64_t)lists * kp;
    launch_scan<Top>(t, d_q, kp, lo, t->d_cand, blocks, s, run_if, gy, qg);
    if constexpr (Top::LDS_KEYS != 0) {
        // k > 64: block-cooperative tree, 16 lists per block per level, ping-pong between d_tmp halves
        constexpr uint32_t LPB = 16;
        ensure(t, (void**)&t->d_tmp, &t->tmp_keys, (size_t)2 * ((lists + LPB - 1) / LPB) * kp, sizeof(uint64_t));
        const uint64_t* in = t->d_cand;
        uint32_t n = lists, lvl = 0;
        while (true) {
            const uint32_t nb = (n + LPB - 1) / LPB;
            uint64_t* out = nb == 1 ? keys_out : t->d_tmp + (size_t)(lvl & 1) * ((lists + LPB - 1) / LPB) * kp;
            hipLaunchKernelGGL((knn_merge_block_kernel<Top::KP>), dim3(nb), dim3(256), 0, s, in, n, kp, LPB, out);
            HIP_CHECK(hipGetLastError());
            if (nb == 1) break;
            in = out; n = nb; ++lvl;
        }
        return;
    }
    // tree: <= 64 lists per block, then one block
    if (lists <= 64) {
        launch_merge<Top>(t->d_cand, lists, kp, lists, keys_out, gy, qg.lists, qg.out, s, run_if, qg.flags);
    } else {
        const uint32_t lpb = 32, mid = (lists + lpb - 1) / lpb;
        ensure(t, (void**)&t->d_tmp, &t->tmp_keys, (size_t)mid * kp * gy, sizeof(uint64_t));
        launch_merge<Top>(t->d_cand, lists, kp, lpb, t->d_tmp, gy, qg.lists, (size_t)mid * kp, s, run_if, qg.flags);
        launch_merge<Top>(t->d_tmp, mid, kp, mid, keys_out, gy, (size_t)mid * kp, qg.out, s, run_if, qg.flags);
    }
}

// Two-stage exact search (knn_kernels.h "bf16 mirror as prefilter"): k <= 4096 (the single pass behind it must be one of
// the two gated forms: register lists, or the radix select), dim a multiple of 128, enough rows for the saved bytes to
// outweigh a dozen short launches.  Leaves the answer's keys in d_pref_keys and *fallback == 0, or
// *fallback == 1: the caller enqueues the single pass behind it, gated by that word.
constexpr uint64_t PREF_MIN_ROWS = 1u << 18;
bool prefilter_applies(const mi_knn* t, uint32_t k) {
    const uint32_t width = t->prefilter == 2 ? 256u : 128u;  // whole 256-byte chunks per mirror row
    return t->prefilter && (k <= 64 || (k <= 4096 && t->select_path)) && t->dim % width == 0 && t->rows >= PREF_MIN_ROWS;
}
// sample_shift > 0 (matrix-pipe form only): every 2^shift-th 16-row tile's keys also go, compactly, to t->d_skeys [nq][sample_stride]
template <int NCH>
void launch_coarse8_batched(mi_knn* t, const uint8_t* m8, const float* d_q, float e0, uint32_t nq, uint32_t blocks, hipStream_t s,
                            int sample_shift = 0, uint64_t sample_stride = 0) {
    if constexpr (NCH == 12) {  // built for dim 768, the width of the reference's table (server/src/clip.rs:140-143)
        if (t->batch_stage1_mfma) {
            // the group's stage 1 on the matrix pipe: exact int8 dot products against the query cut into three 7-bit digits
            hipLaunchKernelGGL(knn_query_digits_kernel, dim3(nq), dim3(256), 0, s, d_q, t->d_g8, (int)t->dim, t->d_digits, t->d_qs, t->d_rho8);
            constexpr int LDS = c8m_lds_bytes(NCH * 64);
            const uint64_t n_tiles = (t->rows + 15) / 16;
            const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)t->n_cu, (n_tiles + 3) / 4));   // 153 KB of LDS: one workgroup per CU
#define MI_C8M(NBLK, QPB)                                                                                                         \
    {                                                                                                                             \
        allow_lds(knn_scan_coarse8_mfma_kernel<NCH, NBLK, QPB>, LDS);                                                             \
        hipLaunchKernelGGL((knn_scan_coarse8_mfma_kernel<NCH, NBLK, QPB>), dim3(grid), dim3(256), LDS, s, m8, t->d_xx, t->d_scale8, \
                           t->d_cfac8, t->rows, t->d_digits, t->d_qs, (int)nq, e0, t->d_keys32, (uint64_t)t->cap,                 \
                           sample_shift ? t->d_skeys : (uint32_t*)nullptr, sample_shift, sample_stride);                          \
    }
            if (nq <= 5) MI_C8M(1, 5)
            else if (nq <= 10) MI_C8M(2, 5)
            else if (nq <= 15) MI_C8M(3, 5)
            else MI_C8M(4, 4)
#undef MI_C8M
            HIP_CHECK(hipGetLastError());
            return;
        }
        if (nq == 8)
            hipLaunchKernelGGL((knn_scan_coarse8_batched_kernel<NCH, 8>), dim3(blocks), dim3(256), 0, s, m8, t->d_xx, t->d_scale8, t->d_cfac8,
                               t->d_g8, t->rows, d_q, e0, t->d_keys32, (uint64_t)t->cap, t->d_rho8);
        else if (nq == 2)
            hipLaunchKernelGGL((knn_scan_coarse8_batched_kernel<NCH, 2>), dim3(blocks), dim3(256), 0, s, m8, t->d_xx, t->d_scale8, t->d_cfac8,
                               t->d_g8, t->rows, d_q, e0, t->d_keys32, (uint64_t)t->cap, t->d_rho8);
        else
            hipLaunchKernelGGL((knn_scan_coarse8_batched_kernel<NCH, 4>), dim3(blocks), dim3(256), 0, s, m8, t->d_xx, t->d_scale8, t->d_cfac8,
                               t->d_g8, t->rows, d_q, e0, t->d_keys32, (uint64_t)t->cap, t->d_rho8);
    } else {
        fail(MI_ERR_UNSUPPORTED, "the batched two-stage search is built for dim 768");
    }
}
// the strides of a group of gy queries on the shard's workspaces (every per-query buffer holds gy copies back to back)
constexpr uint32_t SEL_WORDS = 6 * SEL_BINS + 64;
constexpr uint32_t KNN_GROUP_MAX = 16;   // queries that share one pass over the byte mirror (matrix-pipe stage 1; the vector-ALU form: 8)
// how many of `left` >= 2 queries the next group takes: any count up to 16 on the matrix pipe, 8 / 4 / 2 on the vector ALU
uint32_t group_size(const mi_knn* t, uint32_t left) {
    if (t->batch_stage1_mfma) return std::min(left, KNN_GROUP_MAX);
    return left >= 8 ? 8 : left >= 4 ? 4 : 2;
}
QGroup group_of(const mi_knn* t, uint32_t gy) {
    QGroup g;
    if (gy <= 1) return g;
    g.q = t->dim; g.keys = t->cap; g.sel = SEL_WORDS; g.flags = 4; g.rows = (uint64_t)2 * PREF_CAP; g.coll = 4096; g.out = 4096; g.rho = 1;
    return g;
}
// nq_batch = 2 .. KNN_GROUP_MAX (16) (byte mirror, dim 768): a GROUP of queries, contiguous at d_q: one pass over the mirror writes the
// stage-1 keys of all of them, and every later kernel of the search — the selects, the collect, stage 2, the select over the
// candidates, the sort — runs ONCE for the group with the query as grid.y on per-query copies of the workspaces (QGroup).
// Returns the fallback word of query 0 (query y: + 4 y).
uint32_t* prefilter_pass(mi_knn* t, const float* d_q, uint32_t k, hipStream_t s, uint32_t nq_batch = 0) {
    const uint32_t gy = std::max(1u, nq_batch);
    const QGroup qg = group_of(t, gy);
    ensure(t, (void**)&t->d_keys32, &t->keys32_cap, (size_t)t->cap * gy, sizeof(uint32_t));
    ensure(t, (void**)&t->d_sel, &t->sel_cap, (size_t)SEL_WORDS * gy, sizeof(uint32_t));
    ensure(t, (void**)&t->d_cand, &t->cand_keys, (size_t)4096 * gy, sizeof(uint64_t));
    ensure(t, (void**)&t->d_pref_rows, &t->pref_rows_cap, (size_t)2 * PREF_CAP * gy, sizeof(uint32_t));  // rows, then their exact keys
    ensure(t, (void**)&t->d_pref_keys, &t->pref_keys_cap, (size_t)4096 * gy, sizeof(uint64_t));
    ensure(t, (void**)&t->d_pref_flag, &t->pref_flag_cap, (size_t)4 * gy, sizeof(uint32_t));
    const bool bytes = t->prefilter == 2;
    // The threshold from a SAMPLE of the keys (byte mirror, k <= 64): stage 1 also writes the keys of every 8th tile compactly,
    // the three histogram passes read that eighth, and the collect pass filters ALL keys against the sample's k-th smallest
    // upper bound — which is >= the k-th smallest over all rows, hence a valid (looser) threshold: a few times more candidates
    // for stage 2 (hundreds instead of tens at k = 10), three quarters of the select's traffic gone.  Tiles of 64 rows for one
    // query, of 16 rows for a group on the matrix pipe; the old vector-ALU group kernel does not sample.
    const uint32_t tile_rows = (nq_batch && t->batch_stage1_mfma) ? 16 : 64;
    const uint64_t n_tiles_s = (t->rows + tile_rows - 1) / tile_rows;
    // (groups only by default: for ONE query the 45 us of histogram reads it saves go into the larger stage 2 — measured equal at
    // k = 10, 2 % slower at k = 64; "prefilter_sample" = 2 samples for single queries too)
    int sample_shift = (bytes && t->pref_sample && k <= 64 && ((nq_batch >= 2 && t->batch_stage1_mfma) || (nq_batch == 0 && t->pref_sample >= 2))) ? 3 : 0;
    uint64_t n_sample = ((n_tiles_s + (1u << sample_shift) - 1) >> sample_shift) * tile_rows;
    if (sample_shift && (t->rows >> sample_shift) < std::max<uint64_t>(4096, 64ull * k)) { sample_shift = 0; n_sample = 0; }   // too small a table: the sample would be no guide
    // sized by the table's CAPACITY, not its rows: a table that grows by a chunk per step (the pipeline) must not reallocate
    // the sample — and wait for the searches in flight — every other query
    const uint64_t sample_stride = sample_shift ? ((((t->cap + tile_rows - 1) / tile_rows + (1u << sample_shift) - 1) >> sample_shift) * tile_rows + 63) / 64 * 64 : 0;
    if (sample_shift) ensure(t, (void**)&t->d_skeys, &t->skeys_cap, (size_t)sample_stride * gy, sizeof(uint32_t));
    const size_t mirror_elems = bytes ? ((size_t)t->cap * t->dim + 1) / 2 : (size_t)t->cap * t->dim;  // in uint16 units
    t->mirror_rows = std::min(t->mirror_rows, t->rows);
    if (bytes && t->g8_ready && t->rows >= 4 * std::max<uint64_t>(t->g8_rows, 1)) t->g8_ready = false;  // the table has grown 4x since the
                                                                                    // channel scales were taken: look again
    if (bytes && !t->g8_ready) t->mirror_rows = 0;  // new scales: every byte row changes
    // the table grew past the mirror's allocation: the rows mirrored so far move over, only the new ones are converted
    const size_t keep_m = bytes ? ((size_t)t->mirror_rows * t->dim + 1) / 2 : (size_t)t->mirror_rows * t->dim;
    ensure_keep(t, (void**)&t->d_mirror, &t->mirror_cap, mirror_elems, sizeof(uint16_t), keep_m);
    ensure_keep(t, (void**)&t->d_xx, &t->xx_cap, (size_t)t->cap, sizeof(float), (size_t)t->mirror_rows);
    if (bytes) {
        ensure_keep(t, (void**)&t->d_scale8, &t->scale8_cap, (size_t)t->cap, sizeof(float), (size_t)t->mirror_rows);
        ensure_keep(t, (void**)&t->d_cfac8, &t->cfac8_cap, (size_t)t->cap, sizeof(float), (size_t)t->mirror_rows);
        ensure(t, (void**)&t->d_rho8, &t->rho8_cap, (size_t)KNN_GROUP_MAX, sizeof(float));
        ensure(t, (void**)&t->d_digits, &t->digits_cap, (size_t)KNN_GROUP_MAX * 3 * t->dim + 64, sizeof(int8_t));   // a group's queries as int8 digits
        ensure(t, (void**)&t->d_qs, &t->qs_cap, (size_t)KNN_GROUP_MAX * 4, sizeof(float));
        ensure(t, (void**)&t->d_g8, &t->g8_cap, (size_t)2 * t->dim, sizeof(float));
        if (!t->g8_ready) {  // the channel scales: RMS per dimension over a sample spread over the table (any positive values are correct)
            const uint64_t sample = std::min<uint64_t>(t->rows, 1u << 16), stride = t->rows / sample;
            HIP_CHECK(hipMemsetAsync(t->d_g8 + t->dim, 0, t->dim * sizeof(float), s));
            hipLaunchKernelGGL(knn_channel_sumsq_kernel, dim3(256), dim3(256), 0, s, t->table, sample, stride, (int)t->dim, t->d_g8 + t->dim);
            hipLaunchKernelGGL(knn_channel_scale_kernel, dim3((t->dim + 255) / 256), dim3(256), 0, s, t->d_g8 + t->dim, sample,
                               (int)t->dim, t->d_g8);
            t->g8_ready = true;
            t->g8_rows = t->rows;
        }
    }
    const uint64_t n_tiles = (t->rows + 63) / 64;
    const uint32_t blocks = std::max<uint32_t>(1u, (uint32_t)std::min<uint64_t>((uint64_t)t->n_cu * 4, (n_tiles + 3) / 4));
    // (a group multiplies the grid by its queries: fewer blocks per query then, or every one of 16 x 2048 blocks pays the
    // histogram pick of the previous pass — a 2048-bin scan — for a few iterations of work)
    const uint32_t hb1 = (uint32_t)std::min<uint64_t>((uint64_t)t->n_cu * 8, (t->rows + 255) / 256);
    const uint32_t hb = std::max<uint32_t>(std::min<uint32_t>(hb1, (uint32_t)t->n_cu), (hb1 + gy - 1) / gy);
    uint32_t* flags = t->d_pref_flag;  // {candidate count, fallback, go}; not in d_sel: the selects below clear that
    uint32_t* key32 = t->d_pref_rows + PREF_CAP;
    uint32_t* count2 = t->d_sel + 6 * SEL_BINS;
    SelState* states = reinterpret_cast<SelState*>(t->d_sel + 6 * SEL_BINS + 4);
    const size_t sel_bytes = (size_t)SEL_WORDS * gy * sizeof(uint32_t);
    HIP_CHECK(hipMemsetAsync(t->d_sel, 0, sel_bytes, s));
    HIP_CHECK(hipMemsetAsync(flags, 0, (size_t)4 * gy * sizeof(uint32_t), s));
    static const uint32_t ring8_bpc = [] { const char* e = std::getenv("MI_KNN_RING_BPC"); return e ? (uint32_t)std::max(1, std::atoi(e)) : 2u; }();  // A/B
    const float e0 = 4.1f * (float)(t->dim + 8) * 0x1p-24f + 2e-6f;  // fp32 summations, norms, divisions
    const float eps = 0x1p-8f + e0;                                  // bf16: 8 significant bits, unit roundoff 2^-8
    if (bytes) {
        uint8_t* m8 = reinterpret_cast<uint8_t*>(t->d_mirror);
        const uint32_t* keys_q = t->d_keys32;   // query y: + y * cap (QGroup)
        // what the histogram passes read: every key, or the sample (its own stride between the queries of a group)
        const uint32_t* keys_h = sample_shift ? t->d_skeys : keys_q;
        const uint64_t n_h = sample_shift ? n_sample : t->rows;
        QGroup qgh = qg;
        if (sample_shift && gy > 1) qgh.keys = sample_stride;
        const uint32_t hbh = sample_shift ? std::max<uint32_t>(1u, (uint32_t)std::min<uint64_t>(hb, (n_h + 1023) / 1024)) : hb;
        switch (t->dim / 64) {
#define MI_CASE(NCH)                                                                                                     \
    case NCH:                                                                                                            \
        if (t->mirror_rows < t->rows) {                                                                                  \
            const uint64_t todo = t->rows - t->mirror_rows;                                                              \
            const uint32_t mb = (uint32_t)std::min<uint64_t>((uint64_t)t->n_cu * 8, (todo + 15) / 16);                   \
            hipLaunchKernelGGL((knn_mirror8_kernel<NCH>), dim3(std::max(mb, 1u)), dim3(256), 0, s, t->table, t->mirror_rows, \
                               t->rows, t->d_g8, m8, t->d_xx, t->d_scale8, t->d_cfac8);                                  \
            t->mirror_rows = t->rows;                                                                                    \
        }                                                                                                                \
        if (nq_batch == 0 && t->coarse_ring == 8)  /* 169 VGPRs: two workgroups per CU are resident, launch exactly those */ \
            hipLaunchKernelGGL((knn_scan_coarse8_kernel<NCH, 8>), dim3(std::min(blocks, (uint32_t)t->n_cu * ring8_bpc)), dim3(256), 0, s, m8, t->d_xx, t->d_scale8, \
                               t->d_cfac8, t->d_g8, t->rows, d_q, e0, t->d_keys32, t->d_rho8,                            \
                               sample_shift ? t->d_skeys : (uint32_t*)nullptr, sample_shift);                            \
        else if (nq_batch == 0)                                                                                          \
            hipLaunchKernelGGL((knn_scan_coarse8_kernel<NCH, 4>), dim3(blocks), dim3(256), 0, s, m8, t->d_xx, t->d_scale8, \
                               t->d_cfac8, t->d_g8, t->rows, d_q, e0, t->d_keys32, t->d_rho8,                            \
                               sample_shift ? t->d_skeys : (uint32_t*)nullptr, sample_shift);                            \
        else                                                                                                             \
            launch_coarse8_batched<NCH>(t, m8, d_q, e0, nq_batch, blocks, s, sample_shift, sample_stride);               \
        for (int p = 0; p < 3; ++p)                                                                                      \
            hipLaunchKernelGGL(knn_select_hist_kernel, dim3(hbh, gy), dim3(256), 0, s, keys_h, n_h, k, p, t->d_sel, states, \
                               (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr, qgh, 0);    \
        hipLaunchKernelGGL(knn_prefilter_collect8_kernel, dim3(hb, gy), dim3(256), 0, s, keys_q, t->d_cfac8, t->rows, k, \
                           t->d_sel, states, t->d_rho8, e0, PREF_CAP, t->d_pref_rows, flags, qg);                        \
        hipLaunchKernelGGL((knn_rescore_kernel<NCH>), dim3(t->n_cu * (gy > 1 ? 2 : 8), gy), dim3(256), 0, s, t->table, d_q, \
                           t->d_pref_rows, flags, PREF_CAP, key32, qg);                                                  \
        break;
            MI_CASE(4) MI_CASE(8) MI_CASE(12) MI_CASE(16)
#undef MI_CASE
            default: fail(MI_ERR_UNSUPPORTED, "dim %u: the byte prefilter is built for dim/64 in {4,8,12,16}", t->dim);
        }
    } else
    switch (t->dim / 64) {
#define MI_CASE(NCH)                                                                                                     \
    case NCH:                                                                                                            \
        if (t->mirror_rows < t->rows) {                                                                                  \
            const uint64_t todo = t->rows - t->mirror_rows;                                                              \
            const uint32_t mb = (uint32_t)std::min<uint64_t>((uint64_t)t->n_cu * 8, (todo + 15) / 16);                   \
            hipLaunchKernelGGL((knn_mirror_kernel<NCH>), dim3(std::max(mb, 1u)), dim3(256), 0, s, t->table, t->mirror_rows, \
                               t->rows, t->d_mirror, t->d_xx);                                                           \
            t->mirror_rows = t->rows;                                                                                    \
        }                                                                                                                \
        hipLaunchKernelGGL((knn_scan_coarse_kernel<NCH>), dim3(blocks), dim3(256), 0, s, t->d_mirror, t->d_xx, t->rows,  \
                           d_q, t->d_keys32);                                                                            \
        for (int p = 0; p < 3; ++p)                                                                                      \
            hipLaunchKernelGGL(knn_select_hist_kernel, dim3(hb), dim3(256), 0, s, t->d_keys32, t->rows, k, p, t->d_sel, states); \
        hipLaunchKernelGGL(knn_prefilter_collect_kernel, dim3(hb), dim3(256), 0, s, t->d_keys32, t->rows, k, t->d_sel,   \
                           states, 2.0f * eps, PREF_CAP, t->d_pref_rows, flags);                                         \
        hipLaunchKernelGGL((knn_rescore_kernel<NCH>), dim3(t->n_cu * 8), dim3(256), 0, s, t->table, d_q, t->d_pref_rows, \
                           flags, PREF_CAP, key32);                                                                      \
        break;
        MI_CASE(2) MI_CASE(4) MI_CASE(8) MI_CASE(12) MI_CASE(16)
#undef MI_CASE
        default: fail(MI_ERR_UNSUPPORTED, "dim %u: the prefilter is built for dim/64 in {2,4,8,12,16}", t->dim);
    }
    // the k smallest (exact key, row) pairs of the candidates: the radix select over the two arrays, unless there were too many
    HIP_CHECK(hipMemsetAsync(t->d_sel, 0, sel_bytes, s));
    const uint32_t* go = flags + 2;
    const uint32_t cb = (uint32_t)t->n_cu * 2;
    for (int p = 0; p < 6; ++p)
        hipLaunchKernelGGL(knn_select_hist_kernel, dim3(cb, gy), dim3(256), 0, s, key32, (uint64_t)PREF_CAP, k, p, t->d_sel, states, go, flags,
                           t->d_pref_rows, qg, 1);
    hipLaunchKernelGGL(knn_select_collect_kernel, dim3(cb, gy), dim3(256), 0, s, key32, (uint64_t)PREF_CAP, k, t->d_sel, states, t->d_cand, count2,
                       go, flags, t->d_pref_rows, qg, 1);
    hipLaunchKernelGGL(knn_select_sort_kernel, dim3(1, gy), dim3(1024), 0, s, t->d_cand, count2, k, t->d_pref_keys, go, qg);
    HIP_CHECK(hipGetLastError());
    return flags + 1;
}

// 64 < k <= 4096: every row's distance key, then the k smallest (distance, id) keys by radix select (knn_kernels.h)
// gy > 1: a group of gy queries (contiguous at d_q) in the same launches: query y gated by run_if[4 y], keys to keys_out + 4096 y
void select_pass(mi_knn* t, const float* d_q, uint32_t k, uint64_t* keys_out, hipStream_t s, const uint32_t* run_if = nullptr, uint32_t gy = 1) {
    const QGroup qg = group_of(t, gy);
    const uint64_t n_tiles = (t->rows + 63) / 64;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((uint64_t)t->n_cu * 4, (n_tiles + 3) / 4);
    blocks = std::max(blocks, 1u);
    ensure(t, (void**)&t->d_keys32, &t->keys32_cap, (size_t)t->cap * gy, sizeof(uint32_t));
    ensure(t, (void**)&t->d_sel, &t->sel_cap, (size_t)SEL_WORDS * gy, sizeof(uint32_t));
    ensure(t, (void**)&t->d_cand, &t->cand_keys, (size_t)4096 * gy, sizeof(uint64_t));
    HIP_CHECK(hipMemsetAsync(t->d_sel, 0, (size_t)SEL_WORDS * gy * sizeof(uint32_t), s));
    switch (t->dim / 64) {
#define MI_CASE(NCH)                                                                                          \
    case NCH:                                                                                                 \
        hipLaunchKernelGGL((knn_scan_kernel<NCH, WaveTopReg, 1>), dim3(blocks, gy), dim3(256), 0, s, t->table, t->rows, d_q, k, \
                           (const uint64_t*)nullptr, (uint64_t*)nullptr, t->d_keys32, run_if, qg);            \
        break;
        MI_CASE(1) MI_CASE(2) MI_CASE(4) MI_CASE(8) MI_CASE(12) MI_CASE(16)
#undef MI_CASE
        default: fail(MI_ERR_UNSUPPORTED, "dim %u: built for dim/64 in {1,2,4,8,12,16}", t->dim);
    }
    HIP_CHECK(hipGetLastError());
    const uint32_t hb = (uint32_t)std::min<uint64_t>((uint64_t)t->n_cu * 8, (t->rows + 255) / 256);
    uint32_t* count = t->d_sel + 6 * SEL_BINS;
    SelState* states = reinterpret_cast<SelState*>(t->d_sel + 6 * SEL_BINS + 4);  // 6 states of 24 bytes behind the counter
    for (int p = 0; p < 6; ++p)
        hipLaunchKernelGGL(knn_select_hist_kernel, dim3(hb, gy), dim3(256), 0, s, t->d_keys32, t->rows, k, p, t->d_sel, states, run_if,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr, qg, 0);
    hipLaunchKernelGGL(knn_select_collect_kernel, dim3(hb, gy), dim3(256), 0, s, t->d_keys32, t->rows, k, t->d_sel, states, t->d_cand, count, run_if,
                       (const uint32_t*)nullptr, (const uint32_t*)nullptr, qg, 0);
    hipLaunchKernelGGL(knn_select_sort_kernel, dim3(1, gy), dim3(1024), 0, s, t->d_cand, count, k, keys_out, run_if, qg);
    HIP_CHECK(hipGetLastError());
}

// ---- feedback of the two-stage search (handles.h: pref_*) -------------------------------------------------------
void pref_window(mi_knn* t) {
    t->pref_skip_left = mi_knn::PREF_SKIP;
    t->pref_consec = 0;
    t->pref_probing = false;
}
void pref_fold(mi_knn* t, uint32_t cand, uint32_t fell_back) {
    t->pref_consec = fell_back ? t->pref_consec + 1 : 0;
    if (t->pref_probing) {  // the two probes behind a skip window: both must have reported before anything is decided
        if (t->pref_reports_due && --t->pref_reports_due == 0) {
            if (t->pref_consec >= 2) pref_window(t);
            else t->pref_probing = false;
        }
        return;
    }
    // (only fallbacks count: a search that re-evaluates even two million rows — 2.7 ms over 10 M rows — still beats the single
    // pass; a rule on the candidate count, tried first, sent such corpora to the slower path)
    (void)cand;
    if (t->pref_consec >= 2) pref_window(t);
}
// fold what has arrived, oldest first, without waiting for anything
void pref_poll(mi_knn* t) {
    for (uint64_t q = t->pref_seq >= (uint64_t)mi_knn::PREF_RING ? t->pref_seq - mi_knn::PREF_RING : 0; q < t->pref_seq; ++q) {
        const int i = (int)(q % mi_knn::PREF_RING);
        if (!t->pref_ev_pending[i]) continue;
        if (hipEventQuery(t->pref_ev[i]) != hipSuccess) { (void)hipGetLastError(); break; }  // keep the order: stop at the first one still in flight
        t->pref_ev_pending[i] = false;
        pref_fold(t, t->h_pref_ring[2 * i], t->h_pref_ring[2 * i + 1]);
    }
}
// behind a two-stage search on `s`: {candidates, fell back} -> pinned memory, an event to say when
void pref_record(mi_knn* t, hipStream_t s) {
    if (!t->h_pref_ring) HIP_CHECK(hipHostMalloc((void**)&t->h_pref_ring, (size_t)mi_knn::PREF_RING * 2 * sizeof(uint32_t), hipHostMallocDefault));
    const int i = (int)(t->pref_seq % mi_knn::PREF_RING);
    if (t->pref_ev_pending[i]) {  // eight readbacks still in flight: this query goes unreported rather than waited for
        if (t->pref_probing && t->pref_reports_due && --t->pref_reports_due == 0) t->pref_probing = false;
        return;
    }
    if (!t->pref_ev[i]) HIP_CHECK(hipEventCreateWithFlags(&t->pref_ev[i], hipEventDisableTiming));
    HIP_CHECK(hipMemcpyAsync(t->h_pref_ring + 2 * i, t->d_pref_flag, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipEventRecord(t->pref_ev[i], s));
    t->pref_ev_pending[i] = true;
    ++t->pref_seq;
}
void pref_reset(mi_knn* t) {
    for (int i = 0; i < mi_knn::PREF_RING; ++i) {
        if (t->pref_ev_pending[i]) (void)hipEventSynchronize(t->pref_ev[i]);
        t->pref_ev_pending[i] = false;
    }
    t->pref_consec = t->pref_skip_left = 0;
    t->pref_probing = false;
    t->pref_probes_left = t->pref_reports_due = 0;
}

// lists a register-path pass writes (the grid of one_pass<WaveTopReg>)
uint32_t reg_pass_lists(const mi_knn* t) {
    const uint64_t n_tiles = (t->rows + 63) / 64;
    return 4 * std::max<uint32_t>(1u, (uint32_t)std::min<uint64_t>((uint64_t)t->n_cu * 4, (n_tiles + 3) / 4));
}

// One query (nq_batch = 0) or a group of nq_batch queries, contiguous at d_q, through the two stages; results to
// d_idx / d_dist [nq][k].  d_keys is sized by the caller (4096 keys per query).
void two_stage(mi_knn* t, const float* d_q, uint32_t k, uint64_t* d_idx, float* d_dist, hipStream_t s, uint32_t nq_batch) {
    const uint32_t gy = std::max(1u, nq_batch);
    const QGroup qg = group_of(t, gy);
    // every workspace of this search at its final size BEFORE the first launch: growing one later would free a buffer
    // that kernels already queued on `s` still use (ensure() only waits for EARLIER searches)
    if (k <= 64) {
        const uint32_t lists = reg_pass_lists(t);
        ensure(t, (void**)&t->d_cand, &t->cand_keys, std::max<size_t>((size_t)4096 * gy, (size_t)lists * k * gy), sizeof(uint64_t));
        if (lists > 64) ensure(t, (void**)&t->d_tmp, &t->tmp_keys, (size_t)((lists + 31) / 32) * k * gy, sizeof(uint64_t));
    }
    const uint32_t* fallback = prefilter_pass(t, d_q, k, s, nq_batch);
    // the single pass, every kernel of it returning at once unless the query's fallback word is set
    if (k <= 64) one_pass<WaveTopReg>(t, d_q, k, nullptr, t->d_keys, s, fallback, gy, qg);
    else select_pass(t, d_q, k, t->d_keys, s, fallback, gy);
    hipLaunchKernelGGL(knn_finalize_kernel, dim3((k + 255) / 256, gy), dim3(256), 0, s, t->d_keys, k, id_map(t),
                       d_idx, d_dist, (size_t)qg.out, (size_t)(gy > 1 ? k : 0), t->d_pref_keys, fallback, qg.flags);
    HIP_CHECK(hipGetLastError());
}

// nq = 2, 4 or 8 queries (contiguous at d_q) through the two stages with ONE pass over the byte mirror and ONE launch of
// every later kernel for the whole group (grid.y = query); bit-identical to nq single searches (the per-(row, query)
// arithmetic of stage 1 is the single kernel's; the selects and stage 2 are the single search's own kernels)
// A group counts as ONE search in the feedback of the two-stage search (handles.h: pref_*): it reports its first query's
// {candidates, fell back} words, it is one of the two probes behind a skip window, and a group of k <= 64 that is sent to
// the batched single pass instead counts the window down (k > 64 goes through search_one per query, which counts itself).
bool batched_two_stage_applies(mi_knn* t, uint32_t k) {
    if (!(t->prefilter == 2 && t->dim == 768 && prefilter_applies(t, k))) return false;
    if (!t->pref_adaptive) return true;
    pref_poll(t);
    if (t->pref_skip_left) {
        if (k <= 64) {
            if (--t->pref_skip_left == 0) { t->pref_probing = true; t->pref_probes_left = t->pref_reports_due = 2; t->pref_consec = 0; }
            ++t->pref_skipped;
        }
        return false;
    }
    if (t->pref_probing) {
        if (t->pref_probes_left) { --t->pref_probes_left; return true; }
        if (k <= 64) ++t->pref_skipped;
        return false;
    }
    return true;
}
void search_batched_two_stage(mi_knn* t, const float* d_q, uint32_t nq, uint32_t k, uint64_t* d_idx, float* d_dist, hipStream_t s) {
    ensure(t, (void**)&t->d_keys, &t->keys_cap, (size_t)4096 * nq, sizeof(uint64_t));
    t->last_prefiltered = true;
    two_stage(t, d_q, k, d_idx, d_dist, s, nq);
    if (t->pref_adaptive) pref_record(t, s);   // the group's first query speaks for it (d_pref_flag[0..1])
}

void search_one(mi_knn* t, const float* d_q, uint32_t k, uint64_t* d_idx, float* d_dist, hipStream_t s) {
    if (t->rows == 0) {  // nothing stored: k "none" entries
        hipLaunchKernelGGL(knn_finalize_kernel, dim3((k + 255) / 256, 1), dim3(256), 0, s,
                           (const uint64_t*)nullptr, k, id_map(t), d_idx, d_dist, (size_t)0, (size_t)0);
        HIP_CHECK(hipGetLastError());
        return;
    }
    const uint32_t passes = (k + 1023) / 1024;
    ensure(t, (void**)&t->d_keys, &t->keys_cap, (size_t)std::max<uint32_t>(passes * 1024, 4096), sizeof(uint64_t));
    t->last_prefiltered = prefilter_applies(t, k);
    if (t->last_prefiltered && t->pref_adaptive) {
        pref_poll(t);
        if (t->pref_skip_left) {  // this corpus has been defeating the mirror: the single pass alone, for a while
            if (--t->pref_skip_left == 0) { t->pref_probing = true; t->pref_probes_left = t->pref_reports_due = 2; t->pref_consec = 0; }
            ++t->pref_skipped;
            t->last_prefiltered = false;
        } else if (t->pref_probing) {
            if (t->pref_probes_left) --t->pref_probes_left;             // one of the two probes
            else { ++t->pref_skipped; t->last_prefiltered = false; }    // their reports are still on the way
        }
    }
    if (t->last_prefiltered) {
        two_stage(t, d_q, k, d_idx, d_dist, s, 0);
        if (t->pref_adaptive) pref_record(t, s);
        return;
    }
    if (k > 64 && k <= 4096 && t->select_path) {
        select_pass(t, d_q, k, t->d_keys, s);
        hipLaunchKernelGGL(knn_finalize_kernel, dim3((k + 255) / 256, 1), dim3(256), 0, s, t->d_keys, k, id_map(t),
                           d_idx, d_dist, (size_t)0, (size_t)0);
        HIP_CHECK(hipGetLastError());
        return;
    }
    for (uint32_t p = 0; p < passes; ++p) {
        const uint32_t kp = std::min(1024u, k - p * 1024);
        uint64_t* out = t->d_keys + (size_t)p * 1024;
        const uint64_t* lo = p ? out - 1 : nullptr;  // last key of the previous pass
        if (kp <= 64) one_pass<WaveTopReg>(t, d_q, kp, lo, out, s);
        else if (kp <= 256) one_pass<WaveTopLds<256>>(t, d_q, kp, lo, out, s);
        else one_pass<WaveTopLds<1024>>(t, d_q, kp, lo, out, s);
    }
    hipLaunchKernelGGL(knn_finalize_kernel, dim3((k + 255) / 256, 1), dim3(256), 0, s, t->d_keys, k, id_map(t),
                       d_idx, d_dist, (size_t)0, (size_t)0);
    HIP_CHECK(hipGetLastError());
}

template <int NQ>
void launch_batched(mi_knn* t, const float* d_q, uint32_t k, uint64_t* cand, uint32_t blocks, hipStream_t s) {
    switch (t->dim / 64) {
        case 12:
            hipLaunchKernelGGL((knn_scan_batched_kernel<12, NQ>), dim3(blocks), dim3(256), 0, s, t->table, t->rows,
                               d_q, k, cand);
            break;
        default: fail(MI_ERR_UNSUPPORTED, "batched search is built for dim 768 (got %u)", t->dim);
    }
    HIP_CHECK(hipGetLastError());
}

// nq in {2,4,8} queries, k <= 64, one table pass
void search_batched(mi_knn* t, const float* d_q, uint32_t nq, uint32_t k, uint64_t* d_idx, float* d_dist,
                    hipStream_t s) {
    const uint64_t n_tiles = (t->rows + 63) / 64;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((uint64_t)t->n_cu * 2, (n_tiles + 3) / 4);
    blocks = std::max(blocks, 1u);
    const uint32_t lists = blocks * 4;
    ensure(t, (void**)&t->d_cand, &t->cand_keys, (size_t)nq * lists * k, sizeof(uint64_t));
    ensure(t, (void**)&t->d_keys, &t->keys_cap, (size_t)std::max<uint32_t>(1024, nq * k), sizeof(uint64_t));
    if (nq == 2) launch_batched<2>(t, d_q, k, t->d_cand, blocks, s);
    else if (nq == 4) launch_batched<4>(t, d_q, k, t->d_cand, blocks, s);
    else launch_batched<8>(t, d_q, k, t->d_cand, blocks, s);
    const size_t cstride = (size_t)lists * k;
    if (lists <= 64) {
        launch_merge<WaveTopReg>(t->d_cand, lists, k, lists, t->d_keys, nq, cstride, k, s);
    } else {
        const uint32_t lpb = 32, mid = (lists + lpb - 1) / lpb;
        ensure(t, (void**)&t->d_tmp, &t->tmp_keys, (size_t)nq * mid * k, sizeof(uint64_t));
        launch_merge<WaveTopReg>(t->d_cand, lists, k, lpb, t->d_tmp, nq, cstride, (size_t)mid * k, s);
        launch_merge<WaveTopReg>(t->d_tmp, mid, k, mid, t->d_keys, nq, (size_t)mid * k, k, s);
    }
    hipLaunchKernelGGL(knn_finalize_kernel, dim3((k + 255) / 256, nq), dim3(256), 0, s, t->d_keys, k, id_map(t),
                       d_idx, d_dist, (size_t)k, (size_t)k);
    HIP_CHECK(hipGetLastError());
}

void check_search_args(const mi_knn* t, const void* q, uint32_t nq, uint32_t k, const void* idx, const void* dist) {
    if (!t) fail(MI_ERR_INVALID, "null table handle");
    if (nq && (!q || !idx || !dist)) fail(MI_ERR_INVALID, "null query/result pointer");
    if (k == 0) fail(MI_ERR_INVALID, "k must be >= 1");
}

// nq queries on the device (contiguous at d_q), in groups that share their passes: up to 16 per group through the two-stage
// search over the byte mirror, else 8 / 4 / 2 over the fp32 rows (k <= 64), else one by one; results [nq][k]
void search_many(mi_knn* t, const float* d_q, uint32_t nq, uint32_t k, uint64_t* d_idx, float* d_dist, hipStream_t s) {
    uint32_t u = 0;
    while (u < nq) {
        const uint32_t left = nq - u;
        if (left >= 2 && t->rows && batched_two_stage_applies(t, k)) {  // one pass over the byte mirror serves a group of up to 16 queries
            const uint32_t b2 = group_size(t, left);
            search_batched_two_stage(t, d_q + (size_t)u * t->dim, b2, k, d_idx + (size_t)u * k, d_dist + (size_t)u * k, s);
            u += b2;
            continue;
        }
        const uint32_t b = (k <= 64 && t->rows && t->dim == 768) ? (left >= 8 ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1) : 1;
        if (b == 1) search_one(t, d_q + (size_t)u * t->dim, k, d_idx + (size_t)u * k, d_dist + (size_t)u * k, s);
        else search_batched(t, d_q + (size_t)u * t->dim, b, k, d_idx + (size_t)u * k, d_dist + (size_t)u * k, s);
        u += b;
    }
}
}  // namespace

namespace mi {
hipStream_t knn_own_stream(mi_knn* t) { return own_stream(t); }
void knn_search_many(mi_knn* t, const float* d_q, uint32_t nq, uint32_t k, uint64_t* d_idx, float* d_dist, hipStream_t s) {
    search_many(t, d_q, nq, k, d_idx, d_dist, s);
}
void knn_grow(mi_knn* t, uint64_t want_rows) { grow(t, want_rows); }
void knn_search_one(mi_knn* t, const float* d_q, uint32_t k, uint64_t* d_idx, float* d_dist, hipStream_t s) {
    search_one(t, d_q, k, d_idx, d_dist, s);
}
void knn_truncate(mi_knn* t, uint64_t rows) {
    std::lock_guard<std::mutex> l(t->mu);
    if (rows >= t->rows) return;
    t->rows = rows;
    t->mirror_rows = std::min(t->mirror_rows, rows);
}
void knn_merge_lists_device(const uint64_t* d_idx_in, const float* d_dist_in, uint32_t lists, uint32_t nq, uint32_t k,
                            size_t idx_stride, size_t dist_stride, uint64_t* d_idx, float* d_dist, hipStream_t s) {
    const uint32_t threads = std::max<uint32_t>(lists * k, k);
    for (uint32_t u0 = 0; u0 < nq; u0 += 65535) {  // grid.y holds 65535 queries
        const uint32_t nu = std::min<uint32_t>(65535, nq - u0);
        hipLaunchKernelGGL(knn_merge_lists_kernel, dim3((threads + 255) / 256, nu), dim3(256), 0, s, d_idx_in + (size_t)u0 * k,
                           d_dist_in + (size_t)u0 * k, lists, k, idx_stride, dist_stride, d_idx + (size_t)u0 * k, d_dist + (size_t)u0 * k);
        HIP_CHECK(hipGetLastError());
    }
}
}  // namespace mi

extern "C" {

int mi_knn_create(uint32_t dim, int device, mi_knn** out) {
    return guarded([&] {
        if (!out) fail(MI_ERR_INVALID, "out is null");
        *out = nullptr;
        if (dim == 0 || dim % 64 != 0) fail(MI_ERR_UNSUPPORTED, "dim must be a positive multiple of 64 (got %u)", dim);
        DeviceGuard g(device);
        auto t = new mi_knn();
        t->device = device; t->dim = dim;
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        t->n_cu = prop.multiProcessorCount;
        HIP_CHECK(hipMalloc((void**)&t->d_q, (size_t)16 * dim * sizeof(float)));
        if (const char* e = std::getenv("MI_KNN_SELECT")) t->select_path = std::atoi(e) != 0;  // A/B hook, read at creation
        if (const char* e = std::getenv("MI_KNN_RING")) t->coarse_ring = std::atoi(e) == 8 ? 8 : 4;
        if (const char* e = std::getenv("MI_KNN_BATCH_STAGE1")) t->batch_stage1_mfma = std::atoi(e) != 0;  // A/B hook, read at creation
        *out = t;
    });
}

void mi_knn_free(mi_knn* t) {
    if (!t) return;
    (void)hipSetDevice(t->device);
    (void)hipDeviceSynchronize();
    t->writes.destroy();
    t->reads.destroy();
    if (t->stream) { (void)hipStreamSynchronize(t->stream); (void)hipStreamDestroy(t->stream); }
    for (void* p : {(void*)t->table, (void*)t->d_q, (void*)t->d_cand, (void*)t->d_tmp, (void*)t->d_keys,
                    (void*)t->d_idx, (void*)t->d_dist, (void*)t->d_keys32, (void*)t->d_sel, (void*)t->d_mirror,
                    (void*)t->d_xx, (void*)t->d_pref_rows, (void*)t->d_pref_keys, (void*)t->d_pref_flag, (void*)t->d_scale8,
                    (void*)t->d_cfac8, (void*)t->d_rho8, (void*)t->d_g8, (void*)t->d_digits, (void*)t->d_qs, (void*)t->d_skeys})
        if (p) (void)hipFree(p);
    for (hipEvent_t e : t->pref_ev)
        if (e) (void)hipEventDestroy(e);
    if (t->h_pref_ring) (void)hipHostFree(t->h_pref_ring);
    delete t;
}

int mi_knn_set_option(mi_knn* t, const char* key, int value) {
    return guarded([&] {
        if (!t || !key) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(t->mu);
        const std::string k(key);
        if (k == "prefilter") {
            // two-stage exact search: a mirror of the rows (1: bf16, + 50 % memory; 2: bytes, + 25 %; built by the next search
            // and kept up to date by every later one) prefilters, the fp32 rows decide; results are those of the single pass
            if (value < 0 || value > 2) fail(MI_ERR_INVALID, "prefilter: 0, 1 or 2 (got %d)", value);
            if (value != t->prefilter && t->d_mirror) {  // another form (or none): the mirror goes, the next search rebuilds
                DeviceGuard g(t->device);
                t->reads.sync();
                for (void** p : {(void**)&t->d_mirror, (void**)&t->d_xx, (void**)&t->d_scale8, (void**)&t->d_cfac8})
                    if (*p) { HIP_CHECK(hipFree(*p)); *p = nullptr; }
                t->mirror_cap = t->xx_cap = t->scale8_cap = t->cfac8_cap = 0;
                t->mirror_rows = 0;
                t->g8_ready = false;
            }
            if (value != t->prefilter) pref_reset(t);
            t->prefilter = value;
        } else if (k == "prefilter_sample") {
            // 1 (default): a GROUP of queries, k <= 64, over the byte mirror takes the collect threshold from a sample of the stage-1
            // keys (an eighth of the select's reads, a few times more rows for stage 2); 2: single queries too; 0: always from all
            // keys.  Same answers either way.
            if (value < 0 || value > 2) fail(MI_ERR_INVALID, "prefilter_sample must be 0, 1 or 2");
            t->pref_sample = value;
        } else if (k == "batch_stage1") {
            // 1 (default): the shared stage 1 of a group of queries on the matrix pipe (int8 MFMA, knn_scan_coarse8_mfma_kernel);
            // 0: the vector-ALU form (knn_scan_coarse8_batched_kernel).  Same answers either way.
            t->batch_stage1_mfma = value != 0;
        } else if (k == "prefilter_adaptive") {
            // 1 (default): a corpus that makes the two-stage search FALL BACK twice in a row is served by the single pass alone
            // for the next 64 single-query searches, then probed again with two queries (only fallbacks count: the candidate
            // count does not, pref_fold); 0: every query tries stage 1
            pref_reset(t);
            t->pref_adaptive = value != 0;
        } else {
            fail(MI_ERR_INVALID, "unknown option '%s' (known: prefilter, prefilter_adaptive, prefilter_sample, batch_stage1)", key);
        }
    });
}

int mi_knn_prefilter_stats(mi_knn* t, uint32_t* candidates, uint32_t* fell_back) {
    return guarded([&] {
        if (!t || !candidates || !fell_back) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(t->mu);
        *candidates = 0; *fell_back = 0;
        if (!t->last_prefiltered) return;
        DeviceGuard g(t->device);
        t->reads.sync();
        uint32_t w[2] = {0, 0};
        HIP_CHECK(hipMemcpy(w, t->d_pref_flag, sizeof w, hipMemcpyDeviceToHost));
        *candidates = w[0]; *fell_back = w[1];
    });
}

int mi_knn_prefilter_state(mi_knn* t, uint32_t out[4]) {
    return guarded([&] {
        if (!t || !out) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(t->device);
        t->reads.sync();   // everything enqueued has finished, so every readback has arrived
        pref_poll(t);
        out[0] = t->pref_skip_left;
        out[1] = t->pref_consec;
        out[2] = (uint32_t)std::min<uint64_t>(t->pref_skipped, 0xFFFFFFFFull);
        out[3] = (uint32_t)std::min<uint64_t>(t->g8_rows, 0xFFFFFFFFull);
    });
}

int mi_knn_set_base(mi_knn* t, uint64_t base) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        std::lock_guard<std::mutex> l(t->mu);
        t->base = base;
    });
}

int mi_knn_reserve(mi_knn* t, uint64_t rows) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(t->device);
        grow(t, rows);
    });
}

int mi_knn_size(const mi_knn* t, uint64_t* rows) {
    return guarded([&] {
        if (!t || !rows) fail(MI_ERR_INVALID, "null argument");
        *rows = t->rows;
    });
}

int mi_knn_append(mi_knn* t, const float* rows, uint64_t n) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        if (n == 0) return;
        if (!rows) fail(MI_ERR_INVALID, "rows is null");
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(t->device);
        own_stream(t);
        grow(t, t->rows + n);
        t->writes.begin(t->stream);
        HIP_CHECK(hipMemcpyAsync(t->table + t->rows * t->dim, rows, n * t->dim * sizeof(float), hipMemcpyHostToDevice,
                                 t->stream));
        t->writes.end(t->stream);
        HIP_CHECK(hipStreamSynchronize(t->stream));
        t->rows += n;
    });
}

int mi_knn_append_device(mi_knn* t, const float* d_rows, uint64_t n, void* stream) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        if (n == 0) return;
        if (!d_rows) fail(MI_ERR_INVALID, "d_rows is null");
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(t->device);
        hipStream_t s = stream ? (hipStream_t)stream : own_stream(t);
        grow(t, t->rows + n);  // a reallocation waits for the handle's work in flight and copies synchronously
        t->writes.begin(s);
        HIP_CHECK(hipMemcpyAsync(t->table + t->rows * t->dim, d_rows, n * t->dim * sizeof(float),
                                 hipMemcpyDeviceToDevice, s));
        // `rows` counts the new rows from now on; a search on any stream waits for this event first
        t->writes.end(s);
        t->rows += n;
    });
}

int mi_knn_append_synthetic(mi_knn* t, uint64_t seed, uint64_t first_row, uint64_t n) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        if (n == 0) return;
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(t->device);
        own_stream(t);
        grow(t, t->rows + n);
        const uint64_t key = [&] {  // synth.py: mix64(seed + GOLDEN)
            uint64_t z = seed + 0x9E3779B97F4A7C15ull;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        }();
        const float scale = (float)(1.0 / std::sqrt(1431655765.0));
        hipLaunchKernelGGL(gen_f32_kernel, dim3(t->n_cu * 8), dim3(256), 0, t->stream, t->table + t->rows * t->dim,
                           key, first_row * t->dim, n * t->dim, scale);
        HIP_CHECK(hipGetLastError());
        t->writes.end(t->stream);
        HIP_CHECK(hipStreamSynchronize(t->stream));
        t->rows += n;
    });
}

int mi_knn_get_rows(mi_knn* t, uint64_t first, uint64_t n, float* out) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        if (n == 0) return;
        if (!out) fail(MI_ERR_INVALID, "out is null");
        std::lock_guard<std::mutex> l(t->mu);
        if (first + n > t->rows) fail(MI_ERR_INVALID, "rows [%llu,%llu) out of range (size %llu)",
                                      (unsigned long long)first, (unsigned long long)(first + n),
                                      (unsigned long long)t->rows);
        DeviceGuard g(t->device);
        own_stream(t);
        t->writes.begin(t->stream);
        HIP_CHECK(hipMemcpyAsync(out, t->table + first * t->dim, n * t->dim * sizeof(float), hipMemcpyDeviceToHost,
                                 t->stream));
        HIP_CHECK(hipStreamSynchronize(t->stream));
    });
}

// ---- persistence of a shard: what SurrealDB's storage does for
// `image.embedding` (server/src/clip.rs:125-137).  File = 32-byte header {"MIKNNv01", u32 dim,
// u32 reserved, u64 rows, u64 base} + rows*dim little-endian f32, streamed through a 64 MiB
// pinned buffer so that neither side needs the table in host memory.
namespace {
struct KnnFileHeader { char magic[8]; uint32_t dim, reserved; uint64_t rows, base; };
static_assert(sizeof(KnnFileHeader) == 32, "header layout");
constexpr size_t IO_CHUNK = 64u << 20;
struct PinnedBuf {
    void* p = nullptr;
    explicit PinnedBuf(size_t b) { HIP_CHECK(hipHostMalloc(&p, b, hipHostMallocDefault)); }
    ~PinnedBuf() { (void)hipHostFree(p); }
};
struct File {
    FILE* f;
    File(const char* path, const char* mode) : f(std::fopen(path, mode)) {}
    ~File() { if (f) std::fclose(f); }
};
}  // namespace

int mi_knn_save(mi_knn* t, const char* path) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        if (!path) fail(MI_ERR_INVALID, "path is null");
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(t->device);
        own_stream(t);
        // crash-safe: everything goes to `<path>.tmp`, is flushed and fsync'ed, and only then renamed over
        // `path` — a full disk or a crash leaves the previous file, never a truncated one that reports success
        const std::string tmp = std::string(path) + ".tmp";
        {
            File f(tmp.c_str(), "wb");
            if (!f.f) fail(MI_ERR_IO, "cannot create %s", tmp.c_str());
            KnnFileHeader h{};
            std::memcpy(h.magic, "MIKNNv01", 8);
            h.dim = t->dim; h.rows = t->rows; h.base = t->base;
            if (std::fwrite(&h, sizeof h, 1, f.f) != 1) fail(MI_ERR_IO, "write to %s failed", tmp.c_str());
            const size_t total = (size_t)t->rows * t->dim * sizeof(float);
            if (total) {
                PinnedBuf buf(std::min(total, IO_CHUNK));
                t->writes.begin(t->stream);
                for (size_t off = 0; off < total; off += IO_CHUNK) {
                    const size_t n = std::min(IO_CHUNK, total - off);
                    HIP_CHECK(hipMemcpyAsync(buf.p, (const char*)t->table + off, n, hipMemcpyDeviceToHost, t->stream));
                    HIP_CHECK(hipStreamSynchronize(t->stream));
                    if (std::fwrite(buf.p, 1, n, f.f) != n) fail(MI_ERR_IO, "write to %s failed (disk full?)", tmp.c_str());
                }
            }
            if (std::fflush(f.f) != 0 || fsync(fileno(f.f)) != 0) fail(MI_ERR_IO, "flush of %s failed", tmp.c_str());
            FILE* fp = f.f;
            f.f = nullptr;
            if (std::fclose(fp) != 0) fail(MI_ERR_IO, "close of %s failed", tmp.c_str());
        }
        if (std::rename(tmp.c_str(), path) != 0) fail(MI_ERR_IO, "cannot rename %s to %s", tmp.c_str(), path);
        {   // the rename becomes durable with its directory
            const std::string p(path);
            const size_t slash = p.find_last_of('/');
            File d((slash == std::string::npos ? std::string(".") : slash == 0 ? std::string("/") : p.substr(0, slash)).c_str(), "r");
            if (d.f) (void)fsync(fileno(d.f));
        }
    });
}

int mi_knn_load(mi_knn* t, const char* path) {
    return guarded([&] {
        if (!t) fail(MI_ERR_INVALID, "null table handle");
        if (!path) fail(MI_ERR_INVALID, "path is null");
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(t->device);
        own_stream(t);
        File f(path, "rb");
        if (!f.f) fail(MI_ERR_IO, "cannot open %s", path);
        KnnFileHeader h{};
        if (std::fread(&h, sizeof h, 1, f.f) != 1 || std::memcmp(h.magic, "MIKNNv01", 8) != 0)
            fail(MI_ERR_IO, "%s is not a MIKNNv01 shard file", path);
        if (h.dim != t->dim) fail(MI_ERR_INVALID, "%s holds dim %u rows, the table has dim %u", path, h.dim, t->dim);
        if (t->rows == 0) t->base = h.base;  // an empty table takes the shard's id range
        else if (h.base != t->base + t->rows)
            fail(MI_ERR_INVALID, "%s starts at id %llu; the table (base %llu, %llu rows) would renumber its rows", path,
                 (unsigned long long)h.base, (unsigned long long)t->base, (unsigned long long)t->rows);
        const size_t total = (size_t)h.rows * t->dim * sizeof(float);
        if (total == 0) return;
        grow(t, t->rows + h.rows);
        PinnedBuf buf(std::min(total, IO_CHUNK));
        char* dst = (char*)(t->table + t->rows * t->dim);
        for (size_t off = 0; off < total; off += IO_CHUNK) {
            const size_t n = std::min(IO_CHUNK, total - off);
            if (std::fread(buf.p, 1, n, f.f) != n) fail(MI_ERR_IO, "%s is truncated", path);
            HIP_CHECK(hipMemcpyAsync(dst + off, buf.p, n, hipMemcpyHostToDevice, t->stream));
            HIP_CHECK(hipStreamSynchronize(t->stream));
        }
        t->rows += h.rows;
    });
}

int mi_knn_search_device(mi_knn* t, const float* d_q, uint32_t nq, uint32_t k, uint64_t* d_idx, float* d_dist,
                         void* stream) {
    return guarded([&] {
        check_search_args(t, d_q, nq, k, d_idx, d_dist);
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(t->device);
        hipStream_t s = stream ? (hipStream_t)stream : own_stream(t);
        t->writes.begin(s);
        t->reads.begin(s);
        for (uint32_t u = 0; u < nq; ++u)
            search_one(t, d_q + (size_t)u * t->dim, k, d_idx + (size_t)u * k, d_dist + (size_t)u * k, s);
        t->reads.end(s);
    });
}

int mi_knn_search_batched_device(mi_knn* t, const float* d_q, uint32_t nq, uint32_t k, uint64_t* d_idx,
                                 float* d_dist, void* stream) {
    return guarded([&] {
        check_search_args(t, d_q, nq, k, d_idx, d_dist);
        if (nq > 16) fail(MI_ERR_UNSUPPORTED, "batched search takes at most 16 queries (got %u)", nq);
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(t->device);
        hipStream_t s = stream ? (hipStream_t)stream : own_stream(t);
        t->writes.begin(s);
        t->reads.begin(s);
        search_many(t, d_q, nq, k, d_idx, d_dist, s);
        t->reads.end(s);
    });
}

// The merge every rank runs on what the all-gather left it, on the device: in = [lists][nq][k].
int mi_knn_merge_device(int device, const uint64_t* d_idx_in, const float* d_dist_in, uint32_t lists, uint32_t nq, uint32_t k,
                        uint64_t* d_idx, float* d_dist, void* stream) {
    return guarded([&] {
        if (k == 0) fail(MI_ERR_INVALID, "k must be >= 1");
        if (nq == 0) return;
        if (!d_idx || !d_dist || (lists && (!d_idx_in || !d_dist_in))) fail(MI_ERR_INVALID, "null argument");
        if ((uint64_t)lists * k > 0xFFFFFFFFull) fail(MI_ERR_UNSUPPORTED, "lists * k too large");
        DeviceGuard g(device);
        knn_merge_lists_device(d_idx_in, d_dist_in, lists, nq, k, (size_t)nq * k, (size_t)nq * k, d_idx, d_dist, (hipStream_t)stream);
    });
}

int mi_knn_search(mi_knn* t, const float* q, uint32_t nq, uint32_t k, uint64_t* idx, float* dist) {
    return guarded([&] {
        check_search_args(t, q, nq, k, idx, dist);
        if (nq == 0) return;
        std::lock_guard<std::mutex> l(t->mu);
        DeviceGuard g(t->device);
        own_stream(t);
        // groups of up to 16 queries: one upload, table passes of 8 / 4 / 2 queries where the register
        // path applies (k <= 64; same arithmetic per query as the single-query pass), one readback
        constexpr uint32_t GROUP = 16;
        ensure(t, (void**)&t->d_idx, &t->idx_cap, (size_t)GROUP * k, sizeof(uint64_t));
        ensure(t, (void**)&t->d_dist, &t->dist_cap, (size_t)GROUP * k, sizeof(float));
        t->writes.begin(t->stream);
        t->reads.begin(t->stream);
        for (uint32_t u0 = 0; u0 < nq; u0 += GROUP) {
            const uint32_t ng = std::min(GROUP, nq - u0);
            HIP_CHECK(hipMemcpyAsync(t->d_q, q + (size_t)u0 * t->dim, (size_t)ng * t->dim * sizeof(float),
                                     hipMemcpyHostToDevice, t->stream));
            uint32_t u = 0;
            while (u < ng) {
                const uint32_t left = ng - u;
                if (left >= 2 && batched_two_stage_applies(t, k)) {
                    const uint32_t b2 = group_size(t, left);
                    search_batched_two_stage(t, t->d_q + (size_t)u * t->dim, b2, k, t->d_idx + (size_t)u * k, t->d_dist + (size_t)u * k, t->stream);
                    u += b2;
                    continue;
                }
                const uint32_t b = (k <= 64 && t->rows && t->dim == 768) ? (left >= 8 ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1) : 1;
                if (b == 1) search_one(t, t->d_q + (size_t)u * t->dim, k, t->d_idx + (size_t)u * k, t->d_dist + (size_t)u * k, t->stream);
                else search_batched(t, t->d_q + (size_t)u * t->dim, b, k, t->d_idx + (size_t)u * k, t->d_dist + (size_t)u * k, t->stream);
                u += b;
            }
            HIP_CHECK(hipMemcpyAsync(idx + (size_t)u0 * k, t->d_idx, (size_t)ng * k * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                     t->stream));
            HIP_CHECK(hipMemcpyAsync(dist + (size_t)u0 * k, t->d_dist, (size_t)ng * k * sizeof(float), hipMemcpyDeviceToHost,
                                     t->stream));
            HIP_CHECK(hipStreamSynchronize(t->stream));
        }
        t->reads.pending = false;
    });
}

}  // extern "C"
