// knn_kernels.h — device code of the cosine K-nearest scan (Seam B).
//
// Replaces the SurrealQL statement `embedding <|K|> $reference` over an
// MTREE ... DIST COSINE TYPE F32 index (server/src/search.rs:70-86,
// server/src/clip.rs:140-143).  Bound: HBM — one pass reads N*dim*4 bytes.
//
// Layout in HBM: table[N][dim] f32 row-major (dim % 64 == 0), exactly the
// reference's stored `embedding: Vec<f32>` rows (server/src/search.rs:13-18).
//
// Scan geometry (wave64): a wave owns a tile of 64 consecutive rows (192 KiB at
// dim 768).  It walks the tile in 16 steps; in step `it` the 16-lane group g
// (= lane >> 4) streams row 16g+it: lane i of the group loads the 16 bytes at
// float offset 64t+4i of every 256-byte chunk t, so each wave-instruction is
// four fully used 256-byte segments.  Per lane four independent fmaf chains
// (one per float of the f32x4) run over t; they are then summed by the
// pairwise xor-butterfly (offsets 1,2 in-lane, 4,8,16,32 across the 16 lanes by
// DPP) that oracle/oracle.c:orc_dot2 restates — same operands, same order,
// same bits.  After the 16 steps lane L holds (q.x, x.x) of row 64*tile+L and
// evaluates  dist = 1 - dot / (sqrt(qq) * sqrt(xx))  once per 64 rows.
//
// Selection: every row becomes a 64-bit key  (monotone_u32(dist) << 32) | row,
// unique per row, so "k smallest keys" IS "(distance asc, id asc), NaN last".
// Each wave keeps its k best keys and a threshold (its current k-th key); a
// tile whose 64 keys all exceed the threshold costs one compare + ballot.
//   KP = 64        (k <= 64): the list lives one key per lane, merged by an
//                  in-register bitonic network over DPP/bpermute.
//   KP = 256, 1024 (k <= KP): per-wave LDS list + pending buffer, bitonic
//                  sort by the wave when the buffer fills.
// The per-wave lists (waves x k keys) go to HBM and are reduced by
// knn_merge_kernel in one or two tree levels; knn_finalize_kernel turns keys
// back into (uint64 id, f32 distance).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_c8;
// (device functions, not the builtins themselves, inside a kernel's lambdas: a builtin the HOST target does not know, met while
// the host pass instantiates the lambda, makes clang drop the kernel's host stub without a diagnostic)
__device__ __forceinline__ rsrc_c8 c8_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void c8_dma16_nt(rsrc_c8 r, uint32_t voff, void* lds_wave_base) {   // 16 bytes per lane, read-once hint
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 2);
}
__device__ __forceinline__ void c8_dma4(rsrc_c8 r, uint32_t voff, void* lds_wave_base) {       // 4 bytes per lane
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, voff, 0, 0, 0);
}

constexpr uint64_t KEY_MAX = 0xFFFFFFFFFFFFFFFFull;

// ---- DPP helpers (16-lane rows) -------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// After the xor-1 and xor-2 steps the four lanes of a quad agree, so the half-mirror
// (lane -> 7-lane) delivers the other quad's sum = the xor-4 partner's; likewise the
// row mirror (lane -> 15-lane) is the xor-8 partner once the 8-lane halves agree.
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]  : xor 1
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]  : xor 2
    v += dpp_mov<0x141>(v);  // row_half_mirror      : xor 4
    v += dpp_mov<0x140>(v);  // row_mirror           : xor 8
    return v;
}

// ---- keys -----------------------------------------------------------------------
__device__ __forceinline__ uint32_t dist_to_u32(float d) {
    uint32_t b = __float_as_uint(d);
    if (d != d) return 0xFFFFFFFFu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float u32_to_dist(uint32_t k) {
    if (k == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
__device__ __forceinline__ uint64_t make_key(float d, uint32_t row) {
    return ((uint64_t)dist_to_u32(d) << 32) | row;
}

__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src) {
    uint32_t lo = __shfl((uint32_t)v, src, 64), hi = __shfl((uint32_t)(v >> 32), src, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m) {
    uint32_t lo = __shfl_xor((uint32_t)v, m, 64), hi = __shfl_xor((uint32_t)(v >> 32), m, 64);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t umin64(uint64_t a, uint64_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint64_t umax64(uint64_t a, uint64_t b) { return a < b ? b : a; }

// ---- per-wave top-k, register form (one key per lane) -----------------------------
struct WaveTopReg {
    static constexpr int KP = 64;
    static constexpr int LDS_KEYS = 0;
    uint64_t best, thr;
    uint32_t k;
    int lane;

    __device__ void init(uint64_t*, uint32_t k_, int lane_) {
        best = KEY_MAX; thr = KEY_MAX; k = k_; lane = lane_;
    }
    __device__ static uint64_t sort_asc(uint64_t v, int lane) {
#pragma unroll
        for (int kk = 2; kk <= 64; kk <<= 1) {
#pragma unroll
            for (int j = kk >> 1; j > 0; j >>= 1) {
                const uint64_t o = shfl_xor64(v, j);
                const bool up = (lane & kk) == 0, lower = (lane & j) == 0;
                v = (lower == up) ? umin64(v, o) : umax64(v, o);
            }
        }
        return v;
    }
    __device__ static uint64_t merge_bitonic(uint64_t v, int lane) {
#pragma unroll
        for (int j = 32; j > 0; j >>= 1) {
            const uint64_t o = shfl_xor64(v, j);
            v = ((lane & j) == 0) ? umin64(v, o) : umax64(v, o);
        }
        return v;
    }
    // wave-collective: every lane offers one key (KEY_MAX = nothing)
    __device__ void offer(uint64_t key) {
        const bool pass = key < thr;
        if (__ballot(pass) == 0ull) return;
        uint64_t c = sort_asc(pass ? key : KEY_MAX, lane);
        c = shfl64(c, 63 - lane);
        best = merge_bitonic(umin64(best, c), lane);
        thr = shfl64(best, (int)k - 1);
    }
    __device__ void finish() {}
    // store the k best keys, ascending
    __device__ void store(uint64_t* out) const {
        if ((uint32_t)lane < k) out[lane] = best;
    }
    __device__ uint64_t lane_key(int) const { return best; }
};

// ---- per-wave top-k, LDS form (KP keys + KP pending) ------------------------------
template <int KP_>
struct WaveTopLds {
    static constexpr int KP = KP_;
    static constexpr int LDS_KEYS = 2 * KP_;
    uint64_t* buf;  // [0,KP) best ascending, [KP,2KP) pending
    uint64_t thr;
    uint32_t k, cnt;
    int lane;

    __device__ static void wave_sync() {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ void init(uint64_t* lds, uint32_t k_, int lane_) {
        buf = lds; k = k_; lane = lane_; cnt = 0; thr = KEY_MAX;
        for (int j = lane; j < 2 * KP; j += 64) buf[j] = KEY_MAX;
        wave_sync();
    }
    __device__ void merge() {
        wave_sync();
        // ascending bitonic sort of all 2KP keys by this wave
        for (int kk = 2; kk <= 2 * KP; kk <<= 1) {
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int p = lane; p < KP; p += 64) {
                    const int a = ((p & ~(j - 1)) << 1) | (p & (j - 1));  // bit j clear
                    const int b = a | j;
                    const uint64_t x = buf[a], y = buf[b];
                    const bool up = (a & kk) == 0;
                    if ((x > y) == up) { buf[a] = y; buf[b] = x; }
                }
                wave_sync();
            }
        }
        thr = buf[k - 1];
        thr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(thr >> 32)) << 32) |
              (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)thr);  // (the builtin returns int: no sign extension)
        wave_sync();
        for (int j = KP + lane; j < 2 * KP; j += 64) buf[j] = KEY_MAX;
        cnt = 0;
        wave_sync();
    }
    __device__ void offer(uint64_t key) {
        const bool pass = key < thr;
        const unsigned long long mask = __ballot(pass);
        if (mask == 0ull) return;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                        __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
        if (pass) buf[KP + cnt + rank] = key;
        cnt += (uint32_t)__popcll(mask);
        if (cnt + 64 > (uint32_t)KP) merge();
    }
    __device__ void finish() {
        if (cnt) merge(); else wave_sync();
    }
    __device__ void store(uint64_t* out) const {
        for (uint32_t j = lane; j < k; j += 64) out[j] = buf[j];
    }
    __device__ uint64_t lane_key(int j) const { return buf[j]; }
};

// ---- the scan ---------------------------------------------------------------------
// One fp32 fmaf per (row element, accumulator); NCH = dim / 64 chunks of 256 bytes.
template <int NCH>
struct RowAcc {
    float d0, d1, d2, d3, s0, s1, s2, s3;
    __device__ __forceinline__ void zero() { d0 = d1 = d2 = d3 = s0 = s1 = s2 = s3 = 0.0f; }
    __device__ __forceinline__ void step(const f32x4& q, const f32x4& x) {
        d0 = __builtin_fmaf(q.x, x.x, d0); d1 = __builtin_fmaf(q.y, x.y, d1);
        d2 = __builtin_fmaf(q.z, x.z, d2); d3 = __builtin_fmaf(q.w, x.w, d3);
        s0 = __builtin_fmaf(x.x, x.x, s0); s1 = __builtin_fmaf(x.y, x.y, s1);
        s2 = __builtin_fmaf(x.z, x.z, s2); s3 = __builtin_fmaf(x.w, x.w, s3);
    }
    __device__ __forceinline__ float dot() const { return row16_sum((d0 + d1) + (d2 + d3)); }
    __device__ __forceinline__ float sumsq() const { return row16_sum((s0 + s1) + (s2 + s3)); }
};

// A group of queries that share their launches (the batched two-stage search): query y = blockIdx.y of a kernel works on
// its OWN copy of every per-query buffer, found at base + y * stride.  Strides in elements of the buffer's type; all
// zero (and gridDim.y == 1) for a single query, which is then exactly the code that ran before the groups existed.
struct QGroup {
    uint32_t q = 0;       // query vectors (floats)
    uint64_t keys = 0;    // one 32-bit distance key per table row (u32)
    uint32_t sel = 0;     // histograms + SelState words + the collect counter (u32)
    uint32_t flags = 0;   // {candidate count, fallback, go, -} (u32)
    uint64_t rows = 0;    // candidate rows, then their exact keys (u32)
    uint32_t coll = 0;    // keys collected by a select (u64)
    uint32_t out = 0;     // sorted result keys (u64)
    uint64_t lists = 0;   // per-wave lists of the register-path scan (u64)
    uint32_t rho = 0;     // per-query scalars (floats)
};

// grid: any number of 256-thread blocks; wave w of the grid takes tiles w, w+W, ...
// cand: [gridDim.x*4][k] keys out.  lo_ptr (nullable): keys <= *lo_ptr are skipped
// (used when k > 1024 is served in several passes).
// KEYS = 0: per-wave top-k lists into cand.  KEYS = 1: no selection here, every row's 32-bit distance key goes to
// all_keys (one 256-byte store per tile: +2 % on the scan; holding the keys of 8 tiles in LDS and flushing them
// together was measured at +17 %) for knn_select_* below.
template <int NCH, class Top, int KEYS = 0>
__global__ __launch_bounds__(256, 2) void knn_scan_kernel(const float* __restrict__ table, uint64_t n_rows,
                                                       const float* __restrict__ q, uint32_t k,
                                                       const uint64_t* __restrict__ lo_ptr,
                                                       uint64_t* __restrict__ cand,
                                                       uint32_t* __restrict__ all_keys = nullptr,
                                                       const uint32_t* __restrict__ run_if = nullptr, QGroup qg = QGroup{}) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int DIM = NCH * 64;
    // query blockIdx.y of a group that shares this launch (QGroup: all strides 0 for a single query)
    q += (size_t)blockIdx.y * qg.q;
    if (cand) cand += (size_t)blockIdx.y * qg.lists;
    if (all_keys) all_keys += (size_t)blockIdx.y * qg.keys;
    if (run_if) run_if += (size_t)blockIdx.y * qg.flags;
    if (run_if && *run_if == 0u) return;  // the fallback behind a prefilter that did not need it
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const uint32_t wave = blockIdx.x * 4 + wib, n_waves = gridDim.x * 4;

    f32x4 qf[NCH];
#pragma unroll
    for (int t = 0; t < NCH; ++t) qf[t] = *reinterpret_cast<const f32x4*>(q + 64 * t + 4 * i);
    float sq;  // sqrt(q.q), same summation order as a row
    {
        RowAcc<NCH> a; a.zero();
#pragma unroll
        for (int t = 0; t < NCH; ++t) a.step(qf[t], qf[t]);
        sq = sqrtf(a.sumsq());
    }
    const uint64_t lo = lo_ptr ? *lo_ptr : 0ull;
    const bool use_lo = lo_ptr != nullptr;

    Top top;
    top.init(reinterpret_cast<uint64_t*>(smem) + (size_t)wib * Top::LDS_KEYS, k, lane);

    const uint64_t n_tiles = (n_rows + 63) >> 6;
    // two row buffers: the loads of step it+1 are in flight while step it is summed
    auto load_row = [&](f32x4 (&x)[NCH], uint64_t r) {
        r = r < n_rows ? r : n_rows - 1;
        const f32x4* p = reinterpret_cast<const f32x4*>(table + r * DIM) + i;
#pragma unroll
        for (int t = 0; t < NCH; ++t) x[t] = __builtin_nontemporal_load(p + 16 * t);
    };
    for (uint64_t tile = wave; tile < n_tiles; tile += n_waves) {
        float mydot = 0.0f, myxx = 1.0f;
        const uint64_t row0 = (tile << 6) + 16 * g;
        auto reduce_row = [&](const f32x4 (&x)[NCH], int it) {
            RowAcc<NCH> a; a.zero();
#pragma unroll
            for (int t = 0; t < NCH; ++t) a.step(qf[t], x[t]);
            const float d = a.dot(), s = a.sumsq();
            if (i == it) { mydot = d; myxx = s; }
        };
        f32x4 xa[NCH], xb[NCH];
        load_row(xa, row0);
#pragma unroll 1
        for (int it = 0; it < 16; it += 2) {
            load_row(xb, row0 + it + 1);
            reduce_row(xa, it);
            load_row(xa, row0 + (it + 2 < 16 ? it + 2 : 15));
            reduce_row(xb, it + 1);
        }
        const uint64_t r = (tile << 6) + lane;
        const float dist = 1.0f - mydot / (sq * sqrtf(myxx));
        if constexpr (KEYS == 1) {
            if (r < n_rows) all_keys[r] = dist_to_u32(dist);
            continue;
        }
        uint64_t key = r < n_rows ? make_key(dist, (uint32_t)r) : KEY_MAX;
        if (use_lo && key <= lo) key = KEY_MAX;
        top.offer(key);
    }
    if constexpr (KEYS != 0) return;
    top.finish();
    top.store(cand + (size_t)wave * k);
}

// ---- selection over ALL distance keys (64 < k <= 4096: the reference's K = 1000) ------------------------------
// The per-wave lists above stop paying when k approaches the rows a wave sees (1 M rows, k = 1000: 2048 lists of
// 488 rows each, everything "passes", the tree merge of 2048 x 1000 keys cost as much as the scan).  Instead the
// scan writes one 32-bit distance key per row (4 bytes per 3 KiB row read: +0.13 % traffic) and the k smallest
// 64-bit keys (distance << 32 | row: unique, so there is exactly one answer and ties break by id) are found by a
// most-significant-digit radix select: 6 digits (11, 11, 10 bits of the distance, then of the row id), one
// histogram pass over the keys per digit; the passes over the row digits return at once unless the k-th distance
// is shared by more rows than fit (exact duplicates).  Then one pass collects the keys <= the k-th and one block
// sorts them.  Independent of the insertion order and of k.
constexpr int SEL_BINS = 2048;
__device__ __forceinline__ int sel_shift(int p) { return p == 0 ? 53 : p == 1 ? 42 : p == 2 ? 32 : p == 3 ? 21 : p == 4 ? 10 : 0; }
__device__ __forceinline__ uint32_t sel_mask(int p) { return (p == 2 || p == 5) ? 0x3FFu : 0x7FFu; }

// The pick of pass p-1 is made by EVERY block of the kernel that runs pass p (all reach the same answer from the
// same complete histogram): state after p-1 passes comes from `states[p-2]` (written by the blocks of the previous
// kernel, all with the same value), the histogram of pass p-1 is scanned by the block (256 threads x 8 bins,
// Hillis-Steele over the partials), and the new state goes to states[p-1].
// prefix = the digits chosen so far (as the high bits of the key), k_rem = rank of the wanted key inside that prefix
// group (1-based), done = the group holds exactly k_rem keys (take them all), fixed = digits chosen.
struct SelState { uint64_t prefix; uint32_t k_rem; int done; int fixed; };
static_assert(sizeof(SelState) == 24, "layout of the state words in d_sel");
__device__ inline SelState sel_advance(const uint32_t* __restrict__ hist, SelState* __restrict__ states, int n_pass, uint32_t k,
                                       uint64_t n_rows) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t s_bin, s_rem, s_cnt;
    SelState st{0ull, k, 0, 0};
    if (n_rows < k) { st.done = 1; return st; }  // fewer rows than k: everything is taken
    if (n_pass == 0) return st;
    if (n_pass >= 2) st = states[n_pass - 2];
    if (st.done) {  // decided earlier: hand the verdict on (the next kernel reads states[n_pass - 1])
        if (threadIdx.x == 0) states[n_pass - 1] = st;
        return st;
    }
    const int p = n_pass - 1;
    const uint32_t* h = hist + p * SEL_BINS;
    uint32_t mine[8], loc = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { mine[j] = h[threadIdx.x * 8 + j]; loc += mine[j]; }
    part[threadIdx.x] = loc;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // inclusive scan of the 256 partials
        const uint32_t v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const uint32_t incl = part[threadIdx.x], excl = incl - loc;
    if (excl < st.k_rem && st.k_rem <= incl) {  // exactly one thread: the wanted rank falls into its 8 bins
        uint32_t acc = excl;
        int j = 0;
        while (j < 7 && acc + mine[j] < st.k_rem) acc += mine[j++];
        s_bin = threadIdx.x * 8 + j; s_rem = st.k_rem - acc; s_cnt = mine[j];
    }
    __syncthreads();
    st.prefix |= (uint64_t)s_bin << sel_shift(p);
    st.k_rem = s_rem;
    st.done = s_cnt == s_rem;  // the whole group is wanted: its lower digits do not matter
    st.fixed = p + 1;
    if (threadIdx.x == 0) states[p] = st;  // every block writes the same words
    __syncthreads();
    return st;
}

// pass `p`: histogram of digit p over the keys that match the prefix chosen so far.
// n_dev (nullable): the number of entries lives on the device (min with n_rows); row_of (nullable): the low word of
// entry r's 64-bit key is row_of[r] instead of r (the candidates of the two-stage search: ties still break by row id).
__global__ __launch_bounds__(256) void knn_select_hist_kernel(const uint32_t* __restrict__ keys, uint64_t n_rows, uint32_t k,
                                                              int p, uint32_t* __restrict__ hist, SelState* __restrict__ states,
                                                              const uint32_t* __restrict__ run_if = nullptr,
                                                              const uint32_t* __restrict__ n_dev = nullptr,
                                                              const uint32_t* __restrict__ row_of = nullptr, QGroup qg = QGroup{},
                                                              int keys_are_rows = 0) {
    __shared__ uint32_t lh[SEL_BINS];
    // query blockIdx.y of a group (QGroup); keys_are_rows: `keys` are the candidates' exact keys (they live behind the rows)
    keys += (size_t)blockIdx.y * (keys_are_rows ? qg.rows : qg.keys);
    hist += (size_t)blockIdx.y * qg.sel;
    states = reinterpret_cast<SelState*>(reinterpret_cast<uint32_t*>(states) + (size_t)blockIdx.y * qg.sel);
    if (run_if) run_if += (size_t)blockIdx.y * qg.flags;
    if (n_dev) n_dev += (size_t)blockIdx.y * qg.flags;
    if (row_of) row_of += (size_t)blockIdx.y * qg.rows;
    if (run_if && *run_if == 0u) return;
    if (n_dev) n_rows = min(n_rows, (uint64_t)*n_dev);
    const SelState st = sel_advance(hist, states, p, k, n_rows);
    if (st.done) return;
    for (int j = threadIdx.x; j < SEL_BINS; j += 256) lh[j] = 0;
    __syncthreads();
    const int sh = sel_shift(p);
    const uint32_t mk = sel_mask(p);
    const uint64_t hi_mask = p == 0 ? 0ull : ~0ull << sel_shift(p - 1);
    auto count_key = [&](uint64_t key, bool live) {
        const bool hit = live && (key & hi_mask) == st.prefix;
        const uint32_t bin = (uint32_t)(key >> sh) & mk;
        // cosine distances of a corpus crowd into a handful of leading digits: when every hit of the wave falls into
        // one bin (always, in pass 0) one lane adds the count instead of 64 lanes hammering one LDS word
        const unsigned long long hits = __ballot(hit);
        if (hits == 0ull) return;
        const uint32_t first = __builtin_amdgcn_readfirstlane(__shfl((int)bin, __ffsll((long long)hits) - 1, 64));
        if (__ballot(hit && bin != first) == 0ull) {
            if ((threadIdx.x & 63) == 0) atomicAdd(&lh[first], (uint32_t)__popcll(hits));
        } else if (hit) {
            atomicAdd(&lh[bin], 1u);
        }
    };
    // 4 consecutive entries per thread (one 16-byte load); the count is a multiple of nothing: tail by hand
    const uint64_t n4 = n_rows >> 2;
    for (uint64_t q4 = (uint64_t)blockIdx.x * 256 + threadIdx.x; q4 < ((n4 + 63) & ~63ull); q4 += (uint64_t)gridDim.x * 256) {
        const bool live = q4 < n4;
        uint4 v = {0u, 0u, 0u, 0u};
        const uint64_t r0 = q4 << 2;
        uint4 lo = {(uint32_t)r0, (uint32_t)(r0 + 1), (uint32_t)(r0 + 2), (uint32_t)(r0 + 3)};
        if (live) {
            v = reinterpret_cast<const uint4*>(keys)[q4];
            if (row_of) lo = reinterpret_cast<const uint4*>(row_of)[q4];
        }
        count_key(((uint64_t)v.x << 32) | lo.x, live);
        count_key(((uint64_t)v.y << 32) | lo.y, live);
        count_key(((uint64_t)v.z << 32) | lo.z, live);
        count_key(((uint64_t)v.w << 32) | lo.w, live);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {  // the last n_rows % 4 entries
        const uint64_t r = (n4 << 2) + threadIdx.x;
        const bool live = threadIdx.x < (n_rows & 3);
        count_key(live ? (((uint64_t)keys[r] << 32) | (row_of ? row_of[r] : (uint32_t)r)) : 0ull, live);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < SEL_BINS; j += 256)
        if (lh[j]) atomicAdd(&hist[p * SEL_BINS + j], lh[j]);
}

// every key <= the k-th smallest goes to out (unordered); count = how many (== min(k, n_rows))
__global__ __launch_bounds__(256) void knn_select_collect_kernel(const uint32_t* __restrict__ keys, uint64_t n_rows, uint32_t k,
                                                                 const uint32_t* __restrict__ hist, SelState* __restrict__ states,
                                                                 uint64_t* __restrict__ out, uint32_t* __restrict__ count,
                                                                 const uint32_t* __restrict__ run_if = nullptr,
                                                                 const uint32_t* __restrict__ n_dev = nullptr,
                                                                 const uint32_t* __restrict__ row_of = nullptr, QGroup qg = QGroup{},
                                                                 int keys_are_rows = 0) {
    keys += (size_t)blockIdx.y * (keys_are_rows ? qg.rows : qg.keys);
    hist += (size_t)blockIdx.y * qg.sel;
    states = reinterpret_cast<SelState*>(reinterpret_cast<uint32_t*>(states) + (size_t)blockIdx.y * qg.sel);
    out += (size_t)blockIdx.y * qg.coll;
    count += (size_t)blockIdx.y * qg.sel;   // the counter lives in the select block
    if (run_if) run_if += (size_t)blockIdx.y * qg.flags;
    if (n_dev) n_dev += (size_t)blockIdx.y * qg.flags;
    if (row_of) row_of += (size_t)blockIdx.y * qg.rows;
    if (run_if && *run_if == 0u) return;
    if (n_dev) n_rows = min(n_rows, (uint64_t)*n_dev);
    const SelState st = sel_advance(hist, states, 6, k, n_rows);
    // the k-th key: the chosen digits; when a whole group was taken its lower digits are free (all ones);
    // fewer rows than k: every key
    uint64_t T = KEY_MAX;
    if (n_rows >= k) T = st.fixed == 6 ? st.prefix : (st.prefix | ((1ull << sel_shift(st.fixed - 1)) - 1ull));
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * 256) {
        const uint64_t key = ((uint64_t)keys[r] << 32) | (row_of ? row_of[r] : (uint32_t)r);
        if (key <= T) {
            const uint32_t at = atomicAdd(count, 1u);
            if (at < k) out[at] = key;
        }
    }
}

// one block: the (<= 4096) collected keys ascending into out[0, k), KEY_MAX behind them
__global__ __launch_bounds__(1024) void knn_select_sort_kernel(const uint64_t* __restrict__ in, const uint32_t* __restrict__ count,
                                                               uint32_t k, uint64_t* __restrict__ out,
                                                               const uint32_t* __restrict__ run_if = nullptr, QGroup qg = QGroup{}) {
    __shared__ uint64_t buf[4096];
    in += (size_t)blockIdx.y * qg.coll;
    count += (size_t)blockIdx.y * qg.sel;
    out += (size_t)blockIdx.y * qg.out;
    if (run_if) run_if += (size_t)blockIdx.y * qg.flags;
    if (run_if && *run_if == 0u) return;
    const uint32_t n = min(*count, k);
    int np = 64;
    while ((uint32_t)np < k) np <<= 1;  // the power of two that holds k (<= 4096)
    for (int j = threadIdx.x; j < np; j += 1024) buf[j] = (uint32_t)j < n ? in[j] : KEY_MAX;
    __syncthreads();
    for (int kk = 2; kk <= np; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int p = threadIdx.x; p < np / 2; p += 1024) {
                const int a = ((p & ~(j - 1)) << 1) | (p & (j - 1)), b = a | j;
                const uint64_t x = buf[a], y = buf[b];
                const bool up = (a & kk) == 0;
                if ((x > y) == up) { buf[a] = y; buf[b] = x; }
            }
            __syncthreads();
        }
    for (uint32_t j = threadIdx.x; j < k; j += 1024) out[j] = buf[j];
}

// ---- two-stage EXACT search: a bf16 mirror as prefilter (opt-in, k <= 4096) --------------------------------------
// The scan above is bound by the bytes of the table; 288 GB of HBM per GPU leave room to keep the rows a second time
// at half the width.  Stage 1 scans the bf16 mirror (dim*2 bytes per row + a stored f32 norm) and writes every row's
// COARSE distance key; the k-th smallest coarse distance t is found by the radix select's three distance passes;
// every row with coarse distance <= t + 2 eps becomes a candidate; stage 2 evaluates the candidates from the fp32
// table with the arithmetic of knn_scan_kernel, bit for bit, and the k smallest of THOSE keys are the answer.
//
// Why it is exact.  With eps >= |coarse - exact| for every row: the k rows of smallest coarse distance have exact
// distances <= t + eps, so the exact k-th distance is <= t + eps, so every row of the exact answer has coarse distance
// <= t + 2 eps and is a candidate.  The bound: x~ = bf16(x) rounds to nearest (8 significant bits: unit roundoff 2^-8),
// |x~_j - x_j| <= 2^-8 |x_j|, hence |q.x~ - q.x| <= 2^-8 sum|q_j x_j| <= 2^-8 |q| |x| (Cauchy-Schwarz), i.e. 2^-8 on the
// cosine; the fp32 summations of both sides (any order) add <= 2 gamma_n each with gamma_n = (n + 8) 2^-24, the norms
// (stored vs recomputed) another 2 gamma_n, the divisions / square roots / subtraction a few ulp of O(1):
// eps = 2^-8 + 4.1 (dim + 8) 2^-24 + 2e-6 (4.10e-3 at dim 768), independent of the data; tests/test_prefilter_bound.py
// drives rows built to sit at the rounding's worst case against it.  Rows the bound does not cover (a non-finite or > 3e38 element, a
// squared norm outside [1e-30, 1e30]) are marked in the mirror and are always candidates.  More candidates than the
// buffer holds (4 M) => the single-pass scan runs instead;
// it is enqueued behind stage 2 either way and returns at once when it is not needed (no host round trip).
#ifndef PREF_DEPTH
#define PREF_DEPTH 3
#endif
constexpr uint32_t PREF_MARK = 0xFFFFFFFEu;  // coarse key of a marked row: no distance maps to it (NaN is 0xFFFFFFFF), it ranks behind every real one
constexpr uint32_t PREF_CAP = 1u << 22;  // candidates stage 2 accepts (4 M rows = 12.9 GB of fp32 rows at dim 768: two fifths of a 10 M-row pass)

// rows [first, end) of the table -> bf16 mirror rows + stored squared norms (-1 = "always a candidate")
template <int NCH>
__global__ __launch_bounds__(256) void knn_mirror_kernel(const float* __restrict__ table, uint64_t first, uint64_t end,
                                                         uint16_t* __restrict__ mirror, float* __restrict__ xx) {
    constexpr int DIM = NCH * 64;
    const int lane = threadIdx.x & 63, i = lane & 15;
    const uint64_t group = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4, n_groups = ((uint64_t)gridDim.x * 256) >> 4;
    for (uint64_t r0 = first + group; r0 < ((end - first + n_groups - 1) / n_groups) * n_groups + first; r0 += n_groups) {
        const bool live = r0 < end;  // (whole 16-lane groups stay in the loop: row16_sum is a cross-lane operation)
        const uint64_t r = live ? r0 : end - 1;
        const f32x4* p = reinterpret_cast<const f32x4*>(table + r * DIM) + i;
        float s = 0.0f;
        bool bad = false;
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const f32x4 v = p[16 * t];
            const float e[4] = {v.x, v.y, v.z, v.w};
            uint32_t b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bad |= !(fabsf(e[j]) <= 3.0e38f);  // NaN, inf, and what bf16 would round to inf
                s = __builtin_fmaf(e[j], e[j], s);
                const uint32_t u = __float_as_uint(e[j]);
                b[j] = (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;  // round to nearest even
            }
            if (live) *reinterpret_cast<uint2*>(mirror + r * DIM + 64 * t + 4 * i) = make_uint2(b[0] | (b[1] << 16), b[2] | (b[3] << 16));
        }
        s = row16_sum(s);
        const unsigned long long bm = __ballot(bad);
        const bool any_bad = ((bm >> (lane & 48)) & 0xFFFFull) != 0ull;
        if (live && i == 0) xx[r] = (any_bad || !(s >= 1.0e-30f && s <= 1.0e30f)) ? -1.0f : s;
    }
}

// stage 1: every row's coarse distance key (PREF_MARK for the marked rows: they must not count towards the k-th) from the mirror; geometry of knn_scan_kernel with
// rows of dim * 2 bytes: lane i of a 16-lane group loads the 16 bytes (8 bf16) at element 128 u + 8 i of its row
template <int NCH>
__global__ __launch_bounds__(256, 2) void knn_scan_coarse_kernel(const uint16_t* __restrict__ mirror, const float* __restrict__ xx,
                                                              uint64_t n_rows, const float* __restrict__ q,
                                                              uint32_t* __restrict__ all_keys) {
    static_assert(NCH % 2 == 0, "rows of whole 256-byte bf16 chunks");
    constexpr int DIM = NCH * 64, U = NCH / 2;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const uint32_t wave = blockIdx.x * 4 + wib, n_waves = gridDim.x * 4;
    float qf[U][8];
    float sq;
    {
        float s = 0.0f;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                qf[u][e] = q[128 * u + 8 * i + e];
                s = __builtin_fmaf(qf[u][e], qf[u][e], s);
            }
        sq = sqrtf(row16_sum(s));
    }
    const uint64_t n_tiles = (n_rows + 63) >> 6;
    auto load_row = [&](u32x4 (&x)[U], uint64_t r) {
        r = r < n_rows ? r : n_rows - 1;
        const u32x4* p = reinterpret_cast<const u32x4*>(mirror + r * DIM) + i;
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p + 16 * u);
    };
    for (uint64_t tile = wave; tile < n_tiles; tile += n_waves) {
        float mydot = 0.0f;
        const uint64_t row0 = (tile << 6) + 16 * g;
        auto reduce_row = [&](const u32x4 (&x)[U], int it) {
            float a[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t w[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[2 * j] = __builtin_fmaf(qf[u][2 * j], __uint_as_float(w[j] << 16), a[2 * j]);
                    a[2 * j + 1] = __builtin_fmaf(qf[u][2 * j + 1], __uint_as_float(w[j] & 0xFFFF0000u), a[2 * j + 1]);
                }
            }
            const float d = row16_sum(((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7])));
            if (i == it) mydot = d;
        };
        // PREF_DEPTH row buffers: PREF_DEPTH - 1 rows of loads in flight while one is summed (rows of dim * 2 bytes are half
        // the fp32 scan's: two buffers left 144 KB per CU in flight against its 288 KB)
        u32x4 xr[PREF_DEPTH][U];
#pragma unroll
        for (int d = 0; d < PREF_DEPTH - 1; ++d) load_row(xr[d], row0 + d);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if (it + PREF_DEPTH - 1 < 16) load_row(xr[(it + PREF_DEPTH - 1) % PREF_DEPTH], row0 + it + PREF_DEPTH - 1);
            reduce_row(xr[it % PREF_DEPTH], it);
        }
        const uint64_t r = (tile << 6) + lane;
        if (r < n_rows) {
            const float s = xx[r];
            all_keys[r] = s < 0.0f ? PREF_MARK : dist_to_u32(1.0f - mydot / (sq * sqrtf(s)));
        }
    }
}

// the candidates: rows whose coarse key is <= key(t + band), t = the k-th smallest coarse distance (from the three
// distance passes of knn_select_hist_kernel); count may exceed cap (stage 2 then hands over to the single-pass scan)
__global__ __launch_bounds__(256) void knn_prefilter_collect_kernel(const uint32_t* __restrict__ keys, uint64_t n_rows, uint32_t k,
                                                                    const uint32_t* __restrict__ hist, SelState* __restrict__ states,
                                                                    float band, uint32_t cap, uint32_t* __restrict__ cand_rows,
                                                                    uint32_t* __restrict__ count) {
    const SelState st = sel_advance(hist, states, 3, k, n_rows);
    uint32_t B = 0xFFFFFFFFu;  // fewer rows than k, or a NaN at rank k: everything
    if (n_rows >= k && st.fixed >= 1) {
        const uint64_t T = st.prefix | ((1ull << sel_shift(st.fixed - 1)) - 1ull);  // undecided digits: all ones (an upper bound)
        const uint32_t T32 = (uint32_t)(T >> 32);
        const float t = u32_to_dist(T32);
        if (T32 < PREF_MARK && t == t) B = dist_to_u32(t + band);  // (fewer than k unmarked rows with a real distance: everything)
    }
    // (whole waves stay in the loop: the append is one atomic per wave)
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t r0 = (uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); r0 < n_rows; r0 += stride) {
        const uint64_t r = r0 + (threadIdx.x & 63);
        const uint32_t key = r < n_rows ? keys[r] : 0xFFFFFFFFu;
        const bool take = r < n_rows && (key <= B || key == PREF_MARK);
        const unsigned long long m = __ballot(take);
        if (m == 0ull) continue;
        uint32_t base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(count, (uint32_t)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (take) {
            const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (at < cap) cand_rows[at] = (uint32_t)r;
        }
    }
}

// ---- the 8-bit form of stage 1 ("prefilter" = 2): a quarter of the single pass's bytes ---------------------------------
// Rows as unsigned bytes u_j = 128 + round(x_j * 127 / a), a = max_j |x_j| (so x^_j = s (u_j - 128), s = a / 127, and
// |x^_j - x_j| <= s / 2), with three floats per row: the scale s, the squared norm, and c = 0.53 s / |x|.  The error of
// the coarse cosine is bounded PER ROW and per query:  |q.x^ - q.x| <= (s / 2) |q|_1, i.e. on the cosine
// eps_r = c_r * rho, rho = |q|_1 / |q|_2 (0.53 instead of 0.5: the fp32 summation of sum q_j u_j, whose terms are up to
// 255 |q_j|, and the rounding of x_j * 127 / a), plus the same 4.1 (dim + 8) 2^-24 + 2e-6 as above.  The key stored per
// row is the UPPER bound U_r = d~_r + eps_r: the k-th smallest of them bounds the exact k-th distance from above, and a
// row can belong to the answer only if its LOWER bound U_r - 2 eps_r is not beyond it.  Random rows sit two orders of
// magnitude inside this bound (their byte errors cancel), so the band is wide for what it needs to catch: ~10^2 (k = 10)
// to ~10^4 (k = 1000) candidates on 10 M iid rows; stage 2 is the same as for the bf16 mirror.
// Channels are normalised first: x'_j = x_j / g_j with one g per dimension for the whole shard (the channel's RMS over the
// first rows mirrored; any positive g is correct, it only decides how tight the bound is) and q'_j = q_j g_j, so that
// q.x = q'.x'.  Without it a few dimensions two orders of magnitude above the rest — what trained CLIP embeddings have —
// set every row's scale, the bytes of all other dimensions collapse around 128 and eps_r grows to 0.2 (measured: 46 % of
// a 1 M-row corpus became candidates); with it a = max_j |x'_j|, rho = |q'|_1 / |q|_2 and the bound is back at a few 1e-3.
__global__ __launch_bounds__(256) void knn_channel_sumsq_kernel(const float* __restrict__ table, uint64_t n_rows, uint64_t row_stride,
                                                                int dim, float* __restrict__ acc) {
    // a sample of n_rows rows, `row_stride` apart (spread over the whole table): sample blockIdx.x, + gridDim.x, ...;
    // thread j owns channels j, j + 256, ... (coalesced across the block)
    for (int c = threadIdx.x; c < dim; c += 256) {
        float s = 0.0f;
        for (uint64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
            const float v = table[r * row_stride * dim + c];
            if (fabsf(v) <= 1.0e18f) s = __builtin_fmaf(v, v, s);  // (non-finite and absurd values do not get a vote)
        }
        atomicAdd(&acc[c], s);
    }
}
__global__ void knn_channel_scale_kernel(const float* __restrict__ acc, uint64_t n_rows, int dim, float* __restrict__ g) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= dim) return;
    const float rms = sqrtf(acc[c] / (float)n_rows);
    g[c] = (rms >= 1.0e-18f && rms <= 1.0e18f) ? rms : 1.0f;
}

__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}

template <int NCH>
__global__ __launch_bounds__(256) void knn_mirror8_kernel(const float* __restrict__ table, uint64_t first, uint64_t end,
                                                          const float* __restrict__ gch, uint8_t* __restrict__ mirror,
                                                          float* __restrict__ xx, float* __restrict__ scale,
                                                          float* __restrict__ cfac) {
    constexpr int DIM = NCH * 64;
    const int lane = threadIdx.x & 63, i = lane & 15;
    const uint64_t group = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 4, n_groups = ((uint64_t)gridDim.x * 256) >> 4;
    f32x4 ginv[NCH];  // 1 / g of this lane's channels
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gch + 64 * t + 4 * i);
        ginv[t] = f32x4{1.0f / gv.x, 1.0f / gv.y, 1.0f / gv.z, 1.0f / gv.w};
    }
    for (uint64_t r0 = first + group; r0 < ((end - first + n_groups - 1) / n_groups) * n_groups + first; r0 += n_groups) {
        const bool live = r0 < end;
        const uint64_t r = live ? r0 : end - 1;
        const f32x4* p = reinterpret_cast<const f32x4*>(table + r * DIM) + i;
        f32x4 v[NCH];
        float s2 = 0.0f, a = 0.0f;
        bool bad = false;
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            v[t] = p[16 * t];
            const float e[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bad |= !(fabsf(e[j]) <= 3.0e38f);
                s2 = __builtin_fmaf(e[j], e[j], s2);  // the norm is the row's own: the cosine is
            }
            v[t] = v[t] * ginv[t];                      // the bytes are the normalised channels'
            a = fmaxf(a, fmaxf(fmaxf(fabsf(v[t].x), fabsf(v[t].y)), fmaxf(fabsf(v[t].z), fabsf(v[t].w))));
        }
        s2 = row16_sum(s2);
        a = row16_max(a);
        bad |= !(a <= 3.0e38f && a >= 1.0e-30f);  // (a channel scale that does not fit this row)
        const unsigned long long bm = __ballot(bad);
        const bool marked = ((bm >> (lane & 48)) & 0xFFFFull) != 0ull || !(s2 >= 1.0e-30f && s2 <= 1.0e30f);
        const float inv = marked ? 0.0f : 127.0f / a, sc = marked ? 0.0f : a / 127.0f;
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const float e[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q8 = (int)rintf(fminf(fmaxf(e[j] * inv, -127.0f), 127.0f)) + 128;
                w |= (uint32_t)(q8 & 0xFF) << (8 * j);
            }
            if (live) *reinterpret_cast<uint32_t*>(mirror + r * DIM + 64 * t + 4 * i) = w;
        }
        if (live && i == 0) {
            xx[r] = marked ? -1.0f : s2;
            scale[r] = sc;
            cfac[r] = marked ? 0.0f : 0.53f * sc / sqrtf(s2) * 1.000001f;
        }
    }
}

// stage 1 on the byte mirror: every row's UPPER-bound key (PREF_MARK for the marked rows).  Lane i of a 16-lane group
// loads the 16 bytes at offset 256 u + 16 i of its row (dim a multiple of 256); rho_out[0] = rho for the collect pass.
template <int NCH, int RING = 4>
__global__ __launch_bounds__(256, 2) void knn_scan_coarse8_kernel(const uint8_t* __restrict__ mirror, const float* __restrict__ xx,
                                                               const float* __restrict__ scale, const float* __restrict__ cfac,
                                                               const float* __restrict__ gch, uint64_t n_rows,
                                                               const float* __restrict__ q, float e0,
                                                               uint32_t* __restrict__ all_keys, float* __restrict__ rho_out,
                                                               uint32_t* __restrict__ sample_keys = nullptr, int sample_shift = 0) {
    // sample_keys (nullable): the keys of every 2^sample_shift-th tile once more, compactly — the threshold of the collect
    // pass is then taken from this sample (its k-th smallest key is >= the k-th smallest of all rows: a valid, looser
    // threshold), and the three histogram passes read an eighth of the keys
    static_assert(NCH % 4 == 0, "rows of whole 256-byte chunks of bytes");
    constexpr int DIM = NCH * 64, U = NCH / 4;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const uint32_t wave = blockIdx.x * 4 + wib, n_waves = gridDim.x * 4;
    float qf[U][16];
    float sq, rho, qsum128;
    {
        float s2 = 0.0f, s1 = 0.0f, ss = 0.0f;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float qe = q[256 * u + 16 * i + e];
                s2 = __builtin_fmaf(qe, qe, s2);
                qf[u][e] = qe * gch[256 * u + 16 * i + e];  // q' = q g: q.x = q'.x'
                s1 += fabsf(qf[u][e]);
                ss += qf[u][e];
            }
        sq = sqrtf(row16_sum(s2));
        rho = row16_sum(s1) / sq * 1.000001f;
        qsum128 = 128.0f * row16_sum(ss);
        if (blockIdx.x == 0 && threadIdx.x == 0) rho_out[0] = rho;
    }
    const uint64_t n_tiles = (n_rows + 63) >> 6;
    auto load_row = [&](u32x4 (&x)[U], uint64_t r) {
        r = r < n_rows ? r : n_rows - 1;
        const u32x4* p = reinterpret_cast<const u32x4*>(mirror + r * DIM) + i;
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p + 16 * u);
    };
    for (uint64_t tile = wave; tile < n_tiles; tile += n_waves) {
        float mydot = 0.0f;
        const uint64_t row0 = (tile << 6) + 16 * g;
        auto reduce_row = [&](const u32x4 (&x)[U], int it) {
            float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t w[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // v_cvt_f32_ubyte0..3: one conversion per byte
                    a[0] = __builtin_fmaf(qf[u][4 * j + 0], (float)(w[j] & 0xFFu), a[0]);
                    a[1] = __builtin_fmaf(qf[u][4 * j + 1], (float)((w[j] >> 8) & 0xFFu), a[1]);
                    a[2] = __builtin_fmaf(qf[u][4 * j + 2], (float)((w[j] >> 16) & 0xFFu), a[2]);
                    a[3] = __builtin_fmaf(qf[u][4 * j + 3], (float)(w[j] >> 24), a[3]);
                }
            }
            const float d = row16_sum((a[0] + a[1]) + (a[2] + a[3]));
            if (i == it) mydot = d;
        };
        // RING - 1 rows of 768 bytes in flight per 16-lane group (A/B: MI_KNN_RING = 4 | 8 at table creation)
        u32x4 xr[RING][U];
#pragma unroll
        for (int d = 0; d < RING - 1; ++d) load_row(xr[d], row0 + d);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if (it + RING - 1 < 16) load_row(xr[(it + RING - 1) % RING], row0 + it + RING - 1);
            reduce_row(xr[it % RING], it);
        }
        const uint64_t r = (tile << 6) + lane;
        if (r < n_rows) {
            const float s = xx[r];
            const float dot = scale[r] * (mydot - qsum128);
            const float upper = (1.0f - dot / (sq * sqrtf(s))) + (cfac[r] * rho + e0);
            all_keys[r] = s < 0.0f ? PREF_MARK : dist_to_u32(upper);
        }
        if (sample_keys && (tile & ((1ull << sample_shift) - 1)) == 0) {
            uint32_t key = 0xFFFFFFFFu;   // a row beyond the table ranks last in the sample
            if (r < n_rows) {
                const float s = xx[r];
                const float dot = scale[r] * (mydot - qsum128);
                const float upper = (1.0f - dot / (sq * sqrtf(s))) + (cfac[r] * rho + e0);
                key = s < 0.0f ? PREF_MARK : dist_to_u32(upper);
            }
            sample_keys[((tile >> sample_shift) << 6) + lane] = key;
        }
    }
}

// stage 1 on the byte mirror for NQ = 4 or 8 queries in ONE pass over the mirror (the throughput form of the two-stage
// search).  Each 16-lane GROUP of a wave keeps its own query (NQ = 8: two) and all four groups read the SAME row — the
// addresses coincide, so the row crosses the memory system once — eight rows ahead.  (Handing rows fetched by different
// groups round with ds_bpermute, the first form, was bound by the CU's one LDS pipe: 9.2 ms per 8 queries over 10 M rows.)
// With the bytes shared, the pass is bound by the vector ALU: 48 byte conversions per row for all queries together plus
// 48 fused multiply-adds per row and query, issued as packed pairs (v_pk_fma_f32).  Per (row, query) the arithmetic is
// knn_scan_coarse8_kernel's, operation for operation — four fmaf chains, pairwise sum, butterfly — hence the same
// upper-bound keys and the same candidates as NQ single searches.  all_keys: [NQ][key_stride], rho_out: [NQ].
template <int NCH, int NQ>
__global__ __launch_bounds__(256, 2) void knn_scan_coarse8_batched_kernel(const uint8_t* __restrict__ mirror, const float* __restrict__ xx,
                                                                        const float* __restrict__ scale, const float* __restrict__ cfac,
                                                                        const float* __restrict__ gch, uint64_t n_rows,
                                                                        const float* __restrict__ q, float e0,
                                                                        uint32_t* __restrict__ all_keys, uint64_t key_stride,
                                                                        float* __restrict__ rho_out) {
    static_assert(NCH % 4 == 0 && (NQ == 2 || NQ == 4 || NQ == 8), "whole 256-byte chunks; 2, 4 or 8 queries");
    // GQ lane groups hold different queries (QPG each); with NQ = 2 the other two groups take the NEXT row: RPS rows per step
    constexpr int DIM = NCH * 64, U = NCH / 4, GQ = NQ < 4 ? NQ : 4, QPG = NQ / GQ, RPS = 4 / GQ, STEPS = 64 / RPS;
    constexpr int AHEAD = NQ == 8 ? 4 : 8;  // rows in flight per group: what the register file leaves beside the queries
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const int qg = g % GQ, rs = g / GQ;  // this group's query slot and its row within a step
    const uint32_t wave = blockIdx.x * 4 + wib, n_waves = gridDim.x * 4;
    f32x2 qf[QPG][U][8];  // pairs (e, e + 1) of the 16 query elements of a 16-byte piece
    float sq[QPG], rho[QPG], qsum128[QPG];
#pragma unroll
    for (int c = 0; c < QPG; ++c) {
        const float* qq = q + (size_t)(qg + GQ * c) * DIM;  // group: query qg (and qg + 4)
        float s2 = 0.0f, s1 = 0.0f, ss = 0.0f;
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float qe = qq[256 * u + 16 * i + e];
                s2 = __builtin_fmaf(qe, qe, s2);
                const float qg = qe * gch[256 * u + 16 * i + e];
                qf[c][u][e >> 1][e & 1] = qg;
                s1 += fabsf(qg);
                ss += qg;
            }
        sq[c] = sqrtf(row16_sum(s2));
        rho[c] = row16_sum(s1) / sq[c] * 1.000001f;
        qsum128[c] = 128.0f * row16_sum(ss);
        if (blockIdx.x == 0 && wib == 0 && i == 0 && rs == 0) rho_out[qg + GQ * c] = rho[c];
    }
    const uint64_t n_tiles = (n_rows + 63) >> 6;
    auto load_row = [&](u32x4 (&x)[U], uint64_t r) {
        r = r < n_rows ? r : n_rows - 1;
        const u32x4* p = reinterpret_cast<const u32x4*>(mirror + r * DIM) + i;  // the same 16 lanes' worth of addresses in all four groups
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(p + 16 * u);
    };
    for (uint64_t tile = wave; tile < n_tiles; tile += n_waves) {
        const uint64_t row0 = tile << 6;
        u32x4 xr[AHEAD][U];
#pragma unroll
        for (int d = 0; d < AHEAD; ++d) load_row(xr[d], row0 + RPS * d + rs);
        float mydot[QPG];
#pragma unroll
        for (int c = 0; c < QPG; ++c) mydot[c] = 0.0f;
#pragma unroll 1
        for (int s0 = 0; s0 < STEPS; s0 += AHEAD) {  // AHEAD steps per trip: the ring slot of a row is a compile-time index
#pragma unroll
            for (int e = 0; e < AHEAD; ++e) {
                const int it = s0 + e, row = RPS * it + rs;  // this group's row of the tile in this step
                u32x4(&x)[U] = xr[e];
                f32x2 a01[QPG], a23[QPG];
#pragma unroll
                for (int c = 0; c < QPG; ++c) { a01[c] = (f32x2){0.0f, 0.0f}; a23[c] = (f32x2){0.0f, 0.0f}; }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t w[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {  // v_cvt_f32_ubyte0..3 once per row, shared by the group's queries
                        const f32x2 b01 = (f32x2){(float)(w[j] & 0xFFu), (float)((w[j] >> 8) & 0xFFu)};
                        const f32x2 b23 = (f32x2){(float)((w[j] >> 16) & 0xFFu), (float)(w[j] >> 24)};
#pragma unroll
                        for (int c = 0; c < QPG; ++c) {
                            a01[c] = __builtin_elementwise_fma(qf[c][u][2 * j], b01, a01[c]);      // chains a[0], a[1]
                            a23[c] = __builtin_elementwise_fma(qf[c][u][2 * j + 1], b23, a23[c]);  // chains a[2], a[3]
                        }
                    }
                }
                if (it + AHEAD < STEPS) load_row(x, row0 + RPS * (it + AHEAD) + rs);  // the slot just used takes the row AHEAD steps on
#pragma unroll
                for (int c = 0; c < QPG; ++c) {
                    const float d = row16_sum((a01[c].x + a01[c].y) + (a23[c].x + a23[c].y));
                    if (i == (row & 15)) mydot[c] = d;
                }
            }
            if (((RPS * (s0 + AHEAD)) & 15) == 0) {  // sixteen rows done: lane i (i % RPS == rs) holds row 16 s + i of its queries
                const uint64_t r = row0 + (uint64_t)(RPS * (s0 + AHEAD) - 16) + i;
                if (r < n_rows && (i % RPS) == rs) {
                    const float s = xx[r], sc = scale[r], cf = cfac[r];
#pragma unroll
                    for (int c = 0; c < QPG; ++c) {
                        const float dot = sc * (mydot[c] - qsum128[c]);
                        const float upper = (1.0f - dot / (sq[c] * sqrtf(s))) + (cf * rho[c] + e0);
                        all_keys[(size_t)(qg + GQ * c) * key_stride + r] = s < 0.0f ? PREF_MARK : dist_to_u32(upper);
                    }
                }
            }
        }
    }
}

// ---- stage 1 on the byte mirror for a group of queries on the MATRIX pipe (round 4) ---------------------------------
// The VALU form above converts every byte to fp32 and pays 48 conversions + 24 packed FMAs per row and query: 2.3 TB/s of
// mirror, 29 % of the HBM peak, "the one kNN kernel far from its roofline" (VERDICT r3).  Here the bytes stay bytes:
//   * the query is cut into three signed 7-bit digits: q'_j = q_j g_j ~ S (16384 a_j + 128 b_j + c_j), Q_j = round(q'_j / S),
//     S = max_j |q'_j| / 2^20, a, b, c in [-64, 64] (knn_query_digits_kernel);
//   * v_mfma_i32_16x16x64_i8 multiplies 16 rows (the mirror's bytes with the top bit flipped: u - 128 as a signed byte) by 16
//     digit columns (5 queries x 3 digits per column block): EXACT integer dot products, 24 MFMAs per 16 rows for 8 queries;
//   * sum_j Q_j (u_j - 128) = 16384 A + 128 B + C (three exact floats, two roundings), dot = scale_r S (that sum), and the key is
//     the same upper bound as before: (1 - dot / (|q| |x|)) + (c_r rho + e0).
// The bound now also has to cover the query's own quantisation: |q^ . x^ - q'. x'| <= (S/2) |x^|_1 + (s/2) |q'|_1 with
// |x^|_1 <= 127 dim s, i.e. rho grows by rho_q = 1.25 x 127 dim S / |q| = 0.116 max|q'| / |q| at dim 768 (a percent of rho); the
// integer sums have no rounding at all, so the 0.53 (for 0.5) in c_r now only pays for the mirror's own rounding.  The
// collect pass reads the same rho (rho_out), so keys and band stay consistent.  The keys differ in their last bits from the
// single-query kernel's (other arithmetic, both inside the bound): the CANDIDATES may differ, the answers cannot.
// Data path: a wave owns tiles of 16 rows (16 dim bytes, contiguous); each tile arrives by LDS-DMA (1 KiB per instruction,
// fully coalesced) into a private ring of three LDS images, two tiles ahead, with the per-row scalars behind it; the image
// is linear with 16-byte chunk c of row r at slot (c & ~15) | ((c & 15) ^ r) (the swizzle is applied to the SOURCE address),
// so the 16 lanes of a ds_read_b128 group hit 16 different bank quads.  Lane (r, g) of step t reads chunk 12 g + t: a
// permutation of k that the digit fragments follow.  No barrier anywhere: nothing is shared between waves.
constexpr int C8M_NB = 3;          // ring of LDS images per wave
typedef int v4i32 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int c8m_tile_bytes(int dim) { return 16 * dim + 768; }   // rows + 3 x 256 B of per-row scalars
__host__ __device__ constexpr int c8m_lds_bytes(int dim) { return 4 * C8M_NB * c8m_tile_bytes(dim); }

// per query: digits[q][3][dim] (signed bytes, most significant digit first), qs[q] = {S, |q|_2, rho + rho_q, -}
__global__ __launch_bounds__(256) void knn_query_digits_kernel(const float* __restrict__ q, const float* __restrict__ gch, int dim,
                                                               int8_t* __restrict__ digits, float* __restrict__ qs,
                                                               float* __restrict__ rho_out) {
    __shared__ float red[3][256];
    const float* qq = q + (size_t)blockIdx.x * dim;
    float s2 = 0.0f, s1 = 0.0f, mx = 0.0f;
    for (int j = threadIdx.x; j < dim; j += 256) {
        const float qe = qq[j], qg = qe * gch[j];
        s2 = __builtin_fmaf(qe, qe, s2);
        s1 += fabsf(qg);
        mx = fmaxf(mx, fabsf(qg));          // (a NaN element leaves mx alone and poisons s2: the key becomes NaN, as it must)
    }
    red[0][threadIdx.x] = s2; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = mx;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d) {
            red[0][threadIdx.x] += red[0][threadIdx.x + d];
            red[1][threadIdx.x] += red[1][threadIdx.x + d];
            red[2][threadIdx.x] = fmaxf(red[2][threadIdx.x], red[2][threadIdx.x + d]);
        }
        __syncthreads();
    }
    const float sq = sqrtf(red[0][0]), S = red[2][0] * 0x1p-20f;
    const float inv = S > 0.0f && S <= 3.0e38f ? 1.0f / S : 0.0f;
    for (int j = threadIdx.x; j < dim; j += 256) {
        float v = qq[j] * gch[j] * inv;
        v = v == v ? fminf(fmaxf(v, -1048576.0f), 1048576.0f) : 0.0f;
        const int Q = (int)rintf(v);
        const int c = ((Q + 64) & 127) - 64, Q1 = (Q - c) >> 7;
        const int b = ((Q1 + 64) & 127) - 64, a = (Q1 - b) >> 7;           // |a| <= 64
        int8_t* d = digits + (size_t)blockIdx.x * 3 * dim + j;
        d[0] = (int8_t)a; d[dim] = (int8_t)b; d[2 * dim] = (int8_t)c;
    }
    if (threadIdx.x == 0) {
        // rho_q with 1.25: |Q_j - q'_j / S| <= 0.5 + 2^20 2^-23 (the two roundings of q' / S), and c_r carries 0.53 where 0.5 x 1.25 is needed
        const float rho = (red[1][0] / sq + 1.25f * 127.0f * (float)dim * S / sq) * 1.000001f;
        qs[4 * blockIdx.x + 0] = S; qs[4 * blockIdx.x + 1] = sq; qs[4 * blockIdx.x + 2] = rho; qs[4 * blockIdx.x + 3] = 0.0f;
        rho_out[blockIdx.x] = rho;
    }
}

// NBLK column blocks of 16 digit columns with QPB queries (3 QPB <= 16 columns) each: <1,5> up to 5 queries, <2,5> up to 10,
// <3,5> up to 15, <4,4> 16; all_keys: [nq][key_stride]
template <int NCH, int NBLK, int QPB>
__global__ __launch_bounds__(256, 1) void knn_scan_coarse8_mfma_kernel(const uint8_t* __restrict__ mirror, const float* __restrict__ xx,
                                                                     const float* __restrict__ scale, const float* __restrict__ cfac,
                                                                     uint64_t n_rows, const int8_t* __restrict__ digits,
                                                                     const float* __restrict__ qs, int nq, float e0,
                                                                     uint32_t* __restrict__ all_keys, uint64_t key_stride,
                                                                     uint32_t* __restrict__ sample_keys = nullptr, int sample_shift = 0,
                                                                     uint64_t sample_stride = 0) {
    static_assert(NCH % 4 == 0 && 3 * QPB <= 16, "rows of whole 256-byte chunks; a block's digit columns fit 16");
    constexpr int DIM = NCH * 64, STEPS = DIM / 64, NDMA = DIM / 64, TILE = c8m_tile_bytes(DIM), ROWS = 16 * DIM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    unsigned char* ring = smem + wib * (C8M_NB * TILE);
    const uint32_t wave = blockIdx.x * 4 + wib, n_waves = gridDim.x * 4;
    const uint64_t n_tiles = (n_rows + 15) >> 4;

    // the digit fragments of this lane's column: column n of block b is digit d = n % 3 of query QPB b + n / 3
    v4i32 bf[NBLK][STEPS];
    float Sq[NBLK], rho[NBLK];   // Sq = S / |q|: the digit scale over the query's norm (one multiplication per key instead of a division)
    int qcol[NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int qi = QPB * b + n / 3, d = n % 3;
        const bool live = n < 3 * QPB && qi < nq;
        qcol[b] = live && d == 0 ? qi : -1;
        Sq[b] = live ? qs[4 * qi + 0] / qs[4 * qi + 1] : 0.0f; rho[b] = live ? qs[4 * qi + 2] : 0.0f;
#pragma unroll
        for (int t = 0; t < STEPS; ++t)
            bf[b][t] = live ? *reinterpret_cast<const v4i32*>(digits + ((size_t)qi * 3 + d) * DIM + (STEPS * g + t) * 16) : (v4i32){0, 0, 0, 0};
    }
    // DMA instruction i fills LDS slots 64 i + lane: slot S -> row S / (DIM/16), position S % (DIM/16), source chunk un-swizzled
    uint32_t src_off[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int slot = 64 * i + lane, r = slot / (DIM / 16), p = slot % (DIM / 16);
        const int c = (p & ~15) | ((p & 15) ^ r);
        src_off[i] = (uint32_t)(r * DIM + c * 16);
    }
    auto issue = [&](uint64_t tile, int buf) {
        unsigned char* dst = ring + buf * TILE;
        const uint64_t row0 = tile << 4;
        const uint64_t left = (n_rows - row0) * DIM;   // bytes of the mirror behind row0: the range check zero-fills a ragged last tile
        const rsrc_c8 mr = c8_rsrc(mirror + row0 * DIM, (uint32_t)(left < (uint64_t)ROWS ? left : (uint64_t)ROWS));
#pragma unroll
        for (int i = 0; i < NDMA; ++i) c8_dma16_nt(mr, src_off[i], dst + 1024 * i);
        const uint32_t nrow = (uint32_t)(n_rows - row0 < 16 ? n_rows - row0 : 16);
        c8_dma4(c8_rsrc(xx + row0, nrow * 4), (uint32_t)lane * 4u, dst + ROWS);
        c8_dma4(c8_rsrc(scale + row0, nrow * 4), (uint32_t)lane * 4u, dst + ROWS + 256);
        c8_dma4(c8_rsrc(cfac + row0, nrow * 4), (uint32_t)lane * 4u, dst + ROWS + 512);
    };
    constexpr int PER_TILE = NDMA + 3;      // vector-memory instructions per tile issue
    uint64_t tile = wave;
    if (tile >= n_tiles) return;
    issue(tile, 0);
    if (tile + n_waves < n_tiles) issue(tile + n_waves, 1);
    int buf = 0;
    for (; tile < n_tiles; tile += n_waves) {
        const uint64_t t2 = tile + 2 * (uint64_t)n_waves;
        const bool more2 = t2 < n_tiles, more1 = tile + n_waves < n_tiles;
        if (more2) issue(t2, buf == 0 ? 2 : buf - 1);   // (buf + 2) % 3: its last reader was the tile before this one
        // behind this tile's DMA: [stores of the tile two back] [DMA next] [stores of the previous tile] [DMA after next]
        // (the key stores are counted as if they were not there: a block without a live query issues none, and a count
        // that is too small only waits for a few older instructions more)
        if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_TILE) : "memory");
        else if (more1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned char* img = ring + buf * TILE;
        v4i32 acc[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) acc[b] = (v4i32){0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            const int c = STEPS * g + t;
            v4i32 a = *reinterpret_cast<const v4i32*>(img + n * DIM + (((c & ~15) | ((c & 15) ^ n)) << 4));
            a = a ^ (v4i32){(int)0x80808080, (int)0x80808080, (int)0x80808080, (int)0x80808080};   // u -> u - 128 as a signed byte
#pragma unroll
            for (int b = 0; b < NBLK; ++b) acc[b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, bf[b][t], acc[b], 0, 0, 0);
        }
        // lane (n, g) holds rows 4 g .. 4 g + 3 of column n; the three digits of a query sit in lanes n, n + 1, n + 2 of the row
        const f32x4 xs = *reinterpret_cast<const f32x4*>(img + ROWS + 16 * g);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(img + ROWS + 256 + 16 * g);
        const f32x4 cf = *reinterpret_cast<const f32x4*>(img + ROWS + 512 + 16 * g);
        const uint64_t row0 = (tile << 4) + 4 * g;
        // per row, once for all queries: scale_r / |x_r| (a reciprocal square root and a product instead of a division and a
        // square root per key: a few ulp, inside e0 like every other rounding of the coarse distance)
        float rs[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rs[j] = sc[j] * __builtin_amdgcn_rsqf(xs[j]);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            u32x4 key;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int A = acc[b][j];
                const int B = __builtin_amdgcn_update_dpp(0, A, 0x101, 0xF, 0xF, true);   // row_shl:1: lane n takes lane n + 1
                const int C = __builtin_amdgcn_update_dpp(0, A, 0x102, 0xF, 0xF, true);   // row_shl:2
                const float D = ((float)A * 16384.0f + (float)B * 128.0f) + (float)C;
                const float upper = (1.0f - (Sq[b] * D) * rs[j]) + (cf[j] * rho[b] + e0);
                key[j] = xs[j] < 0.0f ? PREF_MARK : dist_to_u32(upper);
            }
            // rows beyond n_rows (a ragged last tile) write into the slack of the key array: cap is a multiple of 64 rows
            if (qcol[b] >= 0) {
                *reinterpret_cast<u32x4*>(all_keys + (size_t)qcol[b] * key_stride + row0) = key;
                // every 2^sample_shift-th tile once more, compactly (knn_scan_coarse8_kernel: the sample the threshold is taken from);
                // a row beyond the table must rank LAST there (its key above is arithmetic on zero-filled scalars)
                if (sample_keys && (tile & ((1ull << sample_shift) - 1)) == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) key[j] = row0 + j < n_rows ? key[j] : 0xFFFFFFFFu;
                    *reinterpret_cast<u32x4*>(sample_keys + (size_t)qcol[b] * sample_stride + ((tile >> sample_shift) << 4) + 4 * g) = key;
                }
            } else asm volatile("" ::"v"(key));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this image's reads have retired before it is refilled two tiles on
        buf = buf == 2 ? 0 : buf + 1;
    }
}

// the candidates of the 8-bit stage: rows whose LOWER bound U_r - 2 eps_r is not beyond the k-th smallest upper bound
__global__ __launch_bounds__(256) void knn_prefilter_collect8_kernel(const uint32_t* __restrict__ keys, const float* __restrict__ cfac,
                                                                     uint64_t n_rows, uint32_t k, const uint32_t* __restrict__ hist,
                                                                     SelState* __restrict__ states, const float* __restrict__ rho_ptr,
                                                                     float e0, uint32_t cap, uint32_t* __restrict__ cand_rows,
                                                                     uint32_t* __restrict__ count, QGroup qg = QGroup{}) {
    keys += (size_t)blockIdx.y * qg.keys;
    hist += (size_t)blockIdx.y * qg.sel;
    states = reinterpret_cast<SelState*>(reinterpret_cast<uint32_t*>(states) + (size_t)blockIdx.y * qg.sel);
    rho_ptr += (size_t)blockIdx.y * qg.rho;
    cand_rows += (size_t)blockIdx.y * qg.rows;
    count += (size_t)blockIdx.y * qg.flags;
    const SelState st = sel_advance(hist, states, 3, k, n_rows);
    float tau = __uint_as_float(0x7F800000u);  // fewer rows than k, or a NaN at rank k: everything
    if (n_rows >= k && st.fixed >= 1) {
        const uint64_t T = st.prefix | ((1ull << sel_shift(st.fixed - 1)) - 1ull);
        const uint32_t T32 = (uint32_t)(T >> 32);
        const float t = u32_to_dist(T32);
        if (T32 < PREF_MARK && t == t) tau = t + 2e-6f;
    }
    const float rho = rho_ptr[0];
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t r0 = (uint64_t)blockIdx.x * 256 + (threadIdx.x & ~63u); r0 < n_rows; r0 += stride) {
        const uint64_t r = r0 + (threadIdx.x & 63);
        bool take = false;
        if (r < n_rows) {
            const uint32_t key = keys[r];
            take = key == PREF_MARK || key == 0xFFFFFFFFu || !(u32_to_dist(key) - 2.0f * (cfac[r] * rho + e0) > tau);
        }
        const unsigned long long m = __ballot(take);
        if (m == 0ull) continue;
        uint32_t base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(count, (uint32_t)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (take) {
            const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            if (at < cap) cand_rows[at] = (uint32_t)r;
        }
    }
}

// stage 2: the candidates' exact distance keys — RowAcc, row16_sum and the distance expression of knn_scan_kernel, one
// 16-lane group per row — into key32_out[c] beside cand_rows[c]; the k smallest (key, row) pairs are then found by the
// radix select over those two arrays.  flags[0] = candidate count (in), flags[1] = fallback, flags[2] = go (out): more
// candidates than cap => nothing is computed, fallback = 1, go = 0.
template <int NCH>
__global__ __launch_bounds__(256) void knn_rescore_kernel(const float* __restrict__ table, const float* __restrict__ q,
                                                          const uint32_t* __restrict__ cand_rows, uint32_t* __restrict__ flags,
                                                          uint32_t cap, uint32_t* __restrict__ key32_out, QGroup qg = QGroup{}) {
    constexpr int DIM = NCH * 64;
    q += (size_t)blockIdx.y * qg.q;
    cand_rows += (size_t)blockIdx.y * qg.rows;
    key32_out += (size_t)blockIdx.y * qg.rows;
    flags += (size_t)blockIdx.y * qg.flags;
    const uint32_t C = flags[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        flags[1] = C > cap ? 1u : 0u;
        flags[2] = C > cap ? 0u : 1u;
    }
    if (C > cap) return;
    const int lane = threadIdx.x & 63, i = lane & 15;
    f32x4 qf[NCH];
#pragma unroll
    for (int t = 0; t < NCH; ++t) qf[t] = *reinterpret_cast<const f32x4*>(q + 64 * t + 4 * i);
    float sq;
    {
        RowAcc<NCH> a; a.zero();
#pragma unroll
        for (int t = 0; t < NCH; ++t) a.step(qf[t], qf[t]);
        sq = sqrtf(a.sumsq());
    }
    const uint32_t group = (blockIdx.x * 256 + threadIdx.x) >> 4, n_groups = (gridDim.x * 256) >> 4;
    for (uint32_t c0 = group; c0 < ((C + n_groups - 1) / n_groups) * n_groups; c0 += n_groups) {
        const bool live = c0 < C;
        const uint32_t row = cand_rows[live ? c0 : 0];
        const f32x4* p = reinterpret_cast<const f32x4*>(table + (uint64_t)row * DIM) + i;
        RowAcc<NCH> a; a.zero();
#pragma unroll
        for (int t = 0; t < NCH; ++t) a.step(qf[t], __builtin_nontemporal_load(p + 16 * t));
        const float d = a.dot(), s = a.sumsq();
        if (live && i == 0) key32_out[c0] = dist_to_u32(1.0f - d / (sq * sqrtf(s)));
    }
}

// Q queries in one pass over the table (throughput variant).  Same per-row
// arithmetic per query, so results equal Q single-query scans.  q: [NQ][dim].
template <int NCH, int NQ>
__global__ __launch_bounds__(256) void knn_scan_batched_kernel(const float* __restrict__ table, uint64_t n_rows,
                                                               const float* __restrict__ q, uint32_t k,
                                                               uint64_t* __restrict__ cand /*[NQ][waves][k]*/) {
    constexpr int DIM = NCH * 64;
    __shared__ __attribute__((aligned(16))) float qs[NQ * DIM];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int i = lane & 15, g = lane >> 4;
    const uint32_t wave = blockIdx.x * 4 + wib, n_waves = gridDim.x * 4;
    for (int j = threadIdx.x; j < NQ * DIM; j += 256) qs[j] = q[j];
    __syncthreads();

    float sq[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        RowAcc<NCH> a; a.zero();
#pragma unroll
        for (int t = 0; t < NCH; ++t) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&qs[u * DIM + 64 * t + 4 * i]);
            a.step(v, v);
        }
        sq[u] = sqrtf(a.sumsq());
    }
    WaveTopReg top[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) top[u].init(nullptr, k, lane);

    const uint64_t n_tiles = (n_rows + 63) >> 6;
    for (uint64_t tile = wave; tile < n_tiles; tile += n_waves) {
        float mydot[NQ], myxx = 1.0f;
#pragma unroll
        for (int u = 0; u < NQ; ++u) mydot[u] = 0.0f;
        const uint64_t row0 = (tile << 6) + 16 * g;
        for (int it = 0; it < 16; ++it) {
            uint64_t r = row0 + it;
            r = r < n_rows ? r : n_rows - 1;
            const f32x4* p = reinterpret_cast<const f32x4*>(table + r * DIM) + i;
            f32x4 x[NCH];
#pragma unroll
            for (int t = 0; t < NCH; ++t) x[t] = __builtin_nontemporal_load(p + 16 * t);
            float s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int t = 0; t < NCH; ++t) {
                s0 = __builtin_fmaf(x[t].x, x[t].x, s0); s1 = __builtin_fmaf(x[t].y, x[t].y, s1);
                s2 = __builtin_fmaf(x[t].z, x[t].z, s2); s3 = __builtin_fmaf(x[t].w, x[t].w, s3);
            }
            const float s = row16_sum((s0 + s1) + (s2 + s3));
            if (i == it) myxx = s;
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                float d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
                for (int t = 0; t < NCH; ++t) {
                    // volatile: keep the query fragment an LDS read per use — hoisted out of the row
                    // loop the NQ x NCH fragments would need NQ*48 registers
                    const f32x4 v = *reinterpret_cast<const volatile f32x4*>(&qs[u * DIM + 64 * t + 4 * i]);
                    d0 = __builtin_fmaf(v.x, x[t].x, d0); d1 = __builtin_fmaf(v.y, x[t].y, d1);
                    d2 = __builtin_fmaf(v.z, x[t].z, d2); d3 = __builtin_fmaf(v.w, x[t].w, d3);
                }
                const float d = row16_sum((d0 + d1) + (d2 + d3));
                if (i == it) mydot[u] = d;
            }
        }
        const uint64_t r = (tile << 6) + lane;
        const float sx = sqrtf(myxx);
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const float dist = 1.0f - mydot[u] / (sq[u] * sx);
            top[u].offer(r < n_rows ? make_key(dist, (uint32_t)r) : KEY_MAX);
        }
    }
#pragma unroll
    for (int u = 0; u < NQ; ++u) top[u].store(cand + ((size_t)u * n_waves + wave) * k);
}

// ---- tree reduction of candidate lists ----------------------------------------------
// block b merges lists [b*lpb, min(n_lists,(b+1)*lpb)) of k keys each into one list.
// `in`/`out` are indexed per query by blockIdx.y (strides in keys).
template <class Top>
__global__ __launch_bounds__(256) void knn_merge_kernel(const uint64_t* __restrict__ in, uint32_t n_lists,
                                                        uint32_t k, uint32_t lpb, uint64_t* __restrict__ out,
                                                        size_t in_stride, size_t out_stride,
                                                        const uint32_t* __restrict__ run_if = nullptr, uint32_t run_stride = 0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint64_t wave_best[4][64];  // register form hands its lists over through here
    if (run_if) run_if += (size_t)blockIdx.y * run_stride;
    if (run_if && *run_if == 0u) return;
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    in += blockIdx.y * in_stride;
    out += blockIdx.y * out_stride;
    const uint32_t l0 = blockIdx.x * lpb;
    const uint32_t l1 = min(n_lists, l0 + lpb);
    const uint64_t* src = in + (size_t)l0 * k;
    const uint32_t n_keys = (l1 - l0) * k;

    Top top;
    uint64_t* my_lds = reinterpret_cast<uint64_t*>(smem) + (size_t)wib * Top::LDS_KEYS;
    top.init(my_lds, k, lane);
    for (uint32_t j0 = 0; j0 < n_keys; j0 += 256) {
        const uint32_t j = j0 + threadIdx.x;
        top.offer(j < n_keys ? src[j] : KEY_MAX);
    }
    top.finish();
    if (Top::LDS_KEYS == 0) wave_best[wib][lane] = top.lane_key(lane);
    __syncthreads();
    if (wib == 0) {
        for (int w = 1; w < 4; ++w) {
            if (Top::LDS_KEYS == 0) {
                top.offer(wave_best[w][lane]);
            } else {
                const uint64_t* other = reinterpret_cast<uint64_t*>(smem) + (size_t)w * Top::LDS_KEYS;
                for (uint32_t j = 0; j < (uint32_t)Top::KP; j += 64) {
                    const uint32_t idx = j + lane;
                    top.offer(idx < k ? other[idx] : KEY_MAX);
                }
            }
        }
        top.finish();
        top.store(out + (size_t)blockIdx.x * k);
    }
}

// ---- tree reduction for k > 64: the whole 256-thread block keeps ONE list ------------------
// block b merges lists [b*lpb, min(n_lists,(b+1)*lpb)) of k keys each.  LDS: buf[0,KP) the best
// keys so far (ascending after a merge), buf[KP,2KP) accepted-but-unsorted keys, a shared count
// and threshold.  A key is accepted when it beats the current k-th best; when the pending half
// could overflow, all 256 threads run one bitonic sort of the 2*KP keys (about 5 us for 2048,
// against 25 us for the same sort by a single wave in WaveTopLds).
template <int KP>
__global__ __launch_bounds__(256) void knn_merge_block_kernel(const uint64_t* __restrict__ in, uint32_t n_lists,
                                                              uint32_t k, uint32_t lpb, uint64_t* __restrict__ out) {
    __shared__ uint64_t buf[2 * KP];
    __shared__ uint64_t s_thr;
    __shared__ uint32_t s_cnt;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t l0 = blockIdx.x * lpb;
    const uint32_t l1 = l0 + lpb < n_lists ? l0 + lpb : n_lists;
    const uint64_t* src = in + (size_t)l0 * k;
    const uint32_t n_keys = (l1 - l0) * k;
    for (int j = tid; j < 2 * KP; j += 256) buf[j] = KEY_MAX;
    // A tight first threshold from the sorted inputs: with m = ceil(k / #lists), the largest of the
    // lists' m-th keys bounds the union's k-th key from above (the union holds #lists*m >= k keys
    // that are <= it), so everything above it is rejected before the first sort.
    {
        const uint32_t nl = l1 - l0, m = (k + nl - 1) / nl;
        uint64_t v = 0;
        for (uint32_t l = tid; l < nl; l += 256) { const uint64_t x = src[(size_t)l * k + (m - 1)]; v = x > v ? x : v; }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const uint64_t o = shfl_xor64(v, off); v = o > v ? o : v; }
        if (lane == 0) buf[tid >> 6] = v;  // four wave maxima, read back below before buf is used
        __syncthreads();
        if (tid == 0) {
            uint64_t t0 = buf[0];
            for (int w = 1; w < 4; ++w) t0 = buf[w] > t0 ? buf[w] : t0;
            s_thr = t0 == KEY_MAX ? KEY_MAX : t0 + 1;
            s_cnt = 0;
        }
        __syncthreads();
        if (tid < 4) buf[tid] = KEY_MAX;
    }
    __syncthreads();
    auto block_sort = [&]() {  // ascending bitonic sort of buf[0, 2KP), then reset the pending half
        for (int kk = 2; kk <= 2 * KP; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int p = tid; p < KP; p += 256) {
                    const int a = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                    const int b = a | j;
                    const uint64_t x = buf[a], y = buf[b];
                    const bool up = (a & kk) == 0;
                    if ((x > y) == up) { buf[a] = y; buf[b] = x; }
                }
                __syncthreads();
            }
        if (tid == 0) { s_thr = buf[k - 1]; s_cnt = 0; }
        for (int j = KP + tid; j < 2 * KP; j += 256) buf[j] = KEY_MAX;
        __syncthreads();
    };
    for (uint32_t j0 = 0; j0 < n_keys; j0 += 256) {
        const uint32_t j = j0 + tid;
        const uint64_t key = j < n_keys ? src[j] : KEY_MAX;
        const bool pass = key < s_thr;
        const unsigned long long mask = __ballot(pass);
        uint32_t base = 0;
        if (lane == 0 && mask) base = atomicAdd(&s_cnt, (uint32_t)__popcll(mask));
        base = __shfl(base, 0, 64);
        if (pass) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
            buf[KP + base + rank] = key;
        }
        __syncthreads();
        if (s_cnt + 256 > (uint32_t)KP) block_sort();  // uniform: s_cnt is stable between barriers
    }
    if (s_cnt) block_sort();
    for (uint32_t j = tid; j < k; j += 256) out[(size_t)blockIdx.x * k + j] = buf[j];
}

// Row ids of a shard.  Plain: id = base + local ordinal.  Block-cyclic (a shard of mi_knn_sharded: global row r lives in
// block r / B, blocks are dealt round-robin to the n shards): id = base + ((local / B) * n + rank) * B + local % B —
// monotone in the local ordinal, so "(distance asc, local asc)" inside a shard IS "(distance asc, id asc)".
struct IdMap { uint64_t base; uint32_t block, n, rank; };
__host__ __device__ inline uint64_t id_of_local(const IdMap& m, uint64_t local) {
    if (m.n <= 1 || m.block == 0) return m.base + local;
    return m.base + ((local / m.block) * m.n + m.rank) * m.block + local % m.block;
}

// keys (ascending, KEY_MAX = none) -> (id, distance); one thread per result slot.
// prefilter_keys / fallback (nullable): while *fallback == 0 the keys come from prefilter_keys instead
__global__ void knn_finalize_kernel(const uint64_t* __restrict__ keys, uint32_t n, IdMap map,
                                    uint64_t* __restrict__ idx, float* __restrict__ dist,
                                    size_t key_stride, size_t out_stride,
                                    const uint64_t* __restrict__ prefilter_keys = nullptr,
                                    const uint32_t* __restrict__ fallback = nullptr, uint32_t fallback_stride = 0) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (fallback) fallback += (size_t)blockIdx.y * fallback_stride;
    if (fallback && *fallback == 0u) keys = prefilter_keys;
    const uint64_t key = keys ? keys[blockIdx.y * key_stride + j] : KEY_MAX;
    uint64_t* oi = idx + blockIdx.y * out_stride;
    float* od = dist + blockIdx.y * out_stride;
    if (key == KEY_MAX) {
        oi[j] = MI_KNN_NO_ID;
        od[j] = __uint_as_float(0x7F800000u);
    } else {
        oi[j] = id_of_local(map, (uint32_t)key);
        od[j] = u32_to_dist((uint32_t)(key >> 32));
    }
}

// ---- merge of per-shard result lists on the device (mi_knn_merge_device) ---------------
// in: [lists][nq][k] (id, distance) entries, every list ascending by (dist_to_u32(distance), id) with its
// MI_KNN_NO_ID padding at the tail — what each rank holds after the all-gather.  One thread per entry: its place in
// the merged order is its own position plus, per other list, the number of entries in front of it (a binary search;
// equal entries of two lists keep list order), so nothing is sorted and any lists * k fits.  Same result as
// mi::merge_lists (core.hip).
__device__ __forceinline__ bool merge_less(uint32_t ka, uint64_t ia, uint32_t kb, uint64_t ib) {
    return ka < kb || (ka == kb && ia < ib);
}
__global__ __launch_bounds__(256) void knn_merge_lists_kernel(const uint64_t* __restrict__ idx_in, const float* __restrict__ dist_in,
                                                              uint32_t lists, uint32_t k, size_t idx_stride, size_t dist_stride,
                                                              uint64_t* __restrict__ idx, float* __restrict__ dist) {
    const uint32_t u = blockIdx.y;  // query
    const uint64_t* qi = idx_in + (size_t)u * k;
    const float* qd = dist_in + (size_t)u * k;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < k) {  // slots behind the last real entry: "none"
        uint32_t valid = 0;
        for (uint32_t l = 0; l < lists; ++l) {
            const uint64_t* li = qi + l * idx_stride;
            uint32_t lo = 0, hi = k;  // first NO_ID of the list
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (li[mid] != MI_KNN_NO_ID) lo = mid + 1; else hi = mid;
            }
            valid += lo;
        }
        if (e >= valid) {
            idx[(size_t)u * k + e] = MI_KNN_NO_ID;
            dist[(size_t)u * k + e] = __uint_as_float(0x7F800000u);
        }
    }
    if (e >= lists * k) return;
    const uint32_t l = e / k, p = e % k;
    const uint64_t id = qi[l * idx_stride + p];
    if (id == MI_KNN_NO_ID) return;
    const float d = qd[l * dist_stride + p];
    const uint32_t key = dist_to_u32(d);
    uint32_t rank = p;
    for (uint32_t o = 0; o < lists && rank < k; ++o) {
        if (o == l) continue;
        const uint64_t* li = qi + o * idx_stride;
        const float* ld = qd + o * dist_stride;
        uint32_t lo = 0, hi = k;  // entries of list o in front of this one
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const uint64_t mi_ = li[mid];
            const uint32_t mk = dist_to_u32(ld[mid]);
            const bool front = mi_ != MI_KNN_NO_ID && (o < l ? !merge_less(key, id, mk, mi_) : merge_less(mk, mi_, key, id));
            if (front) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    if (rank < k) {
        idx[(size_t)u * k + rank] = id;
        dist[(size_t)u * k + rank] = d;
    }
}

// ---- seeded corpus generator (image_search_amd/synth.py gen_f32, bit for bit) ---------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float gen1(uint64_t key, uint64_t i, float scale) {
    const uint64_t h = mix64(i + key);
    const int s = (int)((h & 0xffff) + ((h >> 16) & 0xffff) + ((h >> 32) & 0xffff) + (h >> 48));
    return (float)(s - 131070) * scale;
}
// out[j] = value(counter first+j), j < n; n % 4 == 0, out 16-byte aligned
__global__ void gen_f32_kernel(float* __restrict__ out, uint64_t key, uint64_t first, uint64_t n, float scale) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 4;
    for (uint64_t j = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; j < n; j += stride) {
        f32x4 v;
        v.x = gen1(key, first + j, scale); v.y = gen1(key, first + j + 1, scale);
        v.z = gen1(key, first + j + 2, scale); v.w = gen1(key, first + j + 3, scale);
        *reinterpret_cast<f32x4*>(out + j) = v;
    }
}

}  // namespace mi
