// attn64_kernels.h — EXPERIMENT, not part of the library (built only by tools/probe/attn_bench.hip): bf16 attention of
// the vision tower at ViT-L/14's shape (S = 257) with TWO query tiles interleaved in one wave's program order.
// Correct (it passed tests/test_vit_gpu.py when wired in as attention version 3) and slower than attn32: numbers at
// the end of this comment and in DESIGN.md 5.3.
//
// Same arithmetic, LDS images, LDS-DMA double buffering, split last query and stores as attn32_bf16_kernel
// (attn32_kernels.h); what changes is who overlaps with whom.  There, two waves share a SIMD and each runs the
// dependent chain  scores (MFMA) -> exp2 / pack (VALU) -> P V (MFMA)  of ONE 32-query tile: the hardware interleaves
// the two chains as the arbitration falls, and the stamps say a whole tile takes 6.7 k cycles for the older wave of
// a SIMD and 10.1 k for the younger against 5.8 k of matrix-pipe time for both.  Here a workgroup is 4 waves, a wave
// owns TWO query tiles (a, b: 64 queries) and the whole 512-register file, and the two chains are interleaved in
// program order, half a key step apart:
//
//     half-step H1(t):  MFMA  P V of a(t), scores of a(t+1)      VALU  exp2 / pack of b(t)
//     half-step H2(t):  MFMA  P V of b(t), scores of b(t+1)      VALU  exp2 / pack of a(t+1)
//
// 10 MFMAs (320 matrix-pipe cycles) against 16 exp2 + 8 packs (~165 issue cycles) per half-step: the VALU work fits
// in the MFMA gaps (MI355X_MICROARCH.md, "single-issue instructions hidden per MFMA gap"), every K / V fragment read
// from LDS serves 64 queries instead of 32, and no arbitration decides anything.  The next pair's K/V arrive by
// LDS-DMA one piece per half-step; the next pair's query fragments by one load per half-step.
//
// Measured (b = 256, us per layer; attn32 on the same boxes 146-150):
//   4 waves, one per SIMD, 512 registers, each wave doing everything: 170-176.  Taken apart: sweep alone 75, + ctx stores
//     15, + split query 25 (a dependent chain nothing overlaps), + LDS-DMA and query loads 55 (a piece costs its issuing
//     wave 60-185 cycles and there are 26 vector-memory instructions per wave and pair against 144 MFMAs).
//   4 compute + 4 service waves (this file): 158-188.  256 registers per wave make hipcc spill around the sweep; and the
//     sweep itself runs 9.3 k cycles per pair with no memory traffic at all, 12.3 k with the query prefetch, 13.0 k with
//     the DMA (issued by the OTHER wave of the SIMD), 16.3 k with both, against 4.6 k of matrix-pipe time: at S = 257
//     there is one vector-memory instruction per 5 MFMAs (a long-sequence flash kernel has one per 16), and what they
//     cost the computing wave is not only their issue.
#pragma once
#ifndef ATTN32_ABL
#define ATTN32_ABL 0   // ablation mask of rounds 2-3 (one cost removed at a time); 0 = the kernel as it was measured
#endif
#include "../../image_search_amd/csrc/attn32_kernels.h"

namespace mi {

#ifndef ATTN64_Q0
#define ATTN64_Q0 1  // first half-step of the sweep that carries one of the next pair's query-fragment loads
#endif
#ifndef ATTN64_PACE
#define ATTN64_PACE 2  // s_sleep units (64 cycles) between two of a service wave's DMA pieces
#endif

// Both tiles of a wave swept over all key tiles of a compile-time token count (see attn32_sweep_static for the
// single-tile form: same addressing, same per-tile MFMA order, hence the same bits per query).
template <int S_CT, class Hook>
__device__ __forceinline__ void attn64_sweep_static(const unsigned char* __restrict__ Ks, const unsigned char* __restrict__ Vs,
                                                    const bf16x8 (&qa)[4], const bf16x8 (&qb)[4], int lane, v16f (&oa)[2],
                                                    v16f (&ob)[2], float& la, float& lb, Hook&& hook) {
    static_assert(S_CT > 0, "compile-time token count");
    constexpr int NKT = (S_CT + 31) / 32;
    constexpr bool ONE_KEY = (S_CT % 32) == 1;  // the last tile holds a single live key
    const int r = lane & 31, h = lane >> 5;
    const int g16 = lane >> 4, vq = (lane & 15) >> 2, vp = lane & 3;
    const int vc = 2 * (g16 & 1) + (vp >> 1);
    const unsigned char* kaddr[4];
    const unsigned char* vaddr[2][2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kaddr[ks] = Ks + r * 128 + (((2 * ks + h) ^ swz32(r)) << 4);
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int row = 4 * h + vq + 8 * w;
            vaddr[dt][w] = Vs + row * 128 + (((4 * dt + vc) ^ swz32(row)) << 4) + 8 * (vp & 1);
        }
    float lsa[4] = {0.0f, 0.0f, 0.0f, 0.0f}, lsb[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // this lane's share of the row sums, four chains
    bf16x8 kf[4];         // K fragments of the key tile whose scores come next (one buffer: see the loop)
    bf16x8 vf[2][2][2];   // vf[t & 1] = V fragments of key tile t: [d tile][16-key half]
    v16f sa, sb;          // scores of the tile in flight, then its numerators (in place)
    bf16x8 pa0, pa1, pb0, pb1;
    auto load_k = [&](bf16x8 (&dst)[4], int t) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dst[ks] = *reinterpret_cast<const bf16x8*>(kaddr[ks] + 4096 * t);
    };
    auto load_v = [&](bf16x8 (&dst)[2][2], int t, bool one_key) {
#pragma unroll
        for (int kp = 0; kp < 2; ++kp) {
            if (one_key && kp == 1) break;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(vaddr[dt][0] + 4096 * t + 2048 * kp));
                const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(vaddr[dt][1] + 4096 * t + 2048 * kp));
                dst[dt][kp] = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };
    auto zero = [&](v16f& s) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.0f;
    };
    // numerators of key tile t, in place, and their packed halves
    auto numerators = [&](v16f& s, bf16x8& p0, bf16x8& p1, float (&ls)[4], int t) {
        const bool last = t + 1 == NKT;
        if (last && ONE_KEY) {
            // key 32 t is element 0 of the lanes with h = 0; everything else of the tile is padding
            const float p = __builtin_amdgcn_exp2f(s[0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = 0.0f;
            s[0] = h == 0 ? p : 0.0f;
            ls[0] += s[0];
            p0 = pack8(s, 0);
            return;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float p = __builtin_amdgcn_exp2f(s[e]);
            if (last && (S_CT % 32) != 0) {
                const int key = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
                p = key < S_CT ? p : 0.0f;
            }
            s[e] = p;
            ls[e & 3] += p;
        }
        p0 = pack8(s, 0);
        p1 = pack8(s, 8);
    };
    // one half-step: P V of tile X at key tile t (numerators p0, p1), the scores of X at t + 1 into sx, and beside
    // them the numerators of the OTHER tile's pending scores (sy at key tile ty)
    auto half_step = [&](v16f (&ox)[2], v16f& sx, const bf16x8 (&qx)[4], const bf16x8& p0, const bf16x8& p1, int t,
                         v16f& sy, bf16x8& py0, bf16x8& py1, float (&lsy)[4], int ty) {
        const bool last = t + 1 == NKT;
        const bool one_key = last && ONE_KEY;
        const bf16x8 (&vt)[2][2] = vf[t & 1];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) ox[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt[dt][0], p0, ox[dt], 0, 0, 0);
        if (!one_key) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) ox[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt[dt][1], p1, ox[dt], 0, 0, 0);
        }
        if (!last) {
            zero(sx);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qx[ks], sx, 0, 0, 0);
        }
        if (ty < NKT) {
            numerators(sy, py0, py1, lsy, ty);
            // the numerators' users (the P V of the next half-step) sit behind the hook's branch, in another basic block: without
            // a user in THIS block the optimiser sinks the exp2s out of the MFMA gaps they are scheduled into below
            v4u u0 = __builtin_bit_cast(v4u, py0), u1 = __builtin_bit_cast(v4u, py1);
            asm volatile("" : "+v"(u0), "+v"(u1), "+v"(lsy[0]), "+v"(lsy[1]), "+v"(lsy[2]), "+v"(lsy[3]));  // (the row sums' adds too)
            py0 = __builtin_bit_cast(bf16x8, u0);
            py1 = __builtin_bit_cast(bf16x8, u1);
        }
        // the written order above is "all MFMAs, then all VALU"; the schedule asked for is one MFMA, then the VALU its
        // gap hides (two exp2 + two adds + a pack), eight times
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: K(0), V(0); scores of both tiles at key tile 0; numerators of a(0)
    load_k(kf, 0);
    load_v(vf[0], 0, NKT == 1 && ONE_KEY);
    zero(sa);
    zero(sb);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qa[ks], sa, 0, 0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qb[ks], sb, 0, 0, 0);
    numerators(sa, pa0, pa1, lsa, 0);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        // H1(t): P V a(t), scores a(t+1) | numerators b(t).   LDS: K(t+1) — the one K buffer is free (its last readers were
        // the scores b(t) at the end of H2(t-1)) and is next read behind the four P V MFMAs of this half-step
        hook(2 * t);
        if (t + 1 < NKT) load_k(kf, t + 1);
        half_step(oa, sa, qa, pa0, pa1, t, sb, pb0, pb1, lsb, t);
        // H2(t): P V b(t), scores b(t+1) | numerators a(t+1).  LDS: V(t+1) into the slot V(t-1) left at the end of H2(t-1)
        hook(2 * t + 1);
        if (t + 1 < NKT) load_v(vf[(t + 1) & 1], t + 1, t + 2 == NKT && ONE_KEY);
        half_step(ob, sb, qb, pb0, pb1, t, sa, pa0, pa1, lsa, t + 1);
    }
    // this lane's keys are those with (key >> 2) & 1 == h: the other half of the row sum sits 32 lanes away
    const float ha = (lsa[0] + lsa[1]) + (lsa[2] + lsa[3]), hb = (lsb[0] + lsb[1]) + (lsb[2] + lsb[3]);
    la += ha + __shfl_xor(ha, 32, 64);
    lb += hb + __shfl_xor(hb, 32, 64);
}

// Persistent, one workgroup of 8 waves per CU; see attn32_bf16_kernel for the pair loop this one mirrors.
// Two ROLES, one wave of each per SIMD (a workgroup's waves are dealt to the four SIMDs in turn, so wave c and wave
// c + 4 share one):
//   waves 0-3, compute: wave c owns query tiles 2c and 2c + 1 and runs the interleaved sweep above; its only vector-
//     memory work is its own (8 query-fragment loads for the next pair, spread over the sweep, and 8 ctx stores);
//   waves 4-7, service: everything that would stop that stream — the next pair's K/V by LDS-DMA (a piece costs the
//     issuing wave 60-185 cycles, MI355X_MICROARCH.md "LDS-DMA piece issue cost": 18 of them per wave and pair are more
//     than half a sweep), paced over the iteration so that the compute waves' own loads and stores do not queue
//     behind a burst; the split last query's nine per-key-tile partials; their combination a pair later.
// Measured with one wave per SIMD doing all of it (the first form of this kernel): sweep alone 75 us per layer,
// + ctx stores 15, + split query 25, + DMA and query loads 55 = 170 us; attn32 (8 symmetric waves) 148 us.
// cls_only (last layer): compute wave 0 alone sweeps (tile 0 holds the CLS query), through the very code the full
// layer uses, so the CLS-only last layer stays bit-identical to the full one.
#ifdef ATTN32_STAMPS   // diagnostic build of tools/probe/attn_bench.hip: per-segment shader cycles (attn32_stamp_buf is attn32's)
#define ATTN64_STAMP(SLOT) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if (lane == 0) acc_[SLOT] += now_ - last_; last_ = __builtin_amdgcn_s_memtime(); }
#define ATTN64_STAMP_INIT unsigned long long acc_[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#define ATTN64_STAMP_OUT { if (lane == 0) for (int j = 0; j < 6; ++j) attn32_stamp_buf[((size_t)blockIdx.x * 8 + wave) * 8 + j] = acc_[j]; }
#else
#define ATTN64_STAMP(SLOT)
#define ATTN64_STAMP_INIT
#define ATTN64_STAMP_OUT
#endif
template <int S_PAD, int S_CT, bool PRESCALED>
__global__ __launch_bounds__(512, 2) void attn64_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, int D,
                                                             int H, int n_pairs, int cls_only, int force_shift) {
    static_assert(S_CT == 257, "two full query tiles per compute wave and one split query: S = 8 x 32 + 1");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int S = S_CT;
    constexpr int IMG = 2 * S_PAD * 128;  // K then V
    constexpr int NPIECE = S_PAD / 8;     // 1-KiB DMA pieces per matrix
    static_assert(NPIECE == 36, "nine pieces per service wave");
    float* scratch = reinterpret_cast<float*>(smem + 2 * IMG);
    unsigned char* qsplit = reinterpret_cast<unsigned char*>(scratch + 2 * ATTN32_PART);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool service = wave >= 4;
    const int cw = wave & 3;  // compute: tiles 2 cw, 2 cw + 1; service: pieces cw, cw + 4, ... and key tiles cw, cw + 4
    const size_t ld = (size_t)3 * D;
    const int r = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    const bool split = !cls_only && !(ATTN32_ABL & 32);
    constexpr int split_row = S - 1;

    const int rr = lane >> 3, cp = lane & 7;
    const uint32_t voff0 = (uint32_t)rr * (uint32_t)(ld * 2) + 16u * (uint32_t)(cp ^ swz32(rr));
    const uint32_t voff1 = (uint32_t)rr * (uint32_t)(ld * 2) + 16u * (uint32_t)(cp ^ swz32(8 + rr));
    struct Pair { const bf16_t* base; bf16_t* ctx_b; uint32_t bytes; };
    auto pair_of = [&](int pr) {
        const int img = __builtin_amdgcn_readfirstlane(pr / H), hh = __builtin_amdgcn_readfirstlane(pr - img * H);
        Pair q;
        q.base = qkv + (size_t)img * S * ld + hh * 64;
        q.ctx_b = ctx + (size_t)img * S * D + hh * 64;
        q.bytes = (uint32_t)((size_t)S * ld * 2 - (size_t)hh * 128);
        return q;
    };
    auto dma_piece = [&](const Pair& pr, int b, int j) {
        // the descriptor must be PROVABLY wave-uniform or hipcc wraps every DMA in a waterfall loop
        const uintptr_t bp = reinterpret_cast<uintptr_t>(pr.base);
        const bf16_t* base = reinterpret_cast<const bf16_t*>(
            ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(bp >> 32)) << 32) |
            (uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)bp));
        const rsrc_t kvr = make_rsrc(base, (uint32_t)__builtin_amdgcn_readfirstlane((int)pr.bytes));
        unsigned char* Kd = smem + b * IMG;
        unsigned char* Vd = Kd + S_PAD * 128;
        const uint32_t so = (uint32_t)(8 * j) * (uint32_t)(ld * 2);
        const uint32_t vo = (j & 1) ? voff1 : voff0;
        glds16_buf(kvr, vo, so + (uint32_t)D * 2u, Kd + j * 1024);
        glds16_buf(kvr, vo, so + (uint32_t)D * 4u, Vd + j * 1024);
    };
    // query fragment i = 4 x + ks of tile x (0 / 1) of this compute wave
    auto load_q1 = [&](bf16x8 (&q)[2][4], const Pair& pr, int i) {
        const int x = i >> 2, ks = i & 3;
        q[x][ks] = *reinterpret_cast<const bf16x8*>(pr.base + (size_t)(64 * cw + 32 * x + r) * ld + 16 * ks + 8 * h);
    };
    auto scale_q = [&](bf16x8 (&q)[4]) {
        if constexpr (!PRESCALED) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) q[ks][e] = (__bf16)((float)q[ks][e] * ATTN32_C2);
        }
    };
    auto combine = [&](const float* part, bf16_t* ctx_prev) {
        float M = -INFINITY;
#pragma unroll
        for (int t = 0; t < 9; ++t) M = fmaxf(M, part[(t * 2) * ATTN32_PROW + 33]);
        float lt = 0.0f, wgt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float mt = part[(t * 2) * ATTN32_PROW + 33];
            wgt[t] = mt == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(mt - M);
            lt += wgt[t] * part[(t * 2) * ATTN32_PROW + 32];
        }
        const int dt = lane >> 5, w32 = lane & 31, hh = (w32 >> 2) & 1, reg = (w32 & 3) + 4 * (w32 >> 3);
        float acc = 0.0f;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc += wgt[t] * part[(t * 2 + hh) * ATTN32_PROW + dt * 16 + reg];
        ctx_prev[(size_t)split_row * D + lane] = f2bf(acc / lt);
    };
    auto top_barrier = [&] {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };

    int pair = blockIdx.x;
    if (pair >= n_pairs) return;
    Pair cur = pair_of(pair);
    const int last_it = (n_pairs - 1 - (int)blockIdx.x) / G;

    if (service) {
        // ------------------------------------------------------------------------------------------ service waves
        const bool qs_lane = split && wave == 7 && lane < 8;
        v4u qs_n = {0u, 0u, 0u, 0u};  // wave 7, lanes 0-7: the split query's row of the NEXT pair on its way to LDS
        for (int j = cw; j < NPIECE; j += 4) dma_piece(cur, 0, j);
        if (qs_lane) {
            qs_n = *reinterpret_cast<const v4u*>(cur.base + (size_t)split_row * ld + 8 * lane);
            *reinterpret_cast<v4u*>(qsplit + 16 * lane) = qs_n;
        }
        bf16_t* ctx_prev = nullptr;
        ATTN64_STAMP_INIT
#pragma unroll 1
        for (int it = 0; pair < n_pairs; pair += G, ++it) {
            const int b = it & 1;
            const unsigned char* Ks = smem + b * IMG;
            const unsigned char* Vs = Ks + S_PAD * 128;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's pieces of this pair's K/V (and partials) are in LDS
            ATTN64_STAMP(0)
            top_barrier();
            ATTN64_STAMP(1)
            const int next = pair + G;
            const bool more = next < n_pairs && !(ATTN32_ABL & 64);
            const Pair nxt = pair_of(next < n_pairs ? next : pair);
            if (more && qs_lane) qs_n = *reinterpret_cast<const v4u*>(nxt.base + (size_t)split_row * ld + 8 * lane);
            if (split && it > 0 && wave == 7) combine(scratch + (b ^ 1) * ATTN32_PART, ctx_prev);  // the previous pair's split query
            // the other LDS image is free (every wave passed the barrier after its last read of it): nine pieces, paced
            // (an iteration is ~6 k cycles; a burst would queue in the CU's one vector-memory pipe in front of the compute
            // waves' own loads and stores), with the split query's work between them
            // (issued first: a piece lands ~2 k cycles after its issue, and the top-of-loop wait of the next iteration must
            // not be the one to find that out)
            auto pieces = [&](int c0, int c1) {
                for (int c = c0; c < c1; ++c) {
                    if (more) dma_piece(nxt, b ^ 1, cw + 4 * c);
                    __builtin_amdgcn_s_sleep(ATTN64_PACE);
                }
            };
            ATTN64_STAMP(2)
            pieces(0, 9);
            ATTN64_STAMP(3)
            if (split) {
                bf16x8 qs[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qs[ks] = *reinterpret_cast<const bf16x8*>(qsplit + b * 128 + 32 * ks + 16 * h);
                scale_q(qs);
                float* part = scratch + b * ATTN32_PART;
                attn32_split_tiles(Ks, Vs, qs, cw, cw + 4, lane, part);
                if (cw == 0) attn32_split_one_key(Ks, Vs, qs, 8, lane, part);
                ctx_prev = cur.ctx_b;
            }
            if (qs_lane) *reinterpret_cast<v4u*>(qsplit + (b ^ 1) * 128 + 16 * lane) = qs_n;  // next pair's row, read behind the next barrier
            cur = nxt;
            ATTN64_STAMP(4)
        }
        ATTN64_STAMP_OUT
        if (split) {  // the last pair's split query
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            top_barrier();
            if (wave == 7) combine(scratch + (last_it & 1) * ATTN32_PART, ctx_prev);
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------- compute waves
    bf16x8 q_n[2][4];  // the NEXT pair's query fragments of this wave's two tiles
#pragma unroll
    for (int i = 0; i < 8; ++i) load_q1(q_n, cur, i);
    const bool active = !cls_only || wave == 0;  // waves that sweep (and store)
    ATTN64_STAMP_INIT
#pragma unroll 1
    for (int it = 0; pair < n_pairs; pair += G, ++it) {
        const int b = it & 1;
        const unsigned char* Ks = smem + b * IMG;
        const unsigned char* Vs = Ks + S_PAD * 128;
        // this wave's query fragments have landed; younger are only its 8 ctx stores of the previous iteration, which stay
        // in flight (raw s_barrier: __syncthreads() would drain them)
        if (it == 0 || !active || (ATTN32_ABL & 128)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        ATTN64_STAMP(0)
        top_barrier();
        ATTN64_STAMP(1)
        bf16x8 qa[4], qb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { qa[ks] = q_n[0][ks]; qb[ks] = q_n[1][ks]; }
        scale_q(qa);
        scale_q(qb);
        const int next = pair + G;
        const bool more = next < n_pairs && !(ATTN32_ABL & 1024);
        const Pair nxt = pair_of(next < n_pairs ? next : pair);
        // per half-step of the sweep: one of this wave's next query fragments
        // (c is a compile-time constant at every call: q_n must stay in registers)
        auto hook = [&](int c) {
            if (more && c >= ATTN64_Q0 && c < ATTN64_Q0 + 8) load_q1(q_n, nxt, c - ATTN64_Q0);
        };
        auto all_hooks = [&] {
#pragma unroll
            for (int c = 0; c < 18; ++c) hook(c);
        };
        if (active) {
            v16f oa[2], ob[2];
            auto clear = [&](v16f (&o)[2]) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[dt][e] = 0.0f;
            };
            clear(oa);
            clear(ob);
            float la = 0.0f, lb = 0.0f;
            bool shifted = force_shift != 0;
            if (shifted) all_hooks();  // what the sweep would have issued, in front of the stores (vmcnt order)
            if (!shifted) {
                ATTN64_STAMP(2)
                attn64_sweep_static<S_CT>(Ks, Vs, qa, qb, lane, oa, ob, la, lb, hook);
                ATTN64_STAMP(3)
                shifted = !ATTN32_ABL && __any(!(la > ATTN32_L_LO && la < ATTN32_L_HI) || !(lb > ATTN32_L_LO && lb < ATTN32_L_HI));
            }
            if (shifted) {  // rare: a numerator left the exponent range (or the caller asked for the shifted pass)
                auto nohook = [] {};
                clear(oa);
                clear(ob);
                la = lb = 0.0f;
                const float na = attn32_rowmax<S_CT>(Ks, qa, S, 0, 1, lane);
                attn32_sweep<S_CT, true>(Ks, Vs, qa, na == INFINITY ? 0.0f : na, S, 0, 1, lane, oa, la, nohook);
                const float nb = attn32_rowmax<S_CT>(Ks, qb, S, 0, 1, lane);
                attn32_sweep<S_CT, true>(Ks, Vs, qb, nb == INFINITY ? 0.0f : nb, S, 0, 1, lane, ob, lb, nohook);
            }
            attn32_store(oa, 1.0f / la, cur.ctx_b, 64 * cw + r, true, D, lane);
            attn32_store(ob, 1.0f / lb, cur.ctx_b, 64 * cw + 32 + r, true, D, lane);
        } else {
            all_hooks();
        }
        cur = nxt;
        ATTN64_STAMP(4)
    }
    ATTN64_STAMP_OUT
    if (split) top_barrier();  // the service waves' last rendezvous (the last pair's partials)
}

}  // namespace mi
