/*
 * oracle.c — CPU restatement of the reference's hot path (byte/float arithmetic).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under image_search_amd/ may link, import
 * or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported CPU baseline.
 *
 * PARITY UNPINNED: the reference (olFi95/image_search) holds no golden vector
 * for the ViT or the kNN.  Its only numeric known answer is
 * average_slices([1,2,4,4,10],[1,1,2,4,0]) = [1,1.5,3,4,5]
 * (server/src/search.rs:157-160), which tests/test_oracle.py checks.
 * The kNN arithmetic lives in an external SurrealDB server (image tag "latest",
 * client crate surrealdb 2.3.7, Cargo.lock:9180) whose source is absent; this
 * file restates the published definition  dist = 1 - a.b / (|a| |b|)  with
 * K nearest returned in ascending distance (server/src/search.rs:70-77,
 * index DDL server/src/clip.rs:140-143: MTREE DIMENSION 768 DIST COSINE TYPE F32)
 * and FIXES a summation order so that GPU and CPU agree bit for bit.
 *
 * Summation order (the contract both sides implement), for a row x and query q
 * of length dim (dim % 64 == 0):
 *   p[m]  = fmaf-chain over t = 0 .. dim/64-1 of q[64t+m]*x[64t+m]   (m = 0..63)
 *   then a pairwise xor-butterfly over m with offsets 1,2,4,8,16,32:
 *   p[m] <- p[m] + p[m^off]   (all m simultaneously; fp add is commutative, so
 *   every m ends with the same bits).  Same for x.x and q.q.
 *   dist = 1.0f - dot / (sqrtf(qq) * sqrtf(xx))        (each op IEEE fp32)
 * Ordering: (dist asc, id asc); NaN (zero-norm row or query) sorts last.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- seeded counter-based generator (the build's own) ---- */

static inline uint64_t orc_mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

/* value(seed, i) = (sum of the four 16-bit fields of h - 131070) * scale,
 * h = mix64(i + mix64(seed + GOLDEN)); integer-only up to one fp32 multiply,
 * so every platform produces the same bits. */
static inline float orc_gen1(uint64_t key, uint64_t i, float scale) {
    uint64_t h = orc_mix64(i + key);
    int32_t s = (int32_t)((h & 0xffff) + ((h >> 16) & 0xffff) + ((h >> 32) & 0xffff) + (h >> 48));
    return (float)(s - 131070) * scale;
}

uint64_t orc_gen_key(uint64_t seed) { return orc_mix64(seed + 0x9E3779B97F4A7C15ULL); }

/* scale that gives the requested standard deviation */
float orc_gen_scale(double std) { return (float)(std / sqrt(1431655765.0)); }

void orc_gen_f32(uint64_t seed, uint64_t first, uint64_t n, float scale, float* out) {
    const uint64_t key = orc_gen_key(seed);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = orc_gen1(key, first + (uint64_t)i, scale);
}

/* ---- average_slices: server/src/search.rs:127-150 ---- */
/* zero-init, add in input order, then divide by (m as f32).  Returns -1 where
 * the reference asserts (empty input); ragged lengths cannot be expressed
 * through this signature and are checked by the caller. */
int orc_average_slices(const float* const* vecs, size_t m, size_t len, float* out) {
    if (m == 0) return -1;
    for (size_t i = 0; i < len; ++i) out[i] = 0.0f;
    for (size_t v = 0; v < m; ++v)
        for (size_t i = 0; i < len; ++i) out[i] += vecs[v][i];
    const float count = (float)m;
    for (size_t i = 0; i < len; ++i) out[i] /= count;
    return 0;
}

/* ---- refine step of web_search_text: server/src/search.rs:28, :60-67 ---- */
/* m == 0 -> query = text; else query = average([average(selected), text]) */
int orc_refine(const float* text, const float* const* selected, size_t m, size_t len, float* out) {
    if (m == 0) { memcpy(out, text, len * sizeof(float)); return 0; }
    float* sel = (float*)malloc(len * sizeof(float));
    if (!sel) return -2;
    orc_average_slices(selected, m, len, sel);
    const float* two[2] = { sel, text };
    orc_average_slices(two, 2, len, out);
    free(sel);
    return 0;
}

/* ---- image_prepare_resnet, arithmetic part: server/src/clip.rs:158-172 ---- */
/* input: RGB8 interleaved 224x224 (after the resize the `image` crate does);
 * output planar CHW f32: data[c*50176 + i] = (p/255 - mean[c]) / std[c] */
void orc_preprocess_rgb8(const uint8_t* hwc, size_t n_images, float* chw) {
    const float mean[3] = { 0.485f, 0.456f, 0.406f };
    const float sd[3] = { 0.229f, 0.224f, 0.225f };
    const size_t P = 224 * 224;
    for (size_t n = 0; n < n_images; ++n)
        for (size_t i = 0; i < P; ++i)
            for (int c = 0; c < 3; ++c) {
                float v = (float)hwc[n * P * 3 + i * 3 + c] / 255.0f;
                chw[n * P * 3 + (size_t)c * P + i] = (v - mean[c]) / sd[c];
            }
}

/* ---- image_prepare_resnet, resize part: server/src/clip.rs:154-155 ----
 * `img.resize_exact(224, 224, FilterType::CatmullRom)` then `.to_rgb8()`.  The resampler lives in the
 * un-vendored `image` crate, pinned at 0.25.8 (Cargo.lock:5008-5009); its source is absent from
 * /root/reference, so this restates the published algorithm of image-0.25.8 src/imageops/sample.rs
 * (`resize` -> `vertical_sample` into an f32 image -> `horizontal_sample` with clamp + round-to-nearest,
 * kernel `catmullrom_kernel(x) = bc_cubic_spline(x, 0.0, 0.5)`, support 2.0) -- PARITY UNPINNED: no test
 * or fixture of the reference pins a resized pixel.  Every operation is fp32 in the order
 * the crate writes it: weights w_i = k((i - (centre - 0.5)) / sratio) for i in [left, right), their sum
 * accumulated in that order, each weight divided by the sum, then t += p_i * w_i in that order.
 * RGB8 input only (what a decoded JPEG is); resizing RGBA8 / L8 and converting afterwards gives the same
 * RGB bytes because the crate filters every channel independently. */
static inline float orc_catmullrom(float x) {
    const float a = fabsf(x);
    float k;
    if (a < 1.0f) k = (9.0f * ((a * a) * a) + -15.0f * (a * a)) + 6.0f;
    else if (a < 2.0f) k = ((-3.0f * ((a * a) * a) + 15.0f * (a * a)) + -24.0f * a) + 12.0f;
    else k = 0.0f;
    return k / 6.0f;
}

/* window [left, right) and the centre used by the kernel, for output index `o` of `n_out` over `n_in` */
static inline void orc_window(uint32_t o, uint32_t n_in, uint32_t n_out, int64_t* left, int64_t* right,
                              float* centre, float* sratio) {
    const float ratio = (float)n_in / (float)n_out;
    *sratio = ratio < 1.0f ? 1.0f : ratio;
    const float support = 2.0f * *sratio;
    const float in = ((float)o + 0.5f) * ratio;
    int64_t l = (int64_t)floorf(in - support);
    if (l < 0) l = 0;
    if (l > (int64_t)n_in - 1) l = (int64_t)n_in - 1;
    int64_t r = (int64_t)ceilf(in + support);
    if (r < l + 1) r = l + 1;
    if (r > (int64_t)n_in) r = (int64_t)n_in;
    *left = l; *right = r; *centre = in - 0.5f;
}

/* src [h][w][3] u8 -> dst [nh][nw][3] u8; returns 0, or -1 on bad sizes / allocation failure */
int orc_resize_catmullrom_rgb8(const uint8_t* src, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint8_t* dst) {
    if (nw == 0 || nh == 0) return -1;
    if (w == 0 || h == 0) { memset(dst, 0, (size_t)nw * nh * 3); return 0; }  /* ImageBuffer::new: zeros */
    if (w == nw && h == nh) { memcpy(dst, src, (size_t)w * h * 3); return 0; }
    float* tmp = (float*)malloc((size_t)nh * w * 3 * sizeof(float));
    if (!tmp) return -1;
    float* ws = (float*)malloc(((size_t)(h > w ? h : w) + 1) * sizeof(float));
    if (!ws) { free(tmp); return -1; }
    for (uint32_t oy = 0; oy < nh; ++oy) {  /* vertical_sample */
        int64_t l, r; float c, sr;
        orc_window(oy, h, nh, &l, &r, &c, &sr);
        float sum = 0.0f;
        for (int64_t i = l; i < r; ++i) { ws[i - l] = orc_catmullrom(((float)i - c) / sr); sum += ws[i - l]; }
        for (int64_t i = l; i < r; ++i) ws[i - l] /= sum;
        for (size_t e = 0; e < (size_t)w * 3; ++e) {
            float t = 0.0f;
            for (int64_t i = l; i < r; ++i) t += (float)src[(size_t)i * w * 3 + e] * ws[i - l];
            tmp[(size_t)oy * w * 3 + e] = t;
        }
    }
    for (uint32_t ox = 0; ox < nw; ++ox) {  /* horizontal_sample */
        int64_t l, r; float c, sr;
        orc_window(ox, w, nw, &l, &r, &c, &sr);
        float sum = 0.0f;
        for (int64_t i = l; i < r; ++i) { ws[i - l] = orc_catmullrom(((float)i - c) / sr); sum += ws[i - l]; }
        for (int64_t i = l; i < r; ++i) ws[i - l] /= sum;
        for (uint32_t y = 0; y < nh; ++y)
            for (int ch = 0; ch < 3; ++ch) {
                float t = 0.0f;
                for (int64_t i = l; i < r; ++i) t += tmp[((size_t)y * w + (size_t)i) * 3 + ch] * ws[i - l];
                t = t < 0.0f ? 0.0f : (t > 255.0f ? 255.0f : t);  /* clamp(t, min, max); NaN cannot occur */
                dst[((size_t)y * nw + ox) * 3 + ch] = (uint8_t)roundf(t);  /* FloatNearest: half away from zero */
            }
    }
    free(ws); free(tmp);
    return 0;
}

/* the whole of image_prepare_resnet (server/src/clip.rs:153-175): one RGB8 image of any size -> CHW f32 */
int orc_image_prepare_resnet(const uint8_t* src, uint32_t w, uint32_t h, float* chw) {
    uint8_t* r = (uint8_t*)malloc((size_t)224 * 224 * 3);
    if (!r) return -1;
    const int rc = orc_resize_catmullrom_rgb8(src, w, h, 224, 224, r);
    if (rc == 0) orc_preprocess_rgb8(r, 1, chw);
    free(r);
    return rc;
}

/* ---- cosine distance in the fixed summation order ---- */

static inline void orc_dot2(const float* q, const float* x, uint32_t dim, float* dot, float* xx) {
    float pd[64], px[64];
    for (int m = 0; m < 64; ++m) { pd[m] = 0.0f; px[m] = 0.0f; }
    for (uint32_t t = 0; t < dim / 64; ++t)
        for (int m = 0; m < 64; ++m) {
            const float a = q[64 * t + m], b = x[64 * t + m];
            pd[m] = fmaf(a, b, pd[m]);
            px[m] = fmaf(b, b, px[m]);
        }
    for (int off = 1; off < 64; off <<= 1) {
        float nd[64], nx[64];
        for (int m = 0; m < 64; ++m) { nd[m] = pd[m] + pd[m ^ off]; nx[m] = px[m] + px[m ^ off]; }
        memcpy(pd, nd, sizeof pd); memcpy(px, nx, sizeof px);
    }
    *dot = pd[0]; *xx = px[0];
}

float orc_sumsq(const float* q, uint32_t dim) { float d, s; orc_dot2(q, q, dim, &d, &s); return s; }

static inline float orc_dist(float dot, float qq, float xx) {
    const float d = 1.0f - dot / (sqrtf(qq) * sqrtf(xx));
    return d != d ? __builtin_nanf("") : d; /* one NaN: +qNaN 0x7FC00000 (sign/payload carry no meaning) */
}

/* monotone u32 key of a distance: ascending key == ascending distance, NaN last */
static inline uint32_t orc_key(float d) {
    uint32_t b; memcpy(&b, &d, 4);
    if (d != d) return 0xFFFFFFFFu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

void orc_cosine_dist(const float* q, const float* rows, uint64_t n, uint32_t dim, float* dist) {
    const float qq = orc_sumsq(q, dim);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)n; ++r) {
        float dot, xx;
        orc_dot2(q, rows + (uint64_t)r * dim, dim, &dot, &xx);
        dist[r] = orc_dist(dot, qq, xx);
    }
}

typedef struct { uint32_t key; uint64_t id; float d; } orc_ent;

static inline int orc_less(const orc_ent* a, const orc_ent* b) {
    return a->key < b->key || (a->key == b->key && a->id < b->id);
}

/* insert into a sorted array of at most k entries */
static void orc_insert(orc_ent* best, uint32_t* cnt, uint32_t k, orc_ent e) {
    if (*cnt == k && !orc_less(&e, &best[k - 1])) return;
    uint32_t pos = (*cnt < k) ? (*cnt)++ : k - 1;
    while (pos > 0 && orc_less(&e, &best[pos - 1])) { best[pos] = best[pos - 1]; --pos; }
    best[pos] = e;
}

/* brute-force top-k of one query over rows [0,n) whose global ids are base + r.
 * Outputs k entries; missing ones (n < k) are idx = UINT64_MAX, dist = +inf. */
int orc_knn(const float* q, const float* rows, uint64_t n, uint32_t dim, uint32_t k,
            uint64_t base, uint64_t* idx, float* dist) {
    if (dim == 0 || dim % 64 != 0 || k == 0) return -1;
    const float qq = orc_sumsq(q, dim);
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    orc_ent* all = (orc_ent*)malloc((size_t)nt * k * sizeof(orc_ent));
    uint32_t* cnts = (uint32_t*)calloc((size_t)nt, sizeof(uint32_t));
    if (!all || !cnts) { free(all); free(cnts); return -2; }
#pragma omp parallel
    {
        int t = 0;
#ifdef _OPENMP
        t = omp_get_thread_num();
#endif
        orc_ent* best = all + (size_t)t * k;
        uint32_t cnt = 0;
#pragma omp for schedule(static)
        for (int64_t r = 0; r < (int64_t)n; ++r) {
            float dot, xx;
            orc_dot2(q, rows + (uint64_t)r * dim, dim, &dot, &xx);
            orc_ent e; e.d = orc_dist(dot, qq, xx); e.key = orc_key(e.d); e.id = base + (uint64_t)r;
            orc_insert(best, &cnt, k, e);
        }
        cnts[t] = cnt;
    }
    orc_ent* fin = (orc_ent*)malloc((size_t)k * sizeof(orc_ent));
    uint32_t fc = 0;
    for (int t = 0; t < nt; ++t)
        for (uint32_t i = 0; i < cnts[t]; ++i) orc_insert(fin, &fc, k, all[(size_t)t * k + i]);
    for (uint32_t i = 0; i < k; ++i) {
        if (i < fc) { idx[i] = fin[i].id; dist[i] = fin[i].d; }
        else { idx[i] = UINT64_MAX; dist[i] = INFINITY; }
    }
    free(fin); free(all); free(cnts);
    return 0;
}

/* merge `lists` candidate lists of k entries each (what every rank holds after
 * the all-gather) into the global top-k: same ordering rule. */
int orc_merge(const uint64_t* idx_in, const float* dist_in, uint32_t lists, uint32_t k,
              uint64_t* idx, float* dist) {
    orc_ent* fin = (orc_ent*)malloc((size_t)k * sizeof(orc_ent));
    if (!fin) return -2;
    uint32_t fc = 0;
    for (uint32_t i = 0; i < lists * k; ++i) {
        if (idx_in[i] == UINT64_MAX) continue;
        orc_ent e; e.d = dist_in[i]; e.key = orc_key(e.d); e.id = idx_in[i];
        orc_insert(fin, &fc, k, e);
    }
    for (uint32_t i = 0; i < k; ++i) {
        if (i < fc) { idx[i] = fin[i].id; dist[i] = fin[i].d; }
        else { idx[i] = UINT64_MAX; dist[i] = INFINITY; }
    }
    free(fin);
    return 0;
}

/* fp64 distances (plain left-to-right sums) — used only to report how far the
 * fp32 order above sits from the real-number answer and the k/k+1 margins. */
void orc_cosine_dist_f64(const float* q, const float* rows, uint64_t n, uint32_t dim, double* dist) {
    double qq = 0; for (uint32_t j = 0; j < dim; ++j) qq += (double)q[j] * q[j];
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)n; ++r) {
        const float* x = rows + (uint64_t)r * dim;
        double d = 0, xx = 0;
        for (uint32_t j = 0; j < dim; ++j) { d += (double)q[j] * x[j]; xx += (double)x[j] * x[j]; }
        dist[r] = 1.0 - d / (sqrt(qq) * sqrt(xx));
    }
}

int orc_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
