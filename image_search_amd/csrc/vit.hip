// vit.hip — host side of Seam A behind the C ABI (include/mi355clip.h):
// mi_clip_load  <- clip::clip_vit_large_patch14::Model::from_file (server/src/clip.rs:46-48)
// mi_clip_embed <- model.forward + to_data                         (server/src/clip.rs:112-124)
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "handles.h"
#include "weights.h"
#include "../../include/mi355clip_ops.h"
#include "preprocess_kernels.h"
#include "vit_kernels.h"
#include "attn32_kernels.h"

using namespace mi;

namespace {

inline uint16_t f32_to_bf16_host(float f) {  // round to nearest even, NaN stays NaN
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

}  // namespace

namespace {

size_t esize(const mi_clip* m) { return m->precision == MI_PRECISION_F32 ? 4 : 2; }

size_t pad256(size_t v) { return (v + 255) / 256 * 256; }

// Row pitch (elements) of the image tower's qkv buffer.  Where the persistent attn32 kernel runs, rows are padded by
// option "qkv_pad" elements (default 128 = 256 bytes; a multiple of 64 so that the GEMM epilogue still stores whole
// 128-byte lines): dense 6 144-byte rows put a head's K / V / q pieces on few memory channels and cost that kernel 15-20 %.
bool attn32_applies(const mi_clip* m) {
    return m->precision != MI_PRECISION_F32 && m->attn_ver >= 2 && !m->text && m->S > 64 && m->S <= 288;
}
size_t qkv_pitch(const mi_clip* m) { return (size_t)3 * m->D + (attn32_applies(m) ? (size_t)m->qkv_pad : 0); }
// Option "qkv_layout" = 1: q|k|v as head-major planes [3][H][Mp][64] instead of token rows — the persistent GEMM's epilogue
// writes one contiguous KiB per store (PpFold::col_stride) and attn32 fetches a head's K / V / q as contiguous blocks.
// Same values at other addresses: the forward is bit-identical.  The buffer (sized for the padded rows) holds either.
bool qkv_head_major(const mi_clip* m) {
    return m->qkv_layout == 1 && attn32_applies(m) && m->precision == MI_PRECISION_BF16 && !m->split_ln && m->D % 256 == 0;
}


template <typename T>
T* dalloc(mi_clip* m, size_t n, std::vector<void*>& bag) {
    void* p = nullptr;
    HIP_CHECK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
    bag.push_back(p);
    (void)m;
    return static_cast<T*>(p);
}

float* upload_f32(mi_clip* m, const std::vector<float>& h) {
    float* d = dalloc<float>(m, h.size(), m->allocs);
    HIP_CHECK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    return d;
}

// GEMM operand in the model's precision; dup_k > 0: every row of k = dup_k values is stored twice, [W | W]
// (the operand of a GEMM whose activations arrive as hi | lo halves, MI_PRECISION_BF16_SPLIT)
void* upload_mat(mi_clip* m, const std::vector<float>& h, size_t dup_k = 0) {
    if (m->precision == MI_PRECISION_F32) return upload_f32(m, h);
    std::vector<uint16_t> b(h.size() * (dup_k ? 2 : 1));
    if (dup_k) {
        for (size_t r = 0; r < h.size() / dup_k; ++r)
            for (size_t c = 0; c < dup_k; ++c) b[r * 2 * dup_k + c] = b[r * 2 * dup_k + dup_k + c] = f32_to_bf16_host(h[r * dup_k + c]);
    } else
    for (size_t i = 0; i < h.size(); ++i) b[i] = f32_to_bf16_host(h[i]);
    uint16_t* d = dalloc<uint16_t>(m, b.size(), m->allocs);
    HIP_CHECK(hipMemcpy(d, b.data(), b.size() * 2, hipMemcpyHostToDevice));
    return d;
}

void load_layers(mi_clip* m, WeightFile& st, const std::string& prefix);

// Every reader of the residual stream is a LayerNorm (LN1, LN2, post_layernorm: modeling_clip.py:353-383, :641-651), and a
// LayerNorm does not see a constant added to all channels of a row.  So the common mode of everything WRITTEN to the stream
// can be removed without changing the function: W <- (I - 11^T / N) W (every column loses its mean over the N output
// channels), b <- b - mean(b), for out_proj and fc2 (here, at load) and the row mean of the pre-LayerNorm's output
// (embed_ln_kernel).  The rows of the stream then have mean ~ 0, which is what the LayerNorm-free loop needs: it rounds the
// UN-normalised row to bf16, and a row r sigma off zero pays r times the LayerNorm tower's rounding (measured, DESIGN.md 3.1:
// in bound up to r = 4, 3 x out of it at r = 16).  Option "ln_center" at load (MI_CLIP_LN_CENTER=0 keeps the weights as read).
void center_writer(std::vector<float>& w, std::vector<float>& b, int N, int K) {
    std::vector<double> cm((size_t)K, 0.0);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) cm[(size_t)k] += (double)w[(size_t)n * K + k];
    for (int k = 0; k < K; ++k) cm[(size_t)k] /= (double)N;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) w[(size_t)n * K + k] = (float)((double)w[(size_t)n * K + k] - cm[(size_t)k]);
    double bm = 0.0;
    for (int n = 0; n < N; ++n) bm += (double)b[(size_t)n];
    bm /= (double)N;
    for (int n = 0; n < N; ++n) b[(size_t)n] = (float)((double)b[(size_t)n] - bm);
}

// LayerNorm(gamma, beta) in front of the linear (W [N][K], b): W' = bf16(W diag(gamma)), c[n] = sum_k W'[n][k] (of the
// ROUNDED weights: the epilogue subtracts mean * c from what the MFMAs accumulated over exactly those), b' = W beta + b.
// y = W LN(x) + b = rstd * (W' x - mean * c) + b'   (gemm_bf16_pp_kernel<EPI_LNF>)
void fold_ln(mi_clip* m, const std::vector<float>& w, const std::vector<float>& b, const std::vector<float>& gamma,
             const std::vector<float>& beta, int N, int K, void** wf, float** c, float** bf) {
    std::vector<uint16_t> w16((size_t)N * K);
    std::vector<float> cv(N), bv(N);
    for (int n = 0; n < N; ++n) {
        double cs = 0.0, bs = (double)b[n];
        for (int k = 0; k < K; ++k) {
            const float v = w[(size_t)n * K + k] * gamma[k];
            const uint16_t h = f32_to_bf16_host(v);
            w16[(size_t)n * K + k] = h;
            uint32_t u = (uint32_t)h << 16;
            float r;
            std::memcpy(&r, &u, 4);
            cs += (double)r;
            bs += (double)w[(size_t)n * K + k] * (double)beta[k];
        }
        cv[n] = (float)cs;
        bv[n] = (float)bs;
    }
    uint16_t* d = dalloc<uint16_t>(m, w16.size(), m->allocs);
    HIP_CHECK(hipMemcpy(d, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    *wf = d;
    *c = upload_f32(m, cv);
    *bf = upload_f32(m, bv);
}

// The text tower of the same checkpoint family (HF `CLIPTextModelWithProjection` names): what the
// reference reaches through embed_anything (server/src/clip.rs:19-23, :35-40).  fp32 only: one
// query is 77 token rows, latency-bound, and the causal mask exists in the fp32 attention kernel.
void load_text_weights(mi_clip* m, const char* path) {
    const std::unique_ptr<WeightFile> file = open_weights(path);
    WeightFile& st = *file;
    const std::string t = "text_model.";
    const TensorInfo& te = st.info(t + "embeddings.token_embedding.weight");
    if (te.shape.size() != 2) fail(MI_ERR_UNSUPPORTED, "token_embedding.weight must be [V,D]");
    m->vocab = (int)te.shape[0];
    m->D = (int)te.shape[1];
    const TensorInfo& po = st.info(t + "embeddings.position_embedding.weight");
    if (po.shape.size() != 2 || po.shape[1] != m->D) fail(MI_ERR_UNSUPPORTED, "position_embedding.weight must be [S,D]");
    m->S = (int)po.shape[0];
    int L = 0;
    while (st.has(t + "encoder.layers." + std::to_string(L) + ".layer_norm1.weight")) ++L;
    if (L == 0) fail(MI_ERR_IO, "no text encoder layers found in '%s'", path);
    m->L = L;
    m->FF = (int)st.info(t + "encoder.layers.0.mlp.fc1.weight").shape.at(0);
    m->E = (int)st.info("text_projection.weight").shape.at(0);
    m->H = m->D / 64;
    auto it = st.meta.find("num_attention_heads");
    if (it != st.meta.end()) m->H = std::atoi(it->second.c_str());
    if (m->H <= 0 || m->D != m->H * 64)
        fail(MI_ERR_UNSUPPORTED, "hidden %d with %d heads: the attention kernels are built for head_dim 64", m->D, m->H);
    if (m->D % 128 != 0 || m->FF % 128 != 0)
        fail(MI_ERR_UNSUPPORTED, "hidden (%d) and intermediate (%d) sizes must be multiples of 128", m->D, m->FF);
    m->image = m->patch = m->grid = 0;
    const int D = m->D;
    m->tok = upload_f32(m, st.read(t + "embeddings.token_embedding.weight", (int64_t)m->vocab * D));
    m->pos = upload_f32(m, st.read(t + "embeddings.position_embedding.weight", (int64_t)m->S * D));
    m->post_w = upload_f32(m, st.read(t + "final_layer_norm.weight", D));
    m->post_b = upload_f32(m, st.read(t + "final_layer_norm.bias", D));
    m->proj = upload_f32(m, st.read("text_projection.weight", (int64_t)m->E * D));
    load_layers(m, st, t);
}

void load_weights(mi_clip* m, const char* path) {
    const std::unique_ptr<WeightFile> file = open_weights(path);  // safetensors or a Burn .mpk record, by content
    WeightFile& st = *file;
    const std::string v = "vision_model.";
    const TensorInfo& pe = st.info(v + "embeddings.patch_embedding.weight");
    if (pe.shape.size() != 4 || pe.shape[1] != 3 || pe.shape[2] != pe.shape[3])
        fail(MI_ERR_UNSUPPORTED, "patch_embedding.weight must be [D,3,P,P]");
    m->D = (int)pe.shape[0];
    m->patch = (int)pe.shape[2];
    const TensorInfo& po = st.info(v + "embeddings.position_embedding.weight");
    if (po.shape.size() != 2 || po.shape[1] != m->D) fail(MI_ERR_UNSUPPORTED, "position_embedding.weight must be [S,D]");
    m->S = (int)po.shape[0];
    m->grid = (int)std::lround(std::sqrt((double)(m->S - 1)));
    if (m->grid * m->grid + 1 != m->S) fail(MI_ERR_UNSUPPORTED, "token count %d is not G*G+1", m->S);
    m->image = m->grid * m->patch;
    int L = 0;
    while (st.has(v + "encoder.layers." + std::to_string(L) + ".layer_norm1.weight")) ++L;
    if (L == 0) fail(MI_ERR_IO, "no encoder layers found in '%s'", path);
    m->L = L;
    m->FF = (int)st.info(v + "encoder.layers.0.mlp.fc1.weight").shape.at(0);
    m->E = (int)st.info("visual_projection.weight").shape.at(0);
    m->H = m->D / 64;
    auto it = st.meta.find("num_attention_heads");
    if (it != st.meta.end()) m->H = std::atoi(it->second.c_str());
    if (m->H <= 0 || m->D != m->H * 64)
        fail(MI_ERR_UNSUPPORTED, "hidden %d with %d heads: the attention kernels are built for head_dim 64", m->D, m->H);
    if (m->D % 128 != 0 || m->FF % 128 != 0)
        fail(MI_ERR_UNSUPPORTED, "hidden (%d) and intermediate (%d) sizes must be multiples of 128", m->D, m->FF);
    const int K = 3 * m->patch * m->patch;
    m->Kp = (K + 63) / 64 * 64;

    const int D = m->D;
    m->cls = upload_f32(m, st.read(v + "embeddings.class_embedding", D));
    m->pos = upload_f32(m, st.read(v + "embeddings.position_embedding.weight", (int64_t)m->S * D));
    m->pre_w = upload_f32(m, st.read(v + "pre_layrnorm.weight", D));
    m->pre_b = upload_f32(m, st.read(v + "pre_layrnorm.bias", D));
    m->post_w = upload_f32(m, st.read(v + "post_layernorm.weight", D));
    m->post_b = upload_f32(m, st.read(v + "post_layernorm.bias", D));
    m->proj = upload_f32(m, st.read("visual_projection.weight", (int64_t)m->E * D));
    {
        const std::vector<float> w = st.read(v + "embeddings.patch_embedding.weight", (int64_t)D * K);
        std::vector<float> wp((size_t)D * m->Kp, 0.0f);
        for (int d = 0; d < D; ++d) std::memcpy(&wp[(size_t)d * m->Kp], &w[(size_t)d * K], (size_t)K * 4);
        m->wpatch = upload_mat(m, wp);
    }
    m->q_prescaled = m->precision == MI_PRECISION_BF16 && m->attn_ver == 2 && m->S > 64 && m->S <= 288;
    // the LayerNorm-free layer loop needs the persistent GEMM on all four linears (256-wide tiles) and a last layer to
    // hand over to; the folded copies of W_qkv / W_fc1 are built beside the plain ones (option "ln_fold" switches per forward)
    m->fold_ready = m->precision == MI_PRECISION_BF16 && !m->split_ln && m->D % 256 == 0 && m->FF % 256 == 0 && m->L >= 2;
    m->ln_center = m->fold_ready && m->ln_center;
    if (m->fold_ready) {
        m->d_fold_offset_rows = dalloc<unsigned long long>(m, 1, m->allocs);
        HIP_CHECK(hipMemset(m->d_fold_offset_rows, 0, sizeof(unsigned long long)));
    }
    load_layers(m, st, v);
}

// the encoder blocks: same tensor names under "vision_model." and "text_model."
void load_layers(mi_clip* m, WeightFile& st, const std::string& v) {
    const int D = m->D, FF = m->FF, L = m->L;
    m->layers.resize(L);
    for (int i = 0; i < L; ++i) {
        const std::string p = v + "encoder.layers." + std::to_string(i) + ".";
        Layer& ly = m->layers[i];
        ly.ln1w = upload_f32(m, st.read(p + "layer_norm1.weight", D));
        ly.ln1b = upload_f32(m, st.read(p + "layer_norm1.bias", D));
        ly.ln2w = upload_f32(m, st.read(p + "layer_norm2.weight", D));
        ly.ln2b = upload_f32(m, st.read(p + "layer_norm2.bias", D));
        std::vector<float> wqkv, bqkv;
        for (const char* n : {"q_proj", "k_proj", "v_proj"}) {
            auto w = st.read(p + "self_attn." + n + ".weight", (int64_t)D * D);
            auto b = st.read(p + "self_attn." + n + ".bias", D);
            if (m->q_prescaled && n[0] == 'q') {
                // attn32 takes q in the exp2 domain: log2(e)/8 goes into W_q and b_q here, in fp32, before the
                // one bf16 rounding of the weights (a constant factor does not change their relative error)
                for (auto& x : w) x *= ATTN32_C2;
                for (auto& x : b) x *= ATTN32_C2;
            }
            wqkv.insert(wqkv.end(), w.begin(), w.end());
            bqkv.insert(bqkv.end(), b.begin(), b.end());
        }
        ly.wqkv = upload_mat(m, wqkv, m->split_ln ? (size_t)D : 0);
        ly.bqkv = upload_f32(m, bqkv);
        {
            std::vector<float> wo = st.read(p + "self_attn.out_proj.weight", (int64_t)D * D), bo = st.read(p + "self_attn.out_proj.bias", D);
            if (m->ln_center) center_writer(wo, bo, D, D);
            ly.wo = upload_mat(m, wo);
            ly.bo = upload_f32(m, bo);
        }
        const std::vector<float> w1 = st.read(p + "mlp.fc1.weight", (int64_t)FF * D), b1 = st.read(p + "mlp.fc1.bias", FF);
        ly.w1 = upload_mat(m, w1, m->split_ln ? (size_t)D : 0);
        ly.b1 = upload_f32(m, b1);
        if (m->fold_ready) {  // (the last layer folds LN1 only: behind its attention it works on the CLS rows, forward())
            fold_ln(m, wqkv, bqkv, st.read(p + "layer_norm1.weight", D), st.read(p + "layer_norm1.bias", D), 3 * D, D, &ly.wqkv_f, &ly.cqkv, &ly.bqkv_f);
            if (i + 1 < L) fold_ln(m, w1, b1, st.read(p + "layer_norm2.weight", D), st.read(p + "layer_norm2.bias", D), FF, D, &ly.w1_f, &ly.c1, &ly.b1_f);
        }
        {
            std::vector<float> w2 = st.read(p + "mlp.fc2.weight", (int64_t)D * FF), b2 = st.read(p + "mlp.fc2.bias", D);
            if (m->ln_center) center_writer(w2, b2, D, FF);
            ly.w2 = upload_mat(m, w2);
            ly.b2 = upload_f32(m, b2);
        }
    }
}


hipStream_t own_stream(mi_clip* m) {
    if (!m->stream) HIP_CHECK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    return m->stream;
}

void ensure_workspace(mi_clip* m, size_t n) {
    if (n <= m->cap) return;
    own_stream(m);
    m->order.sync();  // enqueued forwards still use the buffers freed below
    for (void* p : m->ws) HIP_CHECK(hipFree(p));
    m->ws.clear();
    m->cap = 0;
    const size_t es = esize(m), Mp = pad256(n * m->S), Pp = pad256(n * (m->S - 1));
    auto bytes = [&](size_t b) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, b));
        HIP_CHECK(hipMemsetAsync(p, 0, b, m->stream));
        m->ws.push_back(p);
        return p;
    };
    const size_t px = (size_t)m->image * m->image * 3;
    m->d_in = (float*)bytes(n * px * 4);
    m->d_in2 = (float*)bytes(n * px * 4);  // mi_clip_embed: upload of chunk i+1 under the forward of chunk i
    m->d_rgb = (uint8_t*)bytes(n * px);
    const int sets = (m->precision == MI_PRECISION_BF16) ? std::max(1, m->parts) : 1;
    for (int a = 0; a < sets; ++a) {
        const size_t na = a == 0 ? n : (n + 1) / 2;
        const size_t Ma = pad256(na * m->S), Pa = pad256(na * (m->S - 1));
        if (na == 0) continue;
        m->act[a].col = bytes(Pa * m->Kp * es);
        m->act[a].patch = (float*)bytes(Pa * m->D * 4);
        m->act[a].x = (float*)bytes(Ma * m->D * 4);
        m->act[a].y = bytes(Ma * m->D * es * (m->split_ln ? 2 : 1));
        m->act[a].qkv = bytes(Ma * qkv_pitch(m) * es);
        m->act[a].h = bytes(Ma * m->FF * es);
        m->act[a].delta = (bf16_t*)bytes(Ma * m->D * 2);
        m->act[a].delta2 = (bf16_t*)bytes(Ma * m->D * 2);
        const size_t Ca = pad256(na);
        m->act[a].c_ctx = bytes(Ca * m->D * es);
        m->act[a].c_y = bytes(Ca * m->D * es * (m->split_ln ? 2 : 1));
        m->act[a].c_h = bytes(Ca * m->FF * es);
        m->act[a].c_x = (float*)bytes(Ca * m->D * 4);
        m->act[a].c_d1 = (bf16_t*)bytes(Ca * m->D * 2);
        m->act[a].c_d2 = (bf16_t*)bytes(Ca * m->D * 2);
        if (m->fold_ready) {
            m->act[a].part = (float*)bytes(Ma * (size_t)(m->D / 32) * 8);
            m->act[a].stats = (float*)bytes(Ma * 8);
            m->act[a].c_stats = (float*)bytes(Ca * 8);
        }
    }
    (void)Mp; (void)Pp;
    m->d_out = (float*)bytes(n * m->E * 4);
    m->d_out2 = (float*)bytes(n * m->E * 4);
    HIP_CHECK(hipStreamSynchronize(m->stream));
    m->cap = n;
}

// ---- launches --------------------------------------------------------------------
template <int EPI>
void gemm_p(int precision, const void* X, const void* W, const float* bias, void* out, size_t Mrows, int N, int K,
            int ldo, hipStream_t s) {
    const size_t Mp = (Mrows + 127) / 128 * 128;
    const unsigned blocks = (unsigned)((Mp / 128) * (N / 128));
    if (N % 128 != 0) fail(MI_ERR_UNSUPPORTED, "GEMM N=%d is not a multiple of 128", N);
    if (precision == MI_PRECISION_F32) {
        if (K % 16 != 0) fail(MI_ERR_UNSUPPORTED, "GEMM K=%d is not a multiple of 16", K);
        if (Mp == 128)  // a handful of rows (text queries): 4x more, narrower workgroups
            hipLaunchKernelGGL((gemm_f32_n32_kernel<EPI>), dim3((unsigned)(N / 32)), dim3(256), 0, s, (const float*)X, (const float*)W,
                               bias, (float*)out, N, K, ldo);
        else
            hipLaunchKernelGGL((gemm_f32_kernel<EPI>), dim3(blocks), dim3(256), 0, s, (const float*)X, (const float*)W, bias,
                               (float*)out, N, K, ldo);
    } else {
        if (K % 64 != 0) fail(MI_ERR_UNSUPPORTED, "GEMM K=%d is not a multiple of 64", K);
        static DevOnce once;  // one per EPI instantiation
        if (EPI == EPI_STORE_F32 || EPI == EPI_BIAS_RESID) {
            auto kern = gemm_bf16_kernel<EPI, float>;
            allow_lds_once(once, kern, 65536);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, s, (const bf16_t*)X, (const bf16_t*)W, bias, out, N, K, ldo);
        } else {
            auto kern = gemm_bf16_kernel<EPI, bf16_t>;
            allow_lds_once(once, kern, 65536);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, s, (const bf16_t*)X, (const bf16_t*)W, bias, out, N, K, ldo);
        }
    }
    HIP_CHECK(hipGetLastError());
}

// the persistent 256x256 kernel (gemm_bf16_pp_kernel): shapes it takes, and its launch
bool pp_shape_ok(size_t Mp, int N, int K, int ldo) {
    return N % 256 == 0 && K % 64 == 0 && K >= 128 && Mp * (size_t)K * 2 < (1ull << 32) && Mp * (size_t)ldo * 2 < (1ull << 32) &&
           (size_t)N * K * 2 < (1ull << 32);
}
template <int EPI>
void launch_pp(mi_clip* m, const void* X, const void* W, const float* bias, void* out, size_t Mp, int N, int K, int ldo,
               const PpFold& fold, hipStream_t s) {
    constexpr bool LNF = EPI == EPI_LNF || EPI == EPI_LNF_QGELU;
    constexpr int LDS = 131072 + 18432 + 8 * (LNF ? 1536 : 256);
    const int n_tiles = (int)((Mp / 256) * (N / 256));
    const int grid = std::min(n_tiles * 4, m->n_cu);
    // a short last round (<= a quarter of the CUs busy) is cut into quadrant tasks
    const int left = n_tiles % grid;
    const int n_full = (m->split_tail && left > 0 && left * 4 <= grid) ? n_tiles - left : n_tiles;
    const int nt = N / 256;
    auto fits = [&](int np) { return np > 0 && nt > np && nt % np == 0; };
    const int order = fits(m->gemm_order) ? m->gemm_order : (m->gemm_order > 0 && fits(4)) ? 4 : 0;
    // option "store_nt" (default 1): the plain output stores with the nt cache policy; EPI_RESID24 has none of them
    if (EPI == EPI_RESID24 || m->store_nt) {
        auto kern = gemm_bf16_pp_kernel<EPI, bf16_t, true>;
        static DevOnce once;
        allow_lds_once(once, kern, LDS);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, (const bf16_t*)X, (const bf16_t*)W, bias, out, (int)Mp, N, K, ldo,
                           n_tiles, n_full, order, fold);
    } else {
        auto kern = gemm_bf16_pp_kernel<EPI, bf16_t, (EPI == EPI_RESID24)>;
        static DevOnce once;
        allow_lds_once(once, kern, LDS);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), LDS, s, (const bf16_t*)X, (const bf16_t*)W, bias, out, (int)Mp, N, K, ldo,
                           n_tiles, n_full, order, fold);
    }
    HIP_CHECK(hipGetLastError());
}

// bf16 GEMM with bf16 output: the persistent 256x256 kernel when the shape allows, else 128x128
// col_stride != 64: head-major output planes (PpFold::col_stride, ldo = 64) — the persistent kernel only
template <int EPI>
void gemm(mi_clip* m, const void* X, const void* W, const float* bias, void* out, size_t Mrows, int N, int K, int ldo,
          hipStream_t s, uint32_t col_stride = 64) {
    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_QGELU) {
        const size_t Mp = pad256(Mrows);
        if (m->precision == MI_PRECISION_BF16 && pp_shape_ok(Mp, N, K, col_stride == 64 ? ldo : N)) {
            PpFold f;
            f.col_stride = col_stride;
            launch_pp<EPI>(m, X, W, bias, out, Mp, N, K, ldo, f, s);
            return;
        }
    }
    if (col_stride != 64) fail(MI_ERR_INVALID, "head-major output needs the persistent GEMM");
    gemm_p<EPI>(m->precision, X, W, bias, out, Mrows, N, K, ldo, s);
}

// dispatch on the row width D = 64*VEC*NT
#define MI_LN_DISPATCH(D, CALL)                                                              \
    switch (D) {                                                                             \
        case 128: { constexpr int VEC = 2, NT = 1; CALL; } break;                            \
        case 256: { constexpr int VEC = 4, NT = 1; CALL; } break;                            \
        case 384: { constexpr int VEC = 2, NT = 3; CALL; } break;                            \
        case 512: { constexpr int VEC = 4, NT = 2; CALL; } break;                            \
        case 768: { constexpr int VEC = 4, NT = 3; CALL; } break;                            \
        case 1024: { constexpr int VEC = 4, NT = 4; CALL; } break;                           \
        case 1280: { constexpr int VEC = 4, NT = 5; CALL; } break;                           \
        case 1536: { constexpr int VEC = 4, NT = 6; CALL; } break;                           \
        case 1664: { constexpr int VEC = 2, NT = 13; CALL; } break;                          \
        default: fail(MI_ERR_UNSUPPORTED, "hidden size %d has no LayerNorm instantiation", D); \
    }

// y pitch: D, or 2D with the lo halves behind the hi halves (split_ln)
// x_lo: byte offset of the lo plane when x is the 24-bit residual stream (0: plain fp32 rows)
void layer_norm(mi_clip* m, float* x, const bf16_t* d1, const bf16_t* d2, bool write_back, void* y, const float* w,
                const float* b, size_t rows, hipStream_t s, size_t x_lo = 0) {
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    const int split = m->split_ln ? 1 : 0, y_ld = m->D * (1 + split);
    if (m->precision == MI_PRECISION_F32) {
        MI_LN_DISPATCH(m->D, hipLaunchKernelGGL((ln_kernel<float, VEC, NT, true>), dim3(blocks), dim3(256), 0, s, x, d1, d2, (float*)y, w, b, (int)rows, m->eps, y_ld, 0));
    } else if (write_back) {
        MI_LN_DISPATCH(m->D, hipLaunchKernelGGL((ln_kernel<bf16_t, VEC, NT, true>), dim3(blocks), dim3(256), 0, s, x, d1, d2, (bf16_t*)y, w, b, (int)rows, m->eps, y_ld, split, m->ln_nt, x_lo));
    } else {
        MI_LN_DISPATCH(m->D, hipLaunchKernelGGL((ln_kernel<bf16_t, VEC, NT, false>), dim3(blocks), dim3(256), 0, s, x, d1, d2, (bf16_t*)y, w, b, (int)rows, m->eps, y_ld, split, 0, x_lo));
    }
    HIP_CHECK(hipGetLastError());
}

// first_tile_only: only the leading query tile / block of every (image, head) -- it holds the CLS row
// hm_rows > 0: qkv lies head-major, [3][H][hm_rows][64] (attn32 only; the q/k/v GEMM wrote it so, option "qkv_layout")
void attention(mi_clip* m, const void* qkv, void* ctx, size_t n, hipStream_t s, bool first_tile_only = false, int ld_qkv = 0,
               size_t hm_rows = 0) {
    if (ld_qkv == 0) ld_qkv = 3 * m->D;
    uint32_t head_stride = 64, sel_stride = (uint32_t)m->D;
    if (hm_rows) {
        if (!attn32_applies(m)) fail(MI_ERR_INVALID, "head-major q|k|v is the attn32 kernel's layout");
        ld_qkv = 64;
        head_stride = (uint32_t)(hm_rows * 64);
        sel_stride = (uint32_t)((size_t)m->H * hm_rows * 64);
    }
    if (m->precision == MI_PRECISION_F32) {
        if (m->S <= 272 && m->attn_f32_mfma) {   // on the matrix pipe (exact-f32 MFMA), one workgroup per (image, head)
#define MI_ATTNF(SP)                                                                                                    \
    {                                                                                                                   \
        static DevOnce once;                                                                                            \
        allow_lds_once(once, attn_f32_mfma_kernel<SP>, attnf_lds_bytes(SP));                                            \
        hipLaunchKernelGGL((attn_f32_mfma_kernel<SP>), dim3((unsigned)(n * m->H)), dim3(512), attnf_lds_bytes(SP), s, (const float*)qkv, (float*)ctx, m->S, m->D, m->H, m->text ? 1 : 0, first_tile_only ? 1 : 0); \
    }
            if (m->S <= 80) MI_ATTNF(80)          // the text tower's 77 positions; ViT-B/32
            else if (m->S <= 208) MI_ATTNF(208)   // ViT-B/16 @224
            else MI_ATTNF(272)                    // ViT-L/14, ViT-H/14 @224
#undef MI_ATTNF
        } else {
            const int qb = first_tile_only ? 1 : (m->S + 63) / 64;
            const unsigned blocks = (unsigned)(n * m->H * qb);
            hipLaunchKernelGGL(attn_f32_kernel, dim3(blocks), dim3(64), 0, s, (const float*)qkv, (float*)ctx, m->S, m->D, m->H, m->text ? 1 : 0, first_tile_only ? 1 : 0);
        }
    } else if (attn32_applies(m)) {
#define MI_ATTN32_L(SP, SC, PS, AUX)                                                                                   \
    {                                                                                                                  \
        static DevOnce once;                                                                                           \
        allow_lds_once(once, attn32_bf16_kernel<SP, SC, PS, AUX>, LDS);                                                \
        hipLaunchKernelGGL((attn32_bf16_kernel<SP, SC, PS, AUX>), dim3(grid), dim3(512), LDS, s, (const bf16_t*)qkv, (bf16_t*)ctx, m->S, m->D, m->H, pairs, first_tile_only ? 1 : 0, m->attn_shift ? 1 : 0, m->attn_order, ld_qkv, m->D, head_stride, sel_stride); \
    }
#define MI_ATTN32(SP, SC)                                                                                              \
    {                                                                                                                  \
        constexpr int LDS = attn32_lds_bytes(SP);                                                                      \
        const int pairs = (int)(n * m->H), grid = std::min(pairs, m->n_cu);                                            \
        if (m->q_prescaled) {                                                                                          \
            if (m->attn_nt) MI_ATTN32_L(SP, SC, true, 2) else MI_ATTN32_L(SP, SC, true, 0)                             \
        } else {                                                                                                       \
            if (m->attn_nt) MI_ATTN32_L(SP, SC, false, 2) else MI_ATTN32_L(SP, SC, false, 0)                           \
        }                                                                                                              \
    }
        if (m->S == 257) MI_ATTN32(288, 257)       // ViT-L/14, ViT-H/14 @224
        else if (m->S == 197) MI_ATTN32(224, 197)  // ViT-B/16 @224
        else if (m->S <= 128) MI_ATTN32(128, 0)
        else if (m->S <= 224) MI_ATTN32(224, 0)
        else MI_ATTN32(288, 0)
#undef MI_ATTN32
#undef MI_ATTN32_L
    } else {
        if (m->q_prescaled) fail(MI_ERR_INVALID, "this handle's q weights carry the attn32 scale: attn_ver is fixed at load");
        const unsigned blocks = (unsigned)(n * m->H);
        const int sp = (m->S + 31) / 32 * 32;
#define MI_ATTN(SP, SC)                                                                                              \
    {                                                                                                                \
        static DevOnce once;                                                                                         \
        allow_lds_once(once, attn_bf16_kernel<SP, SC>, SP * 256);                                                    \
        hipLaunchKernelGGL((attn_bf16_kernel<SP, SC>), dim3(blocks), dim3(256), SP * 256, s, (const bf16_t*)qkv, (bf16_t*)ctx, m->S, m->D, m->H, first_tile_only ? 1 : 0, m->text ? 1 : 0); \
    }
        if (m->S == 257) MI_ATTN(288, 257)       // ViT-L/14, ViT-H/14 @224
        else if (m->S == 197) MI_ATTN(224, 197)  // ViT-B/16 @224
        else if (m->S == 50) MI_ATTN(64, 50)     // ViT-B/32 @224
        else if (sp <= 32) MI_ATTN(32, 0)
        else if (sp <= 64) MI_ATTN(64, 0)
        else if (sp <= 96) MI_ATTN(96, 0)        // the text tower's 77 positions
        else if (sp <= 224) MI_ATTN(224, 0)
        else if (sp <= 288) MI_ATTN(288, 0)
        else fail(MI_ERR_UNSUPPORTED, "bf16 attention is built for up to 288 tokens (got %d)", m->S);
#undef MI_ATTN
    }
    HIP_CHECK(hipGetLastError());
}

// y pitch / x planes as in layer_norm; x_bias: X24B_BIAS when the planes were written by the LayerNorm-free layers
void layer_norm_x(mi_clip* m, float* x, const bf16_t* d1, const bf16_t* d2, bool write_back, void* y, const float* w, const float* b,
                  size_t rows, hipStream_t s, size_t x_lo, uint32_t x_bias) {
    if (!x_bias) { layer_norm(m, x, d1, d2, write_back, y, w, b, rows, s, x_lo); return; }
    const unsigned blocks = (unsigned)((rows + 3) / 4);
    if (write_back) {
        MI_LN_DISPATCH(m->D, hipLaunchKernelGGL((ln_kernel<bf16_t, VEC, NT, true>), dim3(blocks), dim3(256), 0, s, x, d1, d2, (bf16_t*)y, w, b, (int)rows, m->eps, m->D, 0, 0, x_lo, x_bias));
    } else {
        MI_LN_DISPATCH(m->D, hipLaunchKernelGGL((ln_kernel<bf16_t, VEC, NT, false>), dim3(blocks), dim3(256), 0, s, x, d1, d2, (bf16_t*)y, w, b, (int)rows, m->eps, m->D, 0, 0, x_lo, x_bias));
    }
    HIP_CHECK(hipGetLastError());
}

// can this forward run without LayerNorm kernels in the layer loop (option "ln_fold")?
bool fold_applies(const mi_clip* m, size_t rows_max) {
    const size_t Mp = pad256(rows_max);
    const int ldq = (int)qkv_pitch(m);   // the q/k/v launch and the last layer's K | V launch write rows of the PADDED pitch
    return m->ln_fold && m->fold_ready && !m->text && pp_shape_ok(Mp, 3 * m->D, m->D, ldq) && pp_shape_ok(Mp, 2 * m->D, m->D, ldq) &&
           pp_shape_ok(Mp, m->D, m->D, m->D) &&
           pp_shape_ok(Mp, m->FF, m->D, m->FF) && pp_shape_ok(Mp, m->D, m->FF, m->D) && Mp * (size_t)(m->D / 32) * 8 < (1ull << 32);
}

// the whole tower on n <= cap device-resident images.
// bf16, n >= 32: the chunk is cut into two halves that run as two independent streams with their
// own activation sets, launches interleaved layer by layer.  The persistent GEMM of one half
// owns every CU's LDS while it runs, but the other half's small kernels share
// the CUs with it, and whichever kernel of the other half is next fills the CUs that fall idle
// in a GEMM's last, partial round of tiles (257 row-tiles never divide by 256 CUs).
//
// Two forms of the bf16 layer loop:
//  * LayerNorm kernels (default): out_proj / fc2 store bf16 `delta` / `delta2` (pure, asynchronous stores from the
//    persistent GEMM).  LN2 normalises x + delta without writing x; the next LN1 forms (x + delta) + delta2 — the same
//    order — writes it back and normalises it.
//  * LayerNorm-free (option "ln_fold", layers 0 .. L-2): the residual stream lives as two planes whose hi plane is bf16(x).
//    q/k/v and fc1 read that plane as it lies, with gamma folded into their weights, and finish the LayerNorm in their
//    epilogue from per-row {rstd, -mean rstd} (EPI_LNF); out_proj / fc2 add their output to the planes in place and emit
//    per-row partial sums (EPI_RESID24), which ln_stats_kernel turns into the next {rstd, -mean rstd}.  The last layer
//    folds its LN1 too (K | V of every token and the CLS rows' queries by two EPI_LNF launches) and keeps LayerNorm kernels
//    for what runs on the CLS rows only (LN2 on n rows, the head).
// fp32 path: the residual add is the GEMM epilogue (x += acc + bias, fp32 read-modify-write).
// front != nullptr: the patch gather and the patch GEMM (the only readers of d_img and the only writers of col / patch) are
// enqueued on that stream instead of the parts' own — a caller that has just uploaded d_img on it (mi_pipeline_ingest: the
// copy stream) lets them run under the PREVIOUS forward's layers, which never touch those buffers; events order them against
// the embed_ln kernels on either side.  Same kernels, same arguments: same bits.
void forward(mi_clip* m, const float* d_img, size_t n, float* d_out, hipStream_t s0, hipStream_t front = nullptr) {
    const int D = m->D, S = m->S, FF = m->FF;
    const size_t px = (size_t)m->image * m->image * 3;
    const bool deferred = m->precision == MI_PRECISION_BF16;
    const int Kln = m->split_ln ? 2 * D : D;  // K of the GEMMs fed by a LayerNorm (q/k/v, fc1): hi | lo halves when split
    const int parts = (deferred && m->parts > 1 && n >= 32) ? m->parts : 1;
    struct Part { size_t n, M, P; const float* img; float* out; hipStream_t s; mi_clip::Act* a; const bf16_t *p1, *p2; size_t xlo; } pt[4];
    const bool fold = deferred && fold_applies(m, (n / parts + 1) * (size_t)S);
    // the residual stream as 24-bit floats in two planes (option "x24"; bf16 tower, D a multiple of 256, no hi + lo LayerNorm outputs)
    const bool x24 = fold || (deferred && m->x24 && !m->split_ln && D % 256 == 0);
    const uint32_t x_bias = fold ? X24B_BIAS : 0u;
    for (size_t p = 0, first = 0; p < (size_t)parts; ++p) {
        const size_t np = n / parts + (p < n % parts ? 1 : 0);
        hipStream_t st = s0;
        if (p > 0) {
            if (!m->aux[p - 1]) HIP_CHECK(hipStreamCreateWithFlags(&m->aux[p - 1], hipStreamNonBlocking));
            st = m->aux[p - 1];
        }
        pt[p] = {np, np * S, np * (S - 1), d_img + first * px, d_out + first * m->E, st, &m->act[p], nullptr, nullptr,
                 x24 ? pad256(np * S) * (size_t)D * 2 : (size_t)0};   // the lo plane lies behind a hi plane of the part's padded rows
        first += np;
    }
    if (parts > 1) {
        HIP_CHECK(hipEventRecord(m->ev_fork, s0));
        for (int p = 1; p < parts; ++p) HIP_CHECK(hipStreamWaitEvent(pt[p].s, m->ev_fork, 0));
    }
    for (int p = 0; p < parts; ++p) {
        Part& q = pt[p];
        hipStream_t es_ = q.s;                 // embed_ln and everything behind it
        hipStream_t fs = front ? front : q.s;  // the front
        if (!m->ev_embed[p]) {
            HIP_CHECK(hipEventCreateWithFlags(&m->ev_embed[p], hipEventDisableTiming));
            HIP_CHECK(hipEventCreateWithFlags(&m->ev_front[p], hipEventDisableTiming));
        }
        if (front && m->ev_embed_set[p]) HIP_CHECK(hipStreamWaitEvent(front, m->ev_embed[p], 0));
        // patch embedding: gather -> GEMM [P,Kp] x [D,Kp]^T -> f32
        const size_t total = q.P * 3 * (size_t)m->patch;  // one thread per patch-row segment
        const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 65535 * 4);
        if (m->precision == MI_PRECISION_F32)
            hipLaunchKernelGGL((im2col_kernel<float>), dim3(blocks), dim3(256), 0, fs, q.img, (float*)q.a->col, (int)q.n, m->grid, m->patch, m->image, m->Kp);
        else if (m->im2col_rows && m->image % 4 == 0 && (3 * m->patch * m->patch) % 4 == 0 && m->Kp % 4 == 0 && (size_t)3 * m->patch * m->image * 4 <= 64 * 1024)
            hipLaunchKernelGGL(im2col_rows_kernel, dim3((unsigned)(q.n * m->grid)), dim3(256), (size_t)3 * m->patch * m->image * 4, fs, q.img,
                               (bf16_t*)q.a->col, m->grid, m->patch, m->image, m->Kp);
        else
            hipLaunchKernelGGL((im2col_kernel<bf16_t>), dim3(blocks), dim3(256), 0, fs, q.img, (bf16_t*)q.a->col, (int)q.n, m->grid, m->patch, m->image, m->Kp);
        HIP_CHECK(hipGetLastError());
        gemm<EPI_STORE_F32>(m, q.a->col, m->wpatch, nullptr, q.a->patch, q.P, D, m->Kp, D, fs);
        if (front) {
            HIP_CHECK(hipEventRecord(m->ev_front[p], front));
            HIP_CHECK(hipStreamWaitEvent(es_, m->ev_front[p], 0));
        }
        fs = es_;
        const size_t rows_launch = fold ? pad256(q.M) : q.M;   // ln_fold: the padding rows get a defined (zero) residual too
        const unsigned lb = (unsigned)((rows_launch + 3) / 4);
        MI_LN_DISPATCH(D, hipLaunchKernelGGL((embed_ln_kernel<VEC, NT>), dim3(lb), dim3(256), 0, fs, q.a->patch, m->cls, m->pos, q.a->x, m->pre_w, m->pre_b, (int)q.M, S, m->eps, q.xlo,
                                             fold ? q.a->stats : (float*)nullptr, (int)rows_launch, fold ? m->d_fold_offset_rows : nullptr, m->ln_center ? 1 : 0));
        if (fold) m->fold_rows_checked += q.M;
        HIP_CHECK(hipEventRecord(m->ev_embed[p], fs));   // `patch` has been read: a later forward's front may overwrite it
        m->ev_embed_set[p] = true;
        HIP_CHECK(hipGetLastError());
    }
    // The pooled output is the CLS row (modeling_clip.py:641-651), so behind the LAST layer's attention
    // only that row of every image is live: its context row is gathered and out_proj, LN2, the MLP and
    // the head run on n rows instead of n*S (same arithmetic per row, so the same bits; the reference's
    // graph computes the other rows and discards them).  Option "full_last" keeps the full last layer.
    const bool full_last = m->full_last;
    const int nb = D / 32;
    const int ldq = (int)qkv_pitch(m);   // row pitch of qkv (padded where attn32 runs)
    // head-major q|k|v (option "qkv_layout"): per part, planes of its padded row count
    const bool hm = qkv_head_major(m) && pp_shape_ok(pad256((n / parts + 1) * (size_t)S), 3 * D, D, 3 * D);
    auto hm_rows = [&](const Part& q) { return hm ? pad256(q.M) : (size_t)0; };
    auto qkv_cs = [&](const Part& q) { return hm ? (uint32_t)(pad256(q.M) * 64) : 64u; };
    const int ldq_o = hm ? 64 : ldq;   // the q/k/v GEMM's output row pitch
    auto ln_stats = [&](Part& q) {   // the partial sums of the GEMM just enqueued -> {rstd, -mean rstd} per (padded) row
        const size_t Mp = pad256(q.M);
        hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)(Mp / 16)), dim3(256), 0, q.s, q.a->part, q.a->stats, (int)Mp, nb, 1.0f / (float)D, m->eps,
                           (int)q.M, m->d_fold_offset_rows);
        HIP_CHECK(hipGetLastError());
        m->fold_rows_checked += q.M;
    };
    for (size_t li = 0; li < m->layers.size(); ++li) {
        const Layer& ly = m->layers[li];
        const bool last = !full_last && li + 1 == m->layers.size();
        const bool lnf = fold && li + 1 < m->layers.size();   // this layer runs without LayerNorm kernels
        // ---- LN1, q/k/v, attention of every part; then the rest of every part
        for (int p = 0; p < parts; ++p) {
            Part& q = pt[p];
            if (fold && last) continue;   // LN1 folded into the K / V and CLS-query GEMMs below
            if (fold) {   // every layer's LN1 is folded, the last one's too (its LN2 is not: it runs on the CLS rows)
                PpFold f;
                f.cvec = ly.cqkv; f.stats = q.a->stats; f.col_stride = qkv_cs(q);
                launch_pp<EPI_LNF>(m, q.a->x, ly.wqkv_f, ly.bqkv_f, q.a->qkv, pad256(q.M), 3 * D, D, ldq_o, f, q.s);
                attention(m, q.a->qkv, q.a->y, q.n, q.s, false, ldq, hm_rows(q));
                continue;
            }
            layer_norm_x(m, q.a->x, q.p1, q.p2, true, q.a->y, ly.ln1w, ly.ln1b, q.M, q.s, q.xlo, x_bias);
            if (last) continue;
            gemm<EPI_BIAS>(m, q.a->y, ly.wqkv, ly.bqkv, q.a->qkv, q.M, 3 * D, Kln, ldq_o, q.s, qkv_cs(q));
            attention(m, q.a->qkv, q.a->y, q.n, q.s, false, ldq, hm_rows(q));
        }
        for (int p = 0; p < parts; ++p) {
            Part& q = pt[p];
            hipStream_t s = q.s;
            if (lnf) {
                PpFold f;
                f.xlo = (uint8_t*)q.a->x + q.xlo; f.part = q.a->part;
                launch_pp<EPI_RESID24>(m, q.a->y, ly.wo, ly.bo, q.a->x, pad256(q.M), D, D, D, f, s);
                ln_stats(q);
                PpFold f1;
                f1.cvec = ly.c1; f1.stats = q.a->stats;
                launch_pp<EPI_LNF_QGELU>(m, q.a->x, ly.w1_f, ly.b1_f, q.a->h, pad256(q.M), FF, D, FF, f1, s);
                launch_pp<EPI_RESID24>(m, q.a->h, ly.w2, ly.b2, q.a->x, pad256(q.M), D, FF, D, f, s);
                ln_stats(q);
                continue;
            }
            if (last) {
                // keys and values of every token, queries of the CLS rows only (the other rows of the
                // leading query tile keep whatever the buffer held: their context rows are never read)
                const size_t es = esize(m);
                const unsigned gb = (unsigned)std::min<size_t>((q.n * (size_t)D / 4 + 255) / 256, 4096);
                // the K | V columns start D elements into a token row, or at plane H of the head-major form
                char* kv_out = (char*)q.a->qkv + (hm ? (size_t)m->H * pad256(q.M) * 64 : (size_t)D) * es;
                if (fold) {
                    // the same EPI_LNF arithmetic as the full layer's q/k/v launch, so the same bits: K | V columns of every
                    // row from the hi plane, and the CLS rows' hi-plane rows + statistics gathered for the query columns
                    PpFold f;
                    f.cvec = ly.cqkv + D; f.stats = q.a->stats; f.col_stride = qkv_cs(q);
                    launch_pp<EPI_LNF>(m, q.a->x, (const char*)ly.wqkv_f + (size_t)D * D * 2, ly.bqkv_f + D, kv_out, pad256(q.M), 2 * D, D, ldq_o, f, s);
                    hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), dim3(gb), dim3(256), 0, s, (const bf16_t*)q.a->x, (bf16_t*)q.a->c_y, (int)q.n, (size_t)S, D);
                    hipLaunchKernelGGL(gather_stats_kernel, dim3((unsigned)((q.n + 255) / 256)), dim3(256), 0, s, q.a->stats, q.a->c_stats, (int)q.n, (size_t)S);
                    PpFold fq;
                    fq.cvec = ly.cqkv; fq.stats = q.a->c_stats;
                    launch_pp<EPI_LNF>(m, q.a->c_y, ly.wqkv_f, ly.bqkv_f, q.a->c_ctx, pad256(q.n), D, D, D, fq, s);
                } else {
                gemm<EPI_BIAS>(m, q.a->y, (const char*)ly.wqkv + (size_t)D * Kln * es, ly.bqkv + D, kv_out, q.M, 2 * D, Kln, ldq_o, s, qkv_cs(q));
                if (deferred) hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), dim3(gb), dim3(256), 0, s, (const bf16_t*)q.a->y, (bf16_t*)q.a->c_y, (int)q.n, (size_t)S, Kln);
                else hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(gb), dim3(256), 0, s, (const float*)q.a->y, (float*)q.a->c_y, (int)q.n, (size_t)S, D);
                gemm<EPI_BIAS>(m, q.a->c_y, ly.wqkv, ly.bqkv, q.a->c_ctx, q.n, D, Kln, D, s);
                }
                if (deferred) hipLaunchKernelGGL((scatter_rows_kernel<bf16_t>), dim3(gb), dim3(256), 0, s, (const bf16_t*)q.a->c_ctx, (bf16_t*)q.a->qkv, (int)q.n, (size_t)S, D, (size_t)ldq_o, (size_t)qkv_cs(q));
                else hipLaunchKernelGGL((scatter_rows_kernel<float>), dim3(gb), dim3(256), 0, s, (const float*)q.a->c_ctx, (float*)q.a->qkv, (int)q.n, (size_t)S, D, (size_t)ldq, (size_t)64);
                HIP_CHECK(hipGetLastError());
                attention(m, q.a->qkv, q.a->y, q.n, s, true, ldq, hm_rows(q));
                const unsigned gb2 = gb;
                if (deferred) hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), dim3(gb2), dim3(256), 0, s, (const bf16_t*)q.a->y, (bf16_t*)q.a->c_ctx, (int)q.n, (size_t)S, D);
                else hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(gb2), dim3(256), 0, s, (const float*)q.a->y, (float*)q.a->c_ctx, (int)q.n, (size_t)S, D);
                if (q.xlo) hipLaunchKernelGGL(gather_rows_x24_kernel, dim3(gb2), dim3(256), 0, s, (const uint16_t*)q.a->x, (const uint8_t*)q.a->x + q.xlo, q.a->c_x, (int)q.n, (size_t)S, D, x_bias);
                else hipLaunchKernelGGL((gather_rows_kernel<float>), dim3(gb2), dim3(256), 0, s, (const float*)q.a->x, q.a->c_x, (int)q.n, (size_t)S, D);
                HIP_CHECK(hipGetLastError());
                if (deferred) {
                    gemm<EPI_BIAS>(m, q.a->c_ctx, ly.wo, ly.bo, q.a->c_d1, q.n, D, D, D, s);
                    layer_norm(m, q.a->c_x, q.a->c_d1, nullptr, false, q.a->c_y, ly.ln2w, ly.ln2b, q.n, s);
                    gemm<EPI_BIAS_QGELU>(m, q.a->c_y, ly.w1, ly.b1, q.a->c_h, q.n, FF, Kln, FF, s);
                    gemm<EPI_BIAS>(m, q.a->c_h, ly.w2, ly.b2, q.a->c_d2, q.n, D, FF, D, s);
                } else {
                    gemm<EPI_BIAS_RESID>(m, q.a->c_ctx, ly.wo, ly.bo, q.a->c_x, q.n, D, D, D, s);
                    layer_norm(m, q.a->c_x, nullptr, nullptr, true, q.a->c_y, ly.ln2w, ly.ln2b, q.n, s);
                    gemm<EPI_BIAS_QGELU>(m, q.a->c_y, ly.w1, ly.b1, q.a->c_h, q.n, FF, Kln, FF, s);
                    gemm<EPI_BIAS_RESID>(m, q.a->c_h, ly.w2, ly.b2, q.a->c_x, q.n, D, FF, D, s);
                }
                MI_LN_DISPATCH(D, hipLaunchKernelGGL((head_kernel<VEC, NT>), dim3((unsigned)((q.n + 7) / 8), 16), dim3(256), 0, s, q.a->c_x, deferred ? q.a->c_d1 : (const bf16_t*)nullptr, deferred ? q.a->c_d2 : (const bf16_t*)nullptr, m->post_w, m->post_b, m->proj, q.out, (int)q.n, 1, m->E, m->eps, (const int*)nullptr));
                HIP_CHECK(hipGetLastError());
                continue;
            }
            if (deferred) gemm<EPI_BIAS>(m, q.a->y, ly.wo, ly.bo, q.a->delta, q.M, D, D, D, s);
            else gemm<EPI_BIAS_RESID>(m, q.a->y, ly.wo, ly.bo, q.a->x, q.M, D, D, D, s);
            if (deferred) layer_norm_x(m, q.a->x, q.a->delta, nullptr, false, q.a->y, ly.ln2w, ly.ln2b, q.M, s, q.xlo, x_bias);
            else layer_norm(m, q.a->x, nullptr, nullptr, true, q.a->y, ly.ln2w, ly.ln2b, q.M, s);
            gemm<EPI_BIAS_QGELU>(m, q.a->y, ly.w1, ly.b1, q.a->h, q.M, FF, Kln, FF, s);
            if (deferred) { gemm<EPI_BIAS>(m, q.a->h, ly.w2, ly.b2, q.a->delta2, q.M, D, FF, D, s); q.p1 = q.a->delta; q.p2 = q.a->delta2; }
            else gemm<EPI_BIAS_RESID>(m, q.a->h, ly.w2, ly.b2, q.a->x, q.M, D, FF, D, s);
        }
        if (last) break;
    }
    for (int p = 0; p < parts && full_last; ++p) {
        Part& q = pt[p];
        // fp32 in both precisions: the embedding that goes to the table is not rounded to bf16 anywhere here
        MI_LN_DISPATCH(D, hipLaunchKernelGGL((head_kernel<VEC, NT>), dim3((unsigned)((q.n + 7) / 8), 16), dim3(256), 0, q.s, q.a->x, q.p1, q.p2, m->post_w, m->post_b, m->proj, q.out, (int)q.n, S, m->E, m->eps, (const int*)nullptr, q.xlo, x_bias));
        HIP_CHECK(hipGetLastError());
    }
    for (int p = 1; p < parts; ++p) {
        HIP_CHECK(hipEventRecord(m->ev_join[p - 1], pt[p].s));
        HIP_CHECK(hipStreamWaitEvent(s0, m->ev_join[p - 1], 0));
    }
}

void ensure_copy_stream(mi_clip* m) {
    if (m->copy_stream) return;
    HIP_CHECK(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
        HIP_CHECK(hipEventCreateWithFlags(&m->ev_up[b], hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&m->ev_used[b], hipEventDisableTiming));
    }
}

// ---- text tower: workspace and forward (fp32) -------------------------------------------
void drop_text_graph(mi_clip* m) {
    if (m->text_graph_exec) (void)hipGraphExecDestroy(m->text_graph_exec);
    if (m->text_graph) (void)hipGraphDestroy(m->text_graph);
    m->text_graph_exec = nullptr; m->text_graph = nullptr;
    m->text_graph_state = 0;
}

void ensure_text_workspace(mi_clip* m, size_t n) {
    if (n <= m->text_cap) return;
    own_stream(m);
    m->order.sync();
    drop_text_graph(m);  // it holds the addresses of the buffers freed here
    for (void* p : m->ws) HIP_CHECK(hipFree(p));
    m->ws.clear();
    m->text_cap = 0;
    const size_t Ma = pad256(n * m->S);
    auto bytes = [&](size_t b) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, b));
        HIP_CHECK(hipMemsetAsync(p, 0, b, m->stream));
        m->ws.push_back(p);
        return p;
    };
    const size_t es = esize(m);
    m->act[0].x = (float*)bytes(Ma * m->D * 4);
    m->act[0].y = bytes(Ma * m->D * es);
    m->act[0].qkv = bytes(Ma * 3 * m->D * es);
    m->act[0].h = bytes(Ma * m->FF * es);
    if (m->precision == MI_PRECISION_BF16) {
        m->act[0].delta = (bf16_t*)bytes(Ma * m->D * 2);
        m->act[0].delta2 = (bf16_t*)bytes(Ma * m->D * 2);
        m->act[0].patch = (float*)bytes((size_t)(m->FF / SKINNY_KC + 1) * SKINNY_ROWS * m->D * 4);  // fc2 partial slabs of forward_text_one
        m->act[0].col = bytes((size_t)m->H * SKINNY_ROWS * m->D * 4);                                // its per-head out_proj slabs (text_attn_out_kernel)
    }
    m->d_ids = (int*)bytes(Ma * sizeof(int));
    m->d_rows = (int*)bytes(n * sizeof(int));
    m->d_out = (float*)bytes(n * m->E * 4);
    HIP_CHECK(hipStreamSynchronize(m->stream));
    m->text_cap = n;
}

// ONE sequence (the request path: server/src/clip.rs:19-23 embeds one query per search) on the skinny GEMMs of
// vit_kernels.h: the same graph and the same bf16 rounding points as forward_text, except that fc2's output reaches the
// residual stream as fp32 partial sums instead of a bf16 delta.  Geometry of the CLIP-L text tower only (D = 768 = one K
// chunk, FF = 4 chunks, at most 80 positions); everything else takes forward_text.
bool text_one_applies(const mi_clip* m, size_t n) {
    return m->text_fast && m->precision == MI_PRECISION_BF16 && !m->split_ln && n == 1 && m->D == SKINNY_KC && m->FF == 4 * SKINNY_KC &&
           m->S <= SKINNY_ROWS && m->D % 16 == 0 && !m->layers.empty();
}
void forward_text_one(mi_clip* m, hipStream_t s) {
    const int D = m->D, S = m->S, FF = m->FF;
    mi_clip::Act& a = m->act[0];
    // the ids come from, and the embedding goes to, pinned staging: both copies are nodes of the captured graph
    HIP_CHECK(hipMemcpyAsync(m->d_ids, m->h_ids_pin, (size_t)S * sizeof(int32_t), hipMemcpyHostToDevice, s));
    bf16_t *y = (bf16_t*)a.y, *qkv = (bf16_t*)a.qkv, *h = (bf16_t*)a.h;
    float* slabs = a.patch;  // [4][SKINNY_ROWS][D] fp32
    const unsigned lb = (unsigned)((S + 3) / 4);
    constexpr int VEC = 4, NT = 3;  // D = 768
    const bf16_t* d1 = nullptr;
    const float* b2 = nullptr;
    const int* ids = m->d_ids;      // first layer: LN1 also assembles x = token + position embeddings
    int n_slabs = 0;
    // out_proj inside the attention launch (option "text_fuse", default): every head's contribution arrives as an fp32 slab
    // and LN2 lands their sum (+ bias) in the residual stream — one launch less per layer, no bf16 rounding of the delta
    const bool fuse = m->text_fuse && D % (16 * TAO_NT) == 0 && S <= SKINNY_ROWS;
    float* oslabs = (float*)a.col;  // [H][SKINNY_ROWS][D] fp32
    for (const Layer& ly : m->layers) {
        // LN1 also lands the previous layer's residual adds: x += out_proj delta + fc2 partial sums + fc2 bias
        hipLaunchKernelGGL((ln_slab_kernel<VEC, NT>), dim3(lb), dim3(256), 0, s, a.x, d1, slabs, n_slabs, b2, y, ly.ln1w, ly.ln1b, S, m->eps, 1,
                           ids, m->tok, m->pos);
        ids = nullptr;
        hipLaunchKernelGGL((skinny_gemm_kernel<SKINNY_BIAS>), dim3(3 * D / 16, 1), dim3(256), 0, s, y, D, (const bf16_t*)ly.wqkv, D, ly.bqkv, qkv, 3 * D);
        if (fuse) {
            constexpr int SP = 96;
            hipLaunchKernelGGL((text_attn_out_kernel<SP>), dim3(m->H, D / (16 * TAO_NT)), dim3(64 * TAO_WAVES), SP * 256, s, qkv, (const bf16_t*)ly.wo, oslabs, S, D, 1);
            hipLaunchKernelGGL((ln_slab_kernel<VEC, NT>), dim3(lb), dim3(256), 0, s, a.x, (const bf16_t*)nullptr, oslabs, m->H, ly.bo, y, ly.ln2w, ly.ln2b, S, m->eps, 1,
                               (const int*)nullptr, (const float*)nullptr, (const float*)nullptr);
        } else {
            attention(m, qkv, y, 1, s);
            hipLaunchKernelGGL((skinny_gemm_kernel<SKINNY_BIAS>), dim3(D / 16, 1), dim3(256), 0, s, y, D, (const bf16_t*)ly.wo, D, ly.bo, a.delta, D);
            hipLaunchKernelGGL((ln_slab_kernel<VEC, NT>), dim3(lb), dim3(256), 0, s, a.x, a.delta, slabs, 0, nullptr, y, ly.ln2w, ly.ln2b, S, m->eps, 0,
                               (const int*)nullptr, (const float*)nullptr, (const float*)nullptr);
        }
        hipLaunchKernelGGL((skinny_gemm_kernel<SKINNY_BIAS_QGELU>), dim3(FF / 16, 1), dim3(256), 0, s, y, D, (const bf16_t*)ly.w1, D, ly.b1, h, FF);
        hipLaunchKernelGGL((skinny_gemm_kernel<SKINNY_SLAB>), dim3(D / 16, FF / SKINNY_KC), dim3(256), 0, s, h, FF, (const bf16_t*)ly.w2, FF, nullptr, slabs, D);
        d1 = fuse ? nullptr : a.delta; b2 = ly.b2; n_slabs = FF / SKINNY_KC;
    }
    // the EOS row: the last layer's residual adds, final_layer_norm and the projection, fp32, one launch
    hipLaunchKernelGGL((text_head_one_kernel<VEC, NT>), dim3((unsigned)((m->E + 3) / 4)), dim3(256), 0, s, a.x, d1, slabs, n_slabs, b2, m->post_w,
                       m->post_b, m->proj, m->d_out, m->d_ids, S, m->E, m->eps);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(m->h_out_pin, m->d_out, (size_t)m->E * sizeof(float), hipMemcpyDeviceToHost, s));
}

// ... replayed as ONE graph launch from the third call on (first call: eager, which also sets the kernels' function
// attributes; second: captured while it is enqueued; `s` must be the handle's own stream)
void forward_text_one_graphed(mi_clip* m, hipStream_t s) {
    if (m->text_graph_state == 2) {
        HIP_CHECK(hipGraphLaunch(m->text_graph_exec, s));
        return;
    }
    if (m->text_graph_state == 0) {
        forward_text_one(m, s);
        m->text_graph_state = 1;
        return;
    }
    HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    try {
        forward_text_one(m, s);
    } catch (...) {
        hipGraph_t dead = nullptr;
        (void)hipStreamEndCapture(s, &dead);
        if (dead) (void)hipGraphDestroy(dead);
        throw;
    }
    // a capture or an instantiation that fails leaves nothing behind: the query is answered eagerly and the next call
    // tries again from a clean state
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    if (hipStreamEndCapture(s, &g) != hipSuccess || !g || hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (ge) (void)hipGraphExecDestroy(ge);
        if (g) (void)hipGraphDestroy(g);
        forward_text_one(m, s);
        return;
    }
    m->text_graph = g;
    m->text_graph_exec = ge;
    m->text_graph_state = 2;
    HIP_CHECK(hipGraphLaunch(m->text_graph_exec, s));
}

// n sequences whose ids are in m->d_ids -> m->d_out [n,E]
void forward_text(mi_clip* m, size_t n, hipStream_t s) {
    const int D = m->D, S = m->S, FF = m->FF;
    const size_t M = n * S;
    mi_clip::Act& a = m->act[0];
    hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)std::min<size_t>((M * (D / 4) + 255) / 256, 65535)), dim3(256), 0, s,
                       m->d_ids, m->tok, m->pos, a.x, M, S, D);
    hipLaunchKernelGGL(text_eos_rows_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, m->d_ids, (int)n, S, m->d_rows);
    HIP_CHECK(hipGetLastError());
    const bool deferred = m->precision == MI_PRECISION_BF16;
    const bf16_t *p1 = nullptr, *p2 = nullptr;  // bf16: the residual adds of the previous layer, applied by its LayerNorms
    for (const Layer& ly : m->layers) {  // the vision tower's block with the causal attention
        layer_norm(m, a.x, p1, p2, true, a.y, ly.ln1w, ly.ln1b, M, s);
        gemm<EPI_BIAS>(m, a.y, ly.wqkv, ly.bqkv, a.qkv, M, 3 * D, D, 3 * D, s);
        attention(m, a.qkv, a.y, n, s);
        if (deferred) {
            gemm<EPI_BIAS>(m, a.y, ly.wo, ly.bo, a.delta, M, D, D, D, s);
            layer_norm(m, a.x, a.delta, nullptr, false, a.y, ly.ln2w, ly.ln2b, M, s);
            gemm<EPI_BIAS_QGELU>(m, a.y, ly.w1, ly.b1, a.h, M, FF, D, FF, s);
            gemm<EPI_BIAS>(m, a.h, ly.w2, ly.b2, a.delta2, M, D, FF, D, s);
            p1 = a.delta; p2 = a.delta2;
        } else {
            gemm<EPI_BIAS_RESID>(m, a.y, ly.wo, ly.bo, a.x, M, D, D, D, s);
            layer_norm(m, a.x, nullptr, nullptr, true, a.y, ly.ln2w, ly.ln2b, M, s);
            gemm<EPI_BIAS_QGELU>(m, a.y, ly.w1, ly.b1, a.h, M, FF, D, FF, s);
            gemm<EPI_BIAS_RESID>(m, a.h, ly.w2, ly.b2, a.x, M, D, FF, D, s);
        }
    }
    MI_LN_DISPATCH(D, hipLaunchKernelGGL((head_kernel<VEC, NT>), dim3((unsigned)((n + 7) / 8), 16), dim3(256), 0, s, a.x, p1, p2, m->post_w, m->post_b, m->proj, m->d_out, (int)n, S, m->E, m->eps, (const int*)m->d_rows));
    HIP_CHECK(hipGetLastError());
}

void free_model(mi_clip* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    for (auto& a : m->aux) if (a) { (void)hipStreamSynchronize(a); (void)hipStreamDestroy(a); }
    m->order.destroy();
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    for (auto& e : m->ev_embed) if (e) (void)hipEventDestroy(e);
    for (auto& e : m->ev_front) if (e) (void)hipEventDestroy(e);
    for (auto& e : m->ev_join) if (e) (void)hipEventDestroy(e);
    if (m->copy_stream) { (void)hipStreamSynchronize(m->copy_stream); (void)hipStreamDestroy(m->copy_stream); }
    for (int b = 0; b < 2; ++b) {
        if (m->ev_up[b]) (void)hipEventDestroy(m->ev_up[b]);
        if (m->ev_used[b]) (void)hipEventDestroy(m->ev_used[b]);
        if (m->d_img_src[b]) (void)hipFree(m->d_img_src[b]);
    }
    if (m->d_img_tmp) (void)hipFree(m->d_img_tmp);
    drop_text_graph(m);
    if (m->h_ids_pin) (void)hipHostFree(m->h_ids_pin);
    if (m->h_out_pin) (void)hipHostFree(m->h_out_pin);
    for (void* p : m->allocs) (void)hipFree(p);
    for (void* p : m->ws) (void)hipFree(p);
    delete m;
}

}  // namespace

namespace mi {
hipStream_t clip_own_stream(mi_clip* m) { return own_stream(m); }
void clip_ensure_workspace(mi_clip* m, size_t n) { ensure_workspace(m, n); }
void clip_ensure_copy_stream(mi_clip* m) { ensure_copy_stream(m); }
void clip_forward(mi_clip* m, const float* d_img, size_t n, float* d_out, hipStream_t s, hipStream_t front) { forward(m, d_img, n, d_out, s, front); }
}  // namespace mi

extern "C" {

int mi_weights_list(const char* weights_path, char* buf, size_t cap, size_t* needed) {
    return guarded([&] {
        if (!weights_path) fail(MI_ERR_INVALID, "weights_path is null");
        const std::unique_ptr<WeightFile> file = open_weights(weights_path);
        std::string out;
        for (const std::string& name : file->names()) {
            const TensorInfo& t = file->info(name);
            out += name + " " + t.dtype + " [";
            for (size_t i = 0; i < t.shape.size(); ++i) out += (i ? "," : "") + std::to_string(t.shape[i]);
            out += "]\n";
        }
        for (const std::string& l : file->skipped_lines()) out += l + "\n";
        if (needed) *needed = out.size() + 1;
        if (buf && cap) {
            const size_t n = std::min(cap - 1, out.size());
            std::memcpy(buf, out.data(), n);
            buf[n] = '\0';
        }
    });
}

int mi_clip_set_option(mi_clip* m, const char* key, int value) {
    return guarded([&] {
        if (!m || !key) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(m->mu);
        DeviceGuard g(m->device);
        const std::string k(key);
        // a captured text query has kernel arguments and launch choices baked in: any option may change them
        if (m->text) { m->order.sync(); drop_text_graph(m); }
        if (k == "full_last") m->full_last = value != 0;
        else if (k == "attn_shift") m->attn_shift = value != 0;
        else if (k == "front_overlap") m->front_overlap = value != 0;
        else if (k == "attn_nt") m->attn_nt = value != 0;
        else if (k == "store_nt") m->store_nt = value != 0;
        else if (k == "qkv_pad") {   // the activation sets are sized by it: rebuild on next use
            if (value < 0 || value > 1024 || value % 64) fail(MI_ERR_INVALID, "qkv_pad must be a multiple of 64 in 0..1024");
            if (m->text) fail(MI_ERR_INVALID, "the text tower's qkv rows are dense");
            if (value != m->qkv_pad) {
                m->order.sync();
                for (void* p : m->ws) HIP_CHECK(hipFree(p));
                m->ws.clear();
                m->cap = 0;
                m->qkv_pad = value;
            }
        } else if (k == "qkv_layout") {   // takes effect with the next forward (every forward rewrites q|k|v)
            if (value < 0 || value > 1) fail(MI_ERR_INVALID, "qkv_layout must be 0 (token rows) or 1 (head-major planes)");
            if (m->text) fail(MI_ERR_INVALID, "the text tower's qkv rows are dense");
            m->qkv_layout = value;
        } else if (k == "attn_order") {
            if (value < 0 || value > 1) fail(MI_ERR_INVALID, "attn_order must be 0 or 1");
            m->attn_order = value;
        }
        else if (k == "split_tail") m->split_tail = value != 0;
        else if (k == "gemm_order") {
            if (value < 0 || value > 16) fail(MI_ERR_INVALID, "gemm_order must be 0..16");
            m->gemm_order = value;
        }
        else if (k == "im2col_rows") m->im2col_rows = value != 0;
        else if (k == "text_fast") m->text_fast = value != 0;
        else if (k == "text_fuse") m->text_fuse = value != 0;
        else if (k == "attn_f32_mfma") m->attn_f32_mfma = value != 0;
        else if (k == "ln_nt") m->ln_nt = value & 3;
        else if (k == "x24") m->x24 = value != 0;   // takes effect with the next forward (every forward rewrites the residual stream)
        else if (k == "ln_fold") {   // takes effect with the next forward (every forward rewrites the residual stream)
            if (value != 0 && !m->fold_ready)
                fail(MI_ERR_UNSUPPORTED, "ln_fold needs the bf16 image tower with hidden and intermediate sizes that are multiples of 256 and at least two layers");
            m->ln_fold = value != 0;
        }
        else if (k == "max_batch") {
            if (value < 1) fail(MI_ERR_INVALID, "max_batch must be >= 1");
            m->max_batch = (size_t)value;
        } else if (k == "parts") {
            if (value < 1 || value > 4) fail(MI_ERR_INVALID, "parts must be 1..4");
            if (m->text) fail(MI_ERR_INVALID, "the text tower runs as one stream");
            if (value != m->parts) {  // the activation sets are sized per part: rebuild on next use
                m->order.sync();
                for (void* p : m->ws) HIP_CHECK(hipFree(p));
                m->ws.clear();
                m->cap = 0;
                m->parts = value;
            }
        } else fail(MI_ERR_INVALID, "unknown option '%s' (full_last, front_overlap, attn_shift, attn_nt, store_nt, attn_order, qkv_pad, qkv_layout, split_tail, gemm_order, im2col_rows, ln_nt, x24, ln_fold, text_fast, text_fuse, attn_f32_mfma, max_batch, parts)", key);
    });
}

int mi_clip_load(const char* weights_path, int device, int precision, mi_clip** out) {
    mi_clip* m = nullptr;
    const int rc = guarded([&] {
        if (!out) fail(MI_ERR_INVALID, "out is null");
        *out = nullptr;
        if (!weights_path) fail(MI_ERR_INVALID, "weights_path is null");
        if (precision != MI_PRECISION_F32 && precision != MI_PRECISION_BF16 && precision != MI_PRECISION_BF16_SPLIT)
            fail(MI_ERR_INVALID, "precision %d: use MI_PRECISION_F32 (0), MI_PRECISION_BF16 (1) or MI_PRECISION_BF16_SPLIT (2)", precision);
        DeviceGuard g(device);
        m = new mi_clip();
        m->device = device;
        m->precision = precision == MI_PRECISION_BF16_SPLIT ? MI_PRECISION_BF16 : precision;
        m->split_ln = precision == MI_PRECISION_BF16_SPLIT;
        if (const char* e = std::getenv("MI_CLIP_MAX_BATCH")) m->max_batch = std::max(1, std::atoi(e));
        // aux streams are created on first use: HIP multiplexes streams onto 4 hardware queues
        // (GPU_MAX_HW_QUEUES), and two streams that share a queue do not overlap
        HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
        for (auto& e : m->ev_join) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (const char* e = std::getenv("MI_CLIP_PARTS")) m->parts = std::min(4, std::max(1, std::atoi(e)));
        if (const char* e = std::getenv("MI_CLIP_FULL_LAST")) m->full_last = std::atoi(e) != 0;
        if (const char* e = std::getenv("MI_CLIP_ATTN")) m->attn_ver = std::atoi(e) == 1 ? 1 : 2;  // fixed at load: decides the q scale
        if (const char* e = std::getenv("MI_GEMM_SPLIT")) m->split_tail = std::atoi(e) != 0;
        if (const char* e = std::getenv("MI_GEMM_ORDER")) m->gemm_order = std::max(0, std::min(16, std::atoi(e)));
        if (const char* e = std::getenv("MI_CLIP_IM2COL")) m->im2col_rows = std::atoi(e) != 0;
        if (const char* e = std::getenv("MI_CLIP_LN_NT")) m->ln_nt = std::atoi(e) & 3;
        if (const char* e = std::getenv("MI_CLIP_X24")) m->x24 = std::atoi(e) != 0;
        if (const char* e = std::getenv("MI_CLIP_LN_CENTER")) m->ln_center = std::atoi(e) != 0;   // fixed at load: it shapes the weights
        if (const char* e = std::getenv("MI_CLIP_QKV_LAYOUT")) m->qkv_layout = std::atoi(e) == 1 ? 1 : 0;
        if (const char* e = std::getenv("MI_CLIP_FRONT_OVERLAP")) m->front_overlap = std::atoi(e) != 0;
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        m->n_cu = prop.multiProcessorCount;
        load_weights(m, weights_path);
        if (const char* e = std::getenv("MI_CLIP_LN_FOLD")) m->ln_fold = std::atoi(e) != 0 && m->fold_ready;
        *out = m;
    });
    if (rc != MI_OK && m) free_model(m);
    return rc;
}

void mi_clip_free(mi_clip* m) { free_model(m); }

int mi_clip_load_text(const char* weights_path, int device, int precision, mi_clip** out) {
    mi_clip* m = nullptr;
    const int rc = guarded([&] {
        if (!out) fail(MI_ERR_INVALID, "out is null");
        *out = nullptr;
        if (!weights_path) fail(MI_ERR_INVALID, "weights_path is null");
        if (precision != MI_PRECISION_F32 && precision != MI_PRECISION_BF16)
            fail(MI_ERR_UNSUPPORTED, "the text tower runs in MI_PRECISION_F32 or MI_PRECISION_BF16");
        DeviceGuard g(device);
        m = new mi_clip();
        m->device = device;
        m->precision = precision;
        m->text = true;
        m->parts = 1;
        if (const char* e = std::getenv("MI_CLIP_TEXT_FAST")) m->text_fast = std::atoi(e) != 0;  // A/B hook, read at load
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        m->n_cu = prop.multiProcessorCount;
        load_text_weights(m, weights_path);
        *out = m;
    });
    if (rc != MI_OK && m) free_model(m);
    return rc;
}

int mi_clip_embed_text(mi_clip* m, const int32_t* input_ids, size_t n, float* out) {
    return guarded([&] {
        if (!m) fail(MI_ERR_INVALID, "null model handle");
        if (!m->text) fail(MI_ERR_INVALID, "this handle holds the image tower: load the text tower with mi_clip_load_text");
        if (n == 0) return;
        if (!input_ids || !out) fail(MI_ERR_INVALID, "null buffer");
        for (size_t i = 0; i < n * (size_t)m->S; ++i)
            if (input_ids[i] < 0 || input_ids[i] >= m->vocab)
                fail(MI_ERR_INVALID, "token id %d at position %zu is outside the vocabulary of %d", input_ids[i], i, m->vocab);
        std::lock_guard<std::mutex> l(m->mu);
        DeviceGuard g(m->device);
        const size_t chunk = std::min(n, m->max_batch);
        ensure_text_workspace(m, chunk);
        if (text_one_applies(m, n)) {  // one query: pinned staging, the copies ride in the graph
            if (!m->h_ids_pin) {
                HIP_CHECK(hipHostMalloc((void**)&m->h_ids_pin, (size_t)m->S * sizeof(int32_t), hipHostMallocDefault));
                HIP_CHECK(hipHostMalloc((void**)&m->h_out_pin, (size_t)m->E * sizeof(float), hipHostMallocDefault));
            }
            std::memcpy(m->h_ids_pin, input_ids, (size_t)m->S * sizeof(int32_t));
            m->order.begin(m->stream);
            forward_text_one_graphed(m, m->stream);
            HIP_CHECK(hipStreamSynchronize(m->stream));
            m->order.pending = false;
            std::memcpy(out, m->h_out_pin, (size_t)m->E * sizeof(float));
            return;
        }
        for (size_t i = 0; i < n; i += chunk) {
            const size_t c = std::min(chunk, n - i);
            m->order.begin(m->stream);
            HIP_CHECK(hipMemcpyAsync(m->d_ids, input_ids + i * m->S, c * m->S * sizeof(int32_t), hipMemcpyHostToDevice, m->stream));
            forward_text(m, c, m->stream);
            HIP_CHECK(hipMemcpyAsync(out + i * m->E, m->d_out, c * m->E * 4, hipMemcpyDeviceToHost, m->stream));
            HIP_CHECK(hipStreamSynchronize(m->stream));
            m->order.pending = false;
        }
    });
}

int mi_clip_info(const mi_clip* m, uint32_t out[8]) {
    return guarded([&] {
        if (!m || !out) fail(MI_ERR_INVALID, "null argument");
        const int v[8] = {m->image, m->patch, m->S, m->D, m->L, m->H, m->FF, m->E};
        for (int i = 0; i < 8; ++i) out[i] = (uint32_t)v[i];
    });
}

int mi_clip_ln_fold_stats(mi_clip* m, uint64_t out[2], int reset) {
    return guarded([&] {
        if (!m || !out) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(m->mu);
        DeviceGuard g(m->device);
        out[0] = out[1] = 0;
        if (!m->d_fold_offset_rows) return;   // no LayerNorm-free loop on this handle: nothing was ever looked at
        m->order.sync();                      // the forwards enqueued so far have counted
        unsigned long long v = 0;
        HIP_CHECK(hipMemcpy(&v, m->d_fold_offset_rows, sizeof v, hipMemcpyDeviceToHost));
        out[0] = v;
        out[1] = m->fold_rows_checked;
        if (reset) {
            HIP_CHECK(hipMemset(m->d_fold_offset_rows, 0, sizeof v));
            m->fold_rows_checked = 0;
        }
    });
}

int mi_clip_embed_device(mi_clip* m, const float* d_nchw, size_t n, float* d_out, void* stream) {
    return guarded([&] {
        if (!m) fail(MI_ERR_INVALID, "null model handle");
        if (m->text) fail(MI_ERR_INVALID, "this handle holds the text tower (mi_clip_load_text)");
        if (n == 0) return;
        if (!d_nchw || !d_out) fail(MI_ERR_INVALID, "null buffer");
        std::lock_guard<std::mutex> l(m->mu);
        DeviceGuard g(m->device);
        hipStream_t s = stream ? (hipStream_t)stream : own_stream(m);
        const size_t px = (size_t)m->image * m->image * 3;
        const size_t chunk = std::min(n, m->max_batch);
        ensure_workspace(m, chunk);
        // the handle's work runs in call order whatever stream each call names (one workspace per handle)
        m->order.begin(s);
        for (size_t i = 0; i < n; i += chunk) {
            const size_t c = std::min(chunk, n - i);
            forward(m, d_nchw + i * px, c, d_out + i * m->E, s);
        }
        m->order.end(s);
    });
}

int mi_clip_embed(mi_clip* m, const float* nchw, size_t n, float* out) {
    return guarded([&] {
        if (!m) fail(MI_ERR_INVALID, "null model handle");
        if (m->text) fail(MI_ERR_INVALID, "this handle holds the text tower (mi_clip_load_text)");
        if (n == 0) return;  // the reference forwards an empty chunk (server/src/clip.rs:112-118)
        if (!nchw || !out) fail(MI_ERR_INVALID, "null buffer");
        std::lock_guard<std::mutex> l(m->mu);
        DeviceGuard g(m->device);
        const size_t px = (size_t)m->image * m->image * 3;
        const size_t chunk = std::min(n, m->max_batch);
        ensure_workspace(m, chunk);
        ensure_copy_stream(m);
        // two input / output buffers: the upload of chunk i+1 (copy stream) runs under the forward of
        // chunk i, the readback of chunk i under the forward of chunk i+1
        float* din[2] = {m->d_in, m->d_in2};
        float* dout[2] = {m->d_out, m->d_out2};
        const size_t nchunks = (n + chunk - 1) / chunk;
        m->order.begin(m->stream);
        auto upload = [&](size_t ci) {
            const size_t i = ci * chunk, c = std::min(chunk, n - i);
            const int b = (int)(ci & 1);
            if (ci >= 2) HIP_CHECK(hipStreamWaitEvent(m->copy_stream, m->ev_used[b], 0));  // forward(ci-2) consumed it
            HIP_CHECK(hipMemcpyAsync(din[b], nchw + i * px, c * px * 4, hipMemcpyHostToDevice, m->copy_stream));
            HIP_CHECK(hipEventRecord(m->ev_up[b], m->copy_stream));
        };
        upload(0);
        for (size_t ci = 0; ci < nchunks; ++ci) {
            const size_t i = ci * chunk, c = std::min(chunk, n - i);
            const int b = (int)(ci & 1);
            HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_up[b], 0));
            forward(m, din[b], c, dout[b], m->stream, m->front_overlap ? m->copy_stream : nullptr);
            HIP_CHECK(hipEventRecord(m->ev_used[b], m->stream));
            if (ci + 1 < nchunks) upload(ci + 1);  // pageable source: blocks this thread while the GPU computes
            HIP_CHECK(hipMemcpyAsync(out + i * m->E, dout[b], c * m->E * 4, hipMemcpyDeviceToHost, m->stream));
        }
        HIP_CHECK(hipStreamSynchronize(m->stream));
        m->order.pending = false;
    });
}

int mi_clip_embed_rgb8(mi_clip* m, const uint8_t* rgb8, size_t n, float* out) {
    return guarded([&] {
        if (!m) fail(MI_ERR_INVALID, "null model handle");
        if (m->text) fail(MI_ERR_INVALID, "this handle holds the text tower (mi_clip_load_text)");
        if (n == 0) return;
        if (!rgb8 || !out) fail(MI_ERR_INVALID, "null buffer");
        std::lock_guard<std::mutex> l(m->mu);
        DeviceGuard g(m->device);
        const size_t plane = (size_t)m->image * m->image, px = plane * 3;
        const size_t chunk = std::min(n, m->max_batch);
        ensure_workspace(m, chunk);
        m->order.begin(m->stream);
        for (size_t i = 0; i < n; i += chunk) {
            const size_t c = std::min(chunk, n - i);
            HIP_CHECK(hipMemcpyAsync(m->d_rgb, rgb8 + i * px, c * px, hipMemcpyHostToDevice, m->stream));
            hipLaunchKernelGGL(preprocess_rgb8_kernel, dim3(1024), dim3(256), 0, m->stream, m->d_rgb, m->d_in, c * plane, plane);
            HIP_CHECK(hipGetLastError());
            forward(m, m->d_in, c, m->d_out, m->stream);
            HIP_CHECK(hipMemcpyAsync(out + i * m->E, m->d_out, c * m->E * 4, hipMemcpyDeviceToHost, m->stream));
            HIP_CHECK(hipStreamSynchronize(m->stream));
            m->order.pending = false;
        }
    });
}

// The whole of server/src/clip.rs:92-124 for one chunk: decoded RGB8 images of any size ->
// CatmullRom resize + ImageNet normalisation on the device, straight into the tower's input.
int mi_clip_embed_images(mi_clip* m, const uint8_t* const* rgb8, const uint32_t* widths, const uint32_t* heights,
                         size_t n, float* out) {
    return guarded([&] {
        if (!m) fail(MI_ERR_INVALID, "null model handle");
        if (m->text) fail(MI_ERR_INVALID, "this handle holds the text tower (mi_clip_load_text)");
        if (n == 0) return;
        if (!rgb8 || !widths || !heights || !out) fail(MI_ERR_INVALID, "null buffer");
        size_t max_src = 0, max_tmp = 0;
        for (size_t i = 0; i < n; ++i) {
            if (!rgb8[i]) fail(MI_ERR_INVALID, "null image %zu", i);
            const char* why = nullptr;
            if (!resize_supported(widths[i], heights[i], (uint32_t)m->image, (uint32_t)m->image, &why))
                fail(MI_ERR_UNSUPPORTED, "image %zu (%ux%u): %s", i, widths[i], heights[i], why);
            max_src = std::max(max_src, (size_t)widths[i] * heights[i] * 3);
            max_tmp = std::max(max_tmp, (size_t)m->image * widths[i] * 3 * 4);
        }
        std::lock_guard<std::mutex> l(m->mu);
        DeviceGuard g(m->device);
        const size_t plane = (size_t)m->image * m->image, px = plane * 3;
        const size_t chunk = std::min(n, m->max_batch);
        ensure_workspace(m, chunk);
        // two source buffers: the upload of image i+1 overlaps the resize of image i
        if (max_src > m->img_src_cap) {
            for (int b = 0; b < 2; ++b) {
                if (m->d_img_src[b]) HIP_CHECK(hipFree(m->d_img_src[b]));
                m->d_img_src[b] = nullptr;
                HIP_CHECK(hipMalloc((void**)&m->d_img_src[b], max_src));
            }
            m->img_src_cap = max_src;
        }
        if (max_tmp > m->img_tmp_cap) {
            if (m->d_img_tmp) HIP_CHECK(hipFree(m->d_img_tmp));
            m->d_img_tmp = nullptr;
            HIP_CHECK(hipMalloc((void**)&m->d_img_tmp, max_tmp));
            m->img_tmp_cap = max_tmp;
        }
        ensure_copy_stream(m);
        m->order.begin(m->stream);
        for (size_t i0 = 0; i0 < n; i0 += chunk) {
            const size_t c = std::min(chunk, n - i0);
            for (size_t j = 0; j < c; ++j) {
                const size_t i = i0 + j;
                const int b = (int)(i & 1);
                if (i >= 2) HIP_CHECK(hipStreamWaitEvent(m->copy_stream, m->ev_used[b], 0));  // buffer b consumed
                HIP_CHECK(hipMemcpyAsync(m->d_img_src[b], rgb8[i], (size_t)widths[i] * heights[i] * 3, hipMemcpyHostToDevice, m->copy_stream));
                HIP_CHECK(hipEventRecord(m->ev_up[b], m->copy_stream));
                HIP_CHECK(hipStreamWaitEvent(m->stream, m->ev_up[b], 0));
                resize_catmullrom_launch<true>(m->d_img_src[b], widths[i], heights[i], (uint32_t)m->image, (uint32_t)m->image,
                                               m->d_img_tmp, nullptr, m->d_in + j * px, m->stream);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipEventRecord(m->ev_used[b], m->stream));
            }
            forward(m, m->d_in, c, m->d_out, m->stream);
            HIP_CHECK(hipMemcpyAsync(out + i0 * m->E, m->d_out, c * m->E * 4, hipMemcpyDeviceToHost, m->stream));
            HIP_CHECK(hipStreamSynchronize(m->stream));
            m->order.pending = false;
        }
    });
}


// ---- op-level entry points (include/mi355clip_ops.h): host buffers in fp32, the op runs
// on the device in `precision`; used by the per-op parity tests.
namespace {
struct Scratch {
    std::vector<void*> p;
    ~Scratch() { for (void* q : p) (void)hipFree(q); }
    void* bytes(size_t b) { void* q = nullptr; HIP_CHECK(hipMalloc(&q, std::max<size_t>(b, 16))); HIP_CHECK(hipMemset(q, 0, std::max<size_t>(b, 16))); p.push_back(q); return q; }
    // host f32 [rows][cols] -> device T [rows_pad][cols]
    void* up(int precision, const float* h, size_t rows, size_t cols, size_t rows_pad) {
        void* d = bytes(rows_pad * cols * (precision == MI_PRECISION_F32 ? 4 : 2));
        if (precision == MI_PRECISION_F32) HIP_CHECK(hipMemcpy(d, h, rows * cols * 4, hipMemcpyHostToDevice));
        else {
            std::vector<uint16_t> b(rows * cols);
            for (size_t i = 0; i < b.size(); ++i) b[i] = f32_to_bf16_host(h[i]);
            HIP_CHECK(hipMemcpy(d, b.data(), b.size() * 2, hipMemcpyHostToDevice));
        }
        return d;
    }
    void down(int precision, const void* d, float* h, size_t n) {
        if (precision == MI_PRECISION_F32) { HIP_CHECK(hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost)); return; }
        std::vector<uint16_t> b(n);
        HIP_CHECK(hipMemcpy(b.data(), d, n * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) { uint32_t u = (uint32_t)b[i] << 16; std::memcpy(&h[i], &u, 4); }
    }
};
}  // namespace

int mi_op_linear(int device, int precision, int epilogue, const float* x, const float* w, const float* bias,
                 float* out, size_t m_rows, int n, int k) {
    return guarded([&] {
        if (!x || !w || !out) fail(MI_ERR_INVALID, "null buffer");
        if (epilogue != EPI_STORE_F32 && !bias) fail(MI_ERR_INVALID, "bias is null");
        DeviceGuard g(device);
        Scratch sc;
        const size_t mp = pad256(m_rows);
        void* dx = sc.up(precision, x, m_rows, k, mp);
        void* dw = sc.up(precision, w, n, k, n);
        float* db = bias ? (float*)sc.up(MI_PRECISION_F32, bias, 1, n, 1) : nullptr;
        const bool f32_out = epilogue == EPI_STORE_F32 || epilogue == EPI_BIAS_RESID;
        void* dout = f32_out ? sc.up(MI_PRECISION_F32, out, m_rows, n, mp) : sc.bytes(mp * n * 4);
        // same dispatcher as the tower (persistent 256x256 kernel when the shape allows);
        // MI_OP_GRID overrides the CU count so that small test shapes exercise several tiles
        // per workgroup and the split last round
        mi_clip mm;
        mm.precision = precision;
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        mm.n_cu = prop.multiProcessorCount;
        if (const char* e = std::getenv("MI_OP_GRID")) mm.n_cu = std::max(1, std::atoi(e));
        if (const char* e = std::getenv("MI_OP_GEMM_ORDER")) mm.gemm_order = std::max(0, std::min(16, std::atoi(e)));
        switch (epilogue) {
            case EPI_STORE_F32: gemm<EPI_STORE_F32>(&mm, dx, dw, db, dout, m_rows, n, k, n, nullptr); break;
            case EPI_BIAS: gemm<EPI_BIAS>(&mm, dx, dw, db, dout, m_rows, n, k, n, nullptr); break;
            case EPI_BIAS_QGELU: gemm<EPI_BIAS_QGELU>(&mm, dx, dw, db, dout, m_rows, n, k, n, nullptr); break;
            case EPI_BIAS_RESID: gemm<EPI_BIAS_RESID>(&mm, dx, dw, db, dout, m_rows, n, k, n, nullptr); break;
            default: fail(MI_ERR_INVALID, "epilogue %d", epilogue);
        }
        HIP_CHECK(hipDeviceSynchronize());
        sc.down(f32_out ? MI_PRECISION_F32 : precision, dout, out, m_rows * n);
    });
}

// the two epilogues of the LayerNorm-free tower, one persistent-GEMM launch each (bf16; MI_OP_GRID / MI_OP_GEMM_ORDER as in mi_op_linear)
namespace {
void op_pp_model(mi_clip& mm, int device) {
    mm.precision = MI_PRECISION_BF16;
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    mm.n_cu = prop.multiProcessorCount;
    if (const char* e = std::getenv("MI_OP_GRID")) mm.n_cu = std::max(1, std::atoi(e));
    if (const char* e = std::getenv("MI_OP_GEMM_ORDER")) mm.gemm_order = std::max(0, std::min(16, std::atoi(e)));
    if (const char* e = std::getenv("MI_GEMM_SPLIT")) mm.split_tail = std::atoi(e) != 0;
    if (const char* e = std::getenv("MI_OP_STORE_NT")) mm.store_nt = std::atoi(e) != 0;   // test hook: cache policy of the output stores
}
}  // namespace

int mi_op_linear_lnf(int device, int epilogue, const float* x, const float* w, const float* bias, const float* c,
                     const float* stats, float* out, size_t m_rows, int n, int k) {
    return guarded([&] {
        if (!x || !w || !bias || !c || !stats || !out) fail(MI_ERR_INVALID, "null buffer");
        if (epilogue != EPI_LNF && epilogue != EPI_LNF_QGELU) fail(MI_ERR_INVALID, "epilogue %d", epilogue);
        const size_t mp = pad256(m_rows);
        if (!pp_shape_ok(mp, n, k, n)) fail(MI_ERR_UNSUPPORTED, "shape [%zu x %d x %d] is not one of the persistent GEMM's", m_rows, n, k);
        DeviceGuard g(device);
        Scratch sc;
        void* dx = sc.up(MI_PRECISION_BF16, x, m_rows, k, mp);
        void* dw = sc.up(MI_PRECISION_BF16, w, n, k, n);
        float* db = (float*)sc.up(MI_PRECISION_F32, bias, 1, n, 1);
        PpFold f;
        f.cvec = (float*)sc.up(MI_PRECISION_F32, c, 1, n, 1);
        f.stats = (float*)sc.up(MI_PRECISION_F32, stats, m_rows, 2, mp);
        void* dout = sc.bytes(mp * n * 2);
        mi_clip mm;
        op_pp_model(mm, device);
        if (epilogue == EPI_LNF) launch_pp<EPI_LNF>(&mm, dx, dw, db, dout, mp, n, k, n, f, nullptr);
        else launch_pp<EPI_LNF_QGELU>(&mm, dx, dw, db, dout, mp, n, k, n, f, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        sc.down(MI_PRECISION_BF16, dout, out, m_rows * n);
    });
}

int mi_op_linear_resid24(int device, const float* x, const float* w, const float* bias, float* xres, float* hi_out,
                         float* part, float* stats, size_t m_rows, int n, int k, float eps) {
    return guarded([&] {
        if (!x || !w || !bias || !xres) fail(MI_ERR_INVALID, "null buffer");
        const size_t mp = pad256(m_rows);
        if (!pp_shape_ok(mp, n, k, n) || mp * (size_t)(n / 32) * 8 >= (1ull << 32))
            fail(MI_ERR_UNSUPPORTED, "shape [%zu x %d x %d] is not one of the persistent GEMM's", m_rows, n, k);
        DeviceGuard g(device);
        Scratch sc;
        void* dx = sc.up(MI_PRECISION_BF16, x, m_rows, k, mp);
        void* dw = sc.up(MI_PRECISION_BF16, w, n, k, n);
        float* db = (float*)sc.up(MI_PRECISION_F32, bias, 1, n, 1);
        // the residual rows as the two planes: b' = bits + 0x8080, hi = b' >> 16, lo = (b' >> 8) & 0xFF; padding rows = 0.0
        const size_t cnt = mp * (size_t)n;
        std::vector<uint16_t> hi(cnt, 0);
        std::vector<uint8_t> lo(cnt, 0x80);
        for (size_t i = 0; i < m_rows * (size_t)n; ++i) {
            uint32_t u;
            std::memcpy(&u, &xres[i], 4);
            u += 0x8080u;
            hi[i] = (uint16_t)(u >> 16);
            lo[i] = (uint8_t)(u >> 8);
        }
        uint16_t* dhi = (uint16_t*)sc.bytes(cnt * 2);
        uint8_t* dlo = (uint8_t*)sc.bytes(cnt);
        HIP_CHECK(hipMemcpy(dhi, hi.data(), cnt * 2, hipMemcpyHostToDevice));
        HIP_CHECK(hipMemcpy(dlo, lo.data(), cnt, hipMemcpyHostToDevice));
        const int nb = n / 32;
        PpFold f;
        f.xlo = dlo;
        f.part = (float*)sc.bytes(mp * nb * 8);
        float* dstats = (float*)sc.bytes(mp * 8);
        mi_clip mm;
        op_pp_model(mm, device);
        launch_pp<EPI_RESID24>(&mm, dx, dw, db, dhi, mp, n, k, n, f, nullptr);
        hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)(mp / 16)), dim3(256), 0, nullptr, f.part, dstats, (int)mp, nb, 1.0f / (float)n, eps);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(hi.data(), dhi, cnt * 2, hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(lo.data(), dlo, cnt, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < m_rows * (size_t)n; ++i) {
            const uint32_t u = (((uint32_t)hi[i] << 16) | ((uint32_t)lo[i] << 8)) - X24B_BIAS;
            std::memcpy(&xres[i], &u, 4);
            if (hi_out) { const uint32_t h = (uint32_t)hi[i] << 16; std::memcpy(&hi_out[i], &h, 4); }
        }
        if (part) HIP_CHECK(hipMemcpy(part, f.part, m_rows * nb * 8, hipMemcpyDeviceToHost));
        if (stats) HIP_CHECK(hipMemcpy(stats, dstats, m_rows * 8, hipMemcpyDeviceToHost));
    });
}

int mi_op_attention(int device, int precision, const float* qkv, float* ctx, size_t n_img, int s_tok, int d, int heads) {
    return guarded([&] {
        if (!qkv || !ctx) fail(MI_ERR_INVALID, "null buffer");
        if (d != heads * 64) fail(MI_ERR_UNSUPPORTED, "head_dim must be 64");
        DeviceGuard g(device);
        Scratch sc;
        mi_clip m;
        m.precision = precision; m.S = s_tok; m.D = d; m.H = heads;
        if (const char* e = std::getenv("MI_OP_ATTN")) m.attn_ver = std::atoi(e) == 1 ? 1 : 2;   // test hook: which bf16 kernel
        if (const char* e = std::getenv("MI_OP_ATTN_SHIFT")) m.attn_shift = std::atoi(e) != 0;
        if (const char* e = std::getenv("MI_OP_ATTN_ORDER")) m.attn_order = std::atoi(e) != 0;   // test hook: first pair of a workgroup
        if (const char* e = std::getenv("MI_OP_ATTN_NT")) m.attn_nt = std::atoi(e) != 0;         // test hook: cache policy of the K / V / q stream
        if (const char* e = std::getenv("MI_OP_ATTN_F32_MFMA")) m.attn_f32_mfma = std::atoi(e) != 0;   // test hook: 0 = the one-thread-per-query fp32 kernel
        const size_t rows = n_img * s_tok;
        void* dq = sc.up(precision, qkv, rows, 3 * (size_t)d, pad256(rows));
        void* dc = sc.bytes(pad256(rows) * d * 4);
        int ldq = 3 * d;
        if (const char* e = std::getenv("MI_OP_ATTN_QKV_PAD")) {   // test hook: the padded row pitch the tower gives q|k|v where attn32 runs
            const int pad = std::atoi(e);
            if (pad > 0 && attn32_applies(&m)) {
                ldq = 3 * d + pad;
                void* padded = sc.bytes(pad256(rows) * (size_t)ldq * 2);
                HIP_CHECK(hipMemset(padded, 0xFF, pad256(rows) * (size_t)ldq * 2));   // NaN patterns between the rows: a stray read shows
                HIP_CHECK(hipMemcpy2D(padded, (size_t)ldq * 2, dq, (size_t)3 * d * 2, (size_t)3 * d * 2, rows, hipMemcpyDeviceToDevice));
                dq = padded;
            }
        }
        size_t hm_rows = 0;
        if (const char* e = std::getenv("MI_OP_ATTN_LAYOUT")) {   // test hook: 1 = head-major planes [3][H][Mp][64] (option "qkv_layout")
            if (std::atoi(e) == 1 && attn32_applies(&m)) {
                hm_rows = pad256(rows);
                std::vector<uint16_t> planes((size_t)3 * heads * hm_rows * 64, 0xFFFFu);   // NaN patterns in the padding rows
                for (size_t r = 0; r < rows; ++r)
                    for (size_t c = 0; c < 3 * (size_t)d; ++c) {
                        const size_t sel = c / d, hh = (c % d) / 64, el = c % 64;
                        planes[((sel * heads + hh) * hm_rows + r) * 64 + el] = f32_to_bf16_host(qkv[r * 3 * (size_t)d + c]);
                    }
                dq = sc.bytes(planes.size() * 2);
                HIP_CHECK(hipMemcpy(dq, planes.data(), planes.size() * 2, hipMemcpyHostToDevice));
            }
        }
        attention(&m, dq, dc, n_img, nullptr, false, ldq, hm_rows);
        HIP_CHECK(hipDeviceSynchronize());
        sc.down(precision, dc, ctx, rows * d);
    });
}

int mi_op_clock_probe(int device, void* stream, float* mhz) {
    return guarded([&] {
        if (!mhz) fail(MI_ERR_INVALID, "null argument");
        DeviceGuard g(device);
        unsigned long long* d = nullptr;
        HIP_CHECK(hipMalloc((void**)&d, 4 * sizeof(unsigned long long)));
        hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d);
        unsigned long long h[4] = {0, 0, 0, 0};
        const hipError_t e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream);
        const hipError_t e2 = e == hipSuccess ? hipStreamSynchronize((hipStream_t)stream) : e;
        (void)hipFree(d);
        HIP_CHECK(e2);
        *mhz = h[1] ? (float)((double)h[0] / (double)h[1] * 100.0) : 0.0f;  // shader cycles per 100 MHz reference tick
    });
}

int mi_op_layernorm(int device, int precision, const float* x, const float* w, const float* b, float* y, size_t rows,
                    int d, float eps) {
    return guarded([&] {
        if (!x || !w || !b || !y) fail(MI_ERR_INVALID, "null buffer");
        DeviceGuard g(device);
        Scratch sc;
        mi_clip m;
        m.precision = precision; m.D = d; m.eps = eps;
        float* dx = (float*)sc.up(MI_PRECISION_F32, x, rows, d, rows);
        float* dw = (float*)sc.up(MI_PRECISION_F32, w, 1, d, 1);
        float* db = (float*)sc.up(MI_PRECISION_F32, b, 1, d, 1);
        void* dy = sc.bytes(rows * d * 4);
        layer_norm(&m, dx, nullptr, nullptr, true, dy, dw, db, rows, nullptr);
        HIP_CHECK(hipDeviceSynchronize());
        sc.down(precision, dy, y, rows * d);
    });
}

}  // extern "C"
