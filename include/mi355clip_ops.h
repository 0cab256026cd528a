/*
 * mi355clip_ops.h — op-level entry points of libmi355clip.so.
 *
 * Not part of the drop-in boundary (that is mi355clip.h): these run ONE device
 * kernel of the vision tower on host fp32 buffers so that tests can check each
 * against the CPU oracle (SURVEY.md §8c "per-op golden vectors").  Each call
 * allocates, copies, launches, synchronises and frees; they are not fast paths.
 * `precision` is MI_PRECISION_F32 or MI_PRECISION_BF16 as in mi355clip.h; in bf16
 * the operands are rounded to bf16 on the way in and results widened on the way out.
 */
#ifndef MI355CLIP_OPS_H
#define MI355CLIP_OPS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MI_EPI_STORE_F32 0  /* out = x w^T                      (patch embedding, no bias) */
#define MI_EPI_BIAS 1       /* out = x w^T + bias               (q/k/v projection)         */
#define MI_EPI_BIAS_QGELU 2 /* out = quick_gelu(x w^T + bias)   (fc1)                      */
#define MI_EPI_BIAS_RESID 3 /* out += x w^T + bias              (out_proj, fc2; fp32 out)  */

/* x [m_rows][k], w [n][k] (PyTorch [out,in]), bias [n] (NULL for STORE_F32),
 * out [m_rows][n]; n % 128 == 0, k % 64 == 0 (bf16) or k % 16 == 0 (f32). */
int mi_op_linear(int device, int precision, int epilogue, const float* x, const float* w, const float* bias,
                 float* out, size_t m_rows, int n, int k);

/* The two GEMM epilogues of the bf16 tower without LayerNorm kernels (mi_clip_set_option "ln_fold"); bf16 operands,
 * the persistent kernel only: n % 256 == 0, k % 64 == 0, k >= 128.
 *
 * mi_op_linear_lnf: the LayerNorm in front of a linear, finished in its epilogue —
 *   out[m][j] = act(stats[m][0] * (x w^T)[m][j] + stats[m][1] * c[j] + bias[j]),
 *   x the UN-normalised rows, w = W diag(gamma), c[j] = sum_k w[j][k], bias = W beta + b, stats[m] = {rstd, -mean rstd};
 *   epilogue MI_EPI_LNF (act = identity: q/k/v) or MI_EPI_LNF_QGELU (fc1).
 * mi_op_linear_resid24: the residual add and the row sums in the epilogue —
 *   xres[m][j] (fp32, in/out; kept as two 24-bit planes around the launch) += bf16(x w^T + bias)[m][j];
 *   hi_out[m][j] = the hi plane = bf16(new xres), the next GEMM's operand; part[m][n/32][2] = {sum, sum of squares}
 *   of the new row per 32-column block; stats[m] = {rstd, -mean rstd} from them (ln_stats_kernel).  hi_out, part,
 *   stats may be NULL. */
#define MI_EPI_LNF 4
#define MI_EPI_LNF_QGELU 5
int mi_op_linear_lnf(int device, int epilogue, const float* x, const float* w, const float* bias, const float* c,
                     const float* stats, float* out, size_t m_rows, int n, int k);
int mi_op_linear_resid24(int device, const float* x, const float* w, const float* bias, float* xres, float* hi_out,
                         float* part, float* stats, size_t m_rows, int n, int k, float eps);

/* softmax(q k^T / 8) v per (image, head): qkv [n_img][s_tok][3*d] (q|k|v, head h =
 * columns h*64..h*64+63 of each third) -> ctx [n_img][s_tok][d]; d == heads*64. */
int mi_op_attention(int device, int precision, const float* qkv, float* ctx, size_t n_img, int s_tok, int d,
                    int heads);

/* y = (x - mean) / sqrt(var + eps) * w + b over the last axis of x [rows][d]. */
int mi_op_layernorm(int device, int precision, const float* x, const float* w, const float* b, float* y,
                    size_t rows, int d, float eps);

/* Diagnostic: the shader clock at this moment of `stream` (a hipStream_t; NULL = the null stream).  One wave runs a
 * fixed dependent-FMA loop (~0.1 ms) between two readings of the shader-cycle counter (s_memtime) and of the 100 MHz
 * reference counter (s_memrealtime); the call waits for it and returns *mhz = shader cycles / reference time.  Enqueued
 * behind a forward on the same stream it reads the clock the chip holds under that load. */
int mi_op_clock_probe(int device, void* stream, float* mhz);

#ifdef __cplusplus
}
#endif
#endif
