/*
 * mi355clip.h — C ABI of libmi355clip.so: the MI355X (gfx950) replacement for the
 * one compute hot path of olFi95/image_search:
 *   Seam A  CLIP ViT-L/14 image tower   [n,3,224,224] f32 -> [n,768] f32
 *   Seam B  cosine K-nearest over the stored [N,768] f32 embedding table
 *   + the 24-line query refinement (average_slices).
 * The reference has no FFI of its own; each entry point below cites the reference
 * call site (file:line under /root/reference) whose semantics it pins.  The Rust
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; opaque handles; caller owns every buffer.
 *   - every function returns MI_OK (0) or a negative MI_ERR_*; nothing throws,
 *     aborts or panics across the ABI (the reference panics/aborts on errors,
 *     server/Cargo.toml:9 — a drop-in must not).  mi_last_error() gives the
 *     thread-local message of the last failure on the calling thread.
 *   - a handle serialises internally (one mutex per handle, the reference's
 *     model: server/src/main.rs:33-34); distinct handles are independent.
 *   - "host" pointers are ordinary memory; "_device" entry points take HIP
 *     device pointers on the handle's device and a hipStream_t passed as void*
 *     (NULL = the handle's own stream) and do not synchronise the stream.
 *     NB: NULL is NOT "the device's NULL stream": a device buffer that another stream is still writing — the legacy default
 *     stream of a framework included, which is what PyTorch's current stream is unless one is set — must be complete
 *     before it is handed over with NULL, or be handed over with the stream that produces it (a dev tool of this
 *     repository got this wrong for three rounds: tools/knn_prefilter_soak.py).
 *     The work a handle enqueues runs in CALL ORDER whichever streams the calls
 *     name: every entry point first makes its stream wait (an event, no host
 *     block) for what the handle enqueued before.  So embed_device(stream A) ->
 *     append_device(stream A) -> search(stream B) scans the appended rows.  One
 *     exception, on purpose: an append does not wait for searches in flight (it
 *     only writes rows they do not scan), so the ingest stream never stalls on a query.
 *   - there is NO CPU fallback: without a usable gfx950 device, creation fails
 *     with MI_ERR_NO_DEVICE.
 */
#ifndef MI355CLIP_H
#define MI355CLIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_ERR_INVALID (-1)     /* bad argument (null pointer, zero dim, ...) */
#define MI_ERR_IO (-2)          /* weights file missing / unreadable / malformed */
#define MI_ERR_HIP (-3)         /* a HIP runtime call failed */
#define MI_ERR_NO_DEVICE (-4)   /* no gfx950 device with that ordinal */
#define MI_ERR_UNSUPPORTED (-5) /* shape outside what the kernels are built for */
#define MI_ERR_OOM (-6)

#define MI_PRECISION_F32 0  /* fp32 in / fp32 accumulate: the parity path (<= 1e-4 rel) */
#define MI_PRECISION_BF16 1 /* bf16 MFMA operands, fp32 accumulate + fp32 residual stream */
#define MI_PRECISION_BF16_SPLIT 2 /* as BF16, but every LayerNorm output feeds its GEMM as a hi + lo bf16 pair (K = 2D
                                   * against [W | W]): for towers whose LayerNorm outputs carry outlier channels */

#define MI_KNN_NO_ID UINT64_MAX /* id written for missing results (fewer than k rows) */

typedef struct mi_clip mi_clip;         /* a loaded vision (or text) tower on one GPU */
typedef struct mi_knn mi_knn;           /* one row-shard of the embedding table on one GPU */
typedef struct mi_pipeline mi_pipeline; /* scan-loop body + query fused on HIP streams (one GPU) */
typedef struct mi_knn_sharded mi_knn_sharded; /* the table row-sharded over several GPUs, one process */
typedef struct mi_index mi_index;       /* table `image` {id, image_path, embedding}: a shard + the path column */

const char* mi_last_error(void);
/* ABI version of this header (bumped on any signature change). */
int mi_abi_version(void);
/* number of visible HIP devices (0 when there is none; never fails hard). */
int mi_device_count(void);

/* ---------------------------------------------------------------- Seam A: ViT */

/* Replaces clip::clip_vit_large_patch14::Model::from_file(path, &device)
 * (server/src/clip.rs:46-48; weights produced by clip/build.rs:75-83).
 * `weights_path` is a Hugging Face safetensors file holding the
 * CLIPVisionModelWithProjection tensors (vision_model.* + visual_projection.weight,
 * F32/F16/BF16); dimensions are read from the tensor shapes, so any CLIP ViT
 * geometry with head_dim 64 loads (ViT-L/14: 24 x 1024, 16 heads, 257 tokens).
 * The handle is meant to stay resident across scans (the reference reloads
 * 1.16 GB on every scan). */
int mi_clip_load(const char* weights_path, int device, int precision, mi_clip** out);
void mi_clip_free(mi_clip* m);
/* `weights_path` may also be the file the reference's `-w` points at (server/src/server_arguments.rs:8-9): the Burn
 * named-MessagePack record `vision_model.mpk` that burn-import writes at build time (clip/build.rs:75-83).  Its field
 * names are those of the GENERATED module, which is not in the reference tree, so its tensors are mapped to the
 * Hugging Face names by shape, module and graph order (Linear weights transposed from Burn's [in, out]).  Both the fused
 * LayerNorm inventory and the DECOMPOSED one of the opset-16 graph the reference builds (clip/scripts/upgrade_opset.py:9-28:
 * gamma / beta as bare constants, scalar and integer constants beside them) are recognised, with the linears as Linear modules
 * or as bare MatMul + Add constants; leaves that cannot be a tower tensor are set aside and listed; an inventory that fits
 * neither form is refused (MI_ERR_UNSUPPORTED) with the inventory in the message.
 * Untested against a real burn-import file (none exists offline); tools/make_synthetic_mpk.py writes the test files.
 *
 * mi_weights_list: the tensors either kind of file holds, as this library names them, one "name dtype [shape]" line
 * each, into buf (NUL-terminated, truncated to cap); *needed (may be NULL) = bytes for the whole listing.  Needs no GPU. */
int mi_weights_list(const char* weights_path, char* buf, size_t cap, size_t* needed);

/* Run-time options of a loaded handle (the MI_CLIP_* environment variables only seed them at load;
 * nothing on the hot path reads the environment):
 *   "max_batch"  images per internal pass (default 256)
 *   "parts"      1..4 sub-chunks of a pass run as independent streams (default 2; bf16 tower)
 *   "full_last"  1 = also compute the rows of the last layer that never reach the output (default 0:
 *                the pooled output is the CLS row, the result is bit-identical either way)
 *   "front_overlap" 1 = where the library uploads the images itself (mi_pipeline_ingest, mi_clip_embed) a forward's front — the
 *                patch gather and the patch GEMM, the only readers of the uploaded batch — is enqueued on the copy stream right
 *                behind its upload and runs under the PREVIOUS forward's layers (events order it against the embed kernels on
 *                either side).  Same bits.  mi_pipeline_stats' forward time then starts at the embed kernel
 *   "attn_shift" 1 = always take the shifted (exact row maximum) pass of the bf16 attention (default 0: taken
 *                only for queries whose softmax numerators leave the exponent range; same result)
 *   "attn_order" which (image, head) pairs a workgroup of the persistent bf16 attention walks: 1 (default) = the 32 workgroups
 *                that share an XCD start on all 16 heads of two images; 0 = workgroup b starts at pair b (an XCD then sits on
 *                two heads for the whole launch: a quarter of its L2 channels).  Same bits
 *   "qkv_pad"    elements added to the row pitch of the bf16 image tower's q|k|v activations where the persistent attention
 *                runs (default 128 = 256 bytes; a multiple of 64 in 0..1024; 0 = dense 3 D rows, the layout of rounds 1-4:
 *                a head's pieces then fall on few memory channels and attention takes 15-20 % longer).  Same bits
 *   "qkv_layout" how the bf16 image tower keeps q|k|v between the q/k/v GEMM and the persistent attention: 0 = token rows
 *                [M][3 D + qkv_pad]; 1 = head-major planes [3][H][Mp][64] (the GEMM's epilogue stores one contiguous KiB per
 *                wave-instruction, attention fetches a head's K / V / q of one image as one contiguous block).  Same bits
 *   "store_nt"   1 (default) = the persistent GEMM writes q|k|v, h and the deltas — outputs a LATER kernel reads — with the nt cache
 *                policy, so that they do not take the XCDs' L2 lines from the operands the same launch streams: - 0.25 ... - 0.45 ms
 *                per 256-image forward; 0 = default policy (A/B hook).  Same bits
 *   "attn_nt"    1 = the persistent attention fetches K, V and q — each read exactly once, by one CU — with the nt cache policy
 *                (A/B hook, default 0: the kernel alone is 8 % faster with it, the tower 0.4 % slower — its K / V are still warm
 *                from the GEMM that wrote them).  Same bits
 *   "split_tail" 0 = do not cut a short last round of GEMM tiles into quadrant tasks (A/B hook)
 *   "gemm_order" tile order of the persistent GEMM: np > 0 (default 4) = an XCD's concurrent tiles are a (32 / np) x np patch
 *                inside one column group of np weight tiles (which stay in its L2); 0 = row-major.  Same bits; -2.3 % per forward
 *   "im2col_rows" 0 = the patch gather in 4P-byte runs instead of the LDS-staged rows form (A/B hook; same bits)
 *   "ln_nt"      bit 0 = LN1 writes the residual stream back with non-temporal stores, bit 1 = LN1's last-use loads are
 *                non-temporal (A/B hook, default 0; same bits, no measurable effect)
 *   "x24"        1 (default) = the bf16 tower keeps its residual stream as 24-bit floats in two planes (16 significant
 *                bits, 3 bytes per element: -14 % LayerNorm traffic, -1 % per forward, error against fp32 unchanged);
 *                0 = fp32 rows.  Neither changes the fp32 parity path or the text tower
 *   "ln_fold"    1 (default) = the bf16 image tower runs layers 0 .. L-2 WITHOUT LayerNorm kernels: gamma is folded into the
 *                q/k/v and fc1 weights at load, out_proj / fc2 add their output to the residual planes in their own epilogue
 *                and emit per-row sums, q/k/v / fc1 finish the LayerNorm in theirs (rstd * (acc - mean * c) + b').  Needs
 *                hidden and intermediate sizes that are multiples of 256 and at least two layers (MI_ERR_UNSUPPORTED when
 *                set to 1 on another geometry; such handles silently keep the LayerNorm kernels).  0 = LayerNorm kernels
 *                (rounds 1-4).  -5.7 % per forward; error against fp32 unchanged; a different rounding sequence, so NOT the
 *                same bits as 0.  "x24" and "ln_nt" only act on the LayerNorm form.  Both weight forms stay resident
 *                so that the option switches per forward: + 0.35 GB per bf16 ViT-L/14 handle (W_qkv, W_fc1 plain and folded)
 *   "attn_f32_mfma" 0 = the fp32 path's attention as one thread per query (rounds 1-4) instead of the exact-f32 MFMA kernel
 *                (another summation order, both far inside 1e-4; A/B hook)
 *   "text_fast"  0 = a single text query takes the batched kernels instead of the skinny-GEMM path (text handles)
 *   "text_fuse"  0 = on that path, attention and out_proj as two launches with a bf16 delta between them (rounds 2-4);
 *                1 (default) = one launch per layer, the heads' out_proj contributions summed in fp32 by the LayerNorm */
int mi_clip_set_option(mi_clip* m, const char* key, int value);

/* geometry of a loaded model: out[0..7] = image, patch, tokens, hidden, layers,
 * heads, ff, proj */
int mi_clip_info(const mi_clip* m, uint32_t out[8]);

/* The LayerNorm-free layer loop ("ln_fold", the bf16 image tower's default) watches its own precondition.  It feeds the
 * q/k/v and fc1 GEMMs bf16(x) of the UN-normalised residual row, so a row whose mean lies r standard deviations off zero
 * carries about r times the rounding error of the LayerNorm tower (which rounds after subtracting the mean).  Measured on
 * seeded ViT-L/14 with a common offset planted on the stream (tools/bf16_acceptance.py; max error / rms against
 * the fp32 tower, stated bound 3e-2): r = 0, 1: 1.5e-2, where the LayerNorm tower is; r = 4: 2.3e-2; r = 16: 8.8e-2 (out of
 * bound, top-1 agreement 96 % instead of 99 %); r = 64: 0.30.  A single massive channel (100 sigma) is harmless: it moves the
 * deviation, not the mean.
 * Two defences.  (1) At load the library removes the common mode of everything that writes to the residual stream (out_proj
 * and fc2 weights and biases, the pre-LayerNorm's output): every reader of the stream is a LayerNorm, so the function is
 * unchanged and the rows have mean ~ 0 whatever the checkpoint's biases are (MI_CLIP_LN_CENTER=0 at load keeps the weights as
 * read; with it the r = 64 study reads 1.5e-2 again).  (2) Every forward counts, on the device, the live token rows (all
 * layers) with mean^2 > 16 var (r > 4):
 *   out[0] = such rows since load / the last reset,  out[1] = rows looked at.
 * out[0] != 0: set option "ln_fold" to 0 (LayerNorm kernels, +6 % per forward, insensitive to r).
 * Waits for the handle's enqueued forwards.  Handles without the loop (fp32, BF16_SPLIT, text, other geometries) report 0, 0.
 * The reference loads whatever checkpoint -w names (server/src/clip.rs:46-48): this is the check that goes with it. */
int mi_clip_ln_fold_stats(mi_clip* m, uint64_t out[2], int reset);

/* Replaces `model.forward(Tensor::from_data(TensorData::new(buf,[n,3,224,224])))`
 * + `output.to_data()` (server/src/clip.rs:112-124).  Host pointers.
 * nchw: [n,3,H,W] contiguous f32 (H = W = 224 for ViT-L/14), as
 * image_prepare_resnet lays it out (server/src/clip.rs:153-175);
 * out: [n,proj] contiguous f32 (proj = 768), NOT L2-normalised.
 * n = 0 is a successful no-op (the reference calls forward on an empty chunk,
 * server/src/clip.rs:112-118).  Any n: tiled internally. */
int mi_clip_embed(mi_clip* m, const float* nchw, size_t n, float* out);

/* Same computation on device-resident buffers (rows (a3) of SURVEY.md §8:
 * removes the two host copies and the blocking readback of
 * server/src/clip.rs:107-124).  Asynchronous on `stream`. */
int mi_clip_embed_device(mi_clip* m, const float* d_nchw, size_t n, float* d_out, void* stream);

/* image_prepare_resnet's arithmetic (server/src/clip.rs:158-172) fused in front
 * of the tower: rgb8 = [n,H,W,3] interleaved u8 already at model resolution
 * (what `resize_exact(..).to_rgb8().as_raw()` returns), host pointers. */
int mi_clip_embed_rgb8(mi_clip* m, const uint8_t* rgb8, size_t n, float* out);

/* host-only restatement of the same arithmetic for callers that keep the
 * reference's two-step flow: rgb8 [n,H,W,3] -> chw f32 [n,3,H,W]. */
int mi_preprocess_rgb8(const uint8_t* rgb8, size_t n, uint32_t height, uint32_t width, float* chw);

/* The resize in front of that arithmetic: `img.resize_exact(224, 224, FilterType::CatmullRom)`
 * (server/src/clip.rs:154), i.e. image-0.25.8's separable resampler (Cargo.lock:5008-5009;
 * imageops/sample.rs: vertical pass into f32, horizontal pass, clamp, round half away from zero),
 * run on GPU `device`.  rgb8 = [height][width][3] interleaved u8 (a decoded photo; RGBA / grey
 * inputs give the same RGB bytes when expanded first, the crate filters channels independently),
 * out = [new_height][new_width][3].  Equal sizes copy, as the crate does.  Extents up to 32768 and
 * reductions up to 255x; zero extents and sizes beyond those limits return MI_ERR_UNSUPPORTED. */
int mi_resize_catmullrom_rgb8(int device, const uint8_t* rgb8, uint32_t width, uint32_t height, uint32_t new_width,
                              uint32_t new_height, uint8_t* out);

/* image_prepare_resnet whole (server/src/clip.rs:153-175): one decoded RGB8 image of any size ->
 * chw f32 [3][224][224], resize and normalisation fused on the device. */
int mi_image_prepare_resnet(int device, const uint8_t* rgb8, uint32_t width, uint32_t height, float* chw);

/* One chunk of the scan loop (server/src/clip.rs:92-124) in one call: n decoded RGB8 images of
 * any sizes (rgb8[i] = [heights[i]][widths[i]][3], host pointers) -> resize + normalise on the
 * device straight into the tower's input -> out [n,768] f32.  Uploads overlap the resize of the
 * previous image. */
int mi_clip_embed_images(mi_clip* m, const uint8_t* const* rgb8, const uint32_t* widths, const uint32_t* heights,
                         size_t n, float* out);

/* The text tower of the same model (HF `CLIPTextModelWithProjection` tensors in the safetensors
 * file): what `clip(state, text)` gets from embed_anything (server/src/clip.rs:19-23, :35-40) and
 * feeds to the refine step / kNN as the query.  Tokenisation stays with the caller:
 * input_ids = [n][positions] int32 (BOS .. EOS, padded; positions = mi_clip_info()[2], 77 for CLIP),
 * the pooled row is the one holding the largest id (the EOS token), as in OpenAI CLIP / candle.
 * out = [n, 768] f32, not normalised.  MI_PRECISION_F32 (parity, 3.3 ms per query) or MI_PRECISION_BF16 (bf16 MFMA
 * GEMMs and causal bf16 attention: the request path, server/src/clip.rs:19-23 sits in front of every search).
 * The handle is freed with mi_clip_free; the image entry points reject it and vice versa. */
int mi_clip_load_text(const char* weights_path, int device, int precision, mi_clip** out);
int mi_clip_embed_text(mi_clip* m, const int32_t* input_ids, size_t n, float* out);

/* ---------------------------------------------------------------- Seam B: kNN */

/* One shard of table `image{embedding}` (server/src/search.rs:13-18; index DDL
 * server/src/clip.rs:140-143: DIMENSION 768, DIST COSINE, TYPE F32) resident in
 * HBM as row-major f32.  dim must be a multiple of 64.  Row ids are
 * base + insertion ordinal (uint64); base defaults to 0 and is the shard's
 * first global row when the table is row-sharded over several GPUs. */
int mi_knn_create(uint32_t dim, int device, mi_knn** out);
void mi_knn_free(mi_knn* t);
int mi_knn_set_base(mi_knn* t, uint64_t base);
/* Options of a shard.  "prefilter" = 1 or 2: two-stage EXACT search for k <= 4096 on shards of >= 2^18 rows — a mirror of the
 * rows (1: bf16, + 50 % device memory, dim % 128 == 0; 2: bytes with a per-row scale, + 25 %, dim % 256 == 0; built by the
 * next search, kept up to date by every later one) is scanned first, the rows that a rigorous error bound (1: data-
 * independent; 2: per row and query) cannot exclude are re-evaluated from the fp32 rows with the single-pass arithmetic:
 * same ids, same distance bits, a half (1) or a quarter (2) of the bytes per query.  Corpora that put more than 2^22
 * rows inside the bound fall back to the single pass on the device.  0 (default) frees the mirror.
 * "batch_stage1" = 1 (default) / 0: how a GROUP of queries (mi_knn_search with nq >= 2, mi_knn_search_batched_device) runs its
 * shared stage 1 over the byte mirror: 1 on the matrix pipe (any group size up to 16; HBM-bound), 0 on the vector ALU
 * (groups of 8 / 4 / 2; the round-3 form, kept for A/B).  Same answers either way.
 * "prefilter_sample" = 1 (default) / 2 / 0: with the byte mirror and k <= 64 a GROUP of queries takes its collect threshold from the
 * k-th smallest upper bound of a SAMPLE of the stage-1 keys (every 8th tile): a valid, looser threshold — a few times more rows
 * for stage 2, an eighth of the select's reads (16 queries per call: - 11 %).  2: single queries too (measured equal at k = 10,
 * 2 % slower at k = 64).  0: always the threshold over all keys.  Same answers either way.
 * "prefilter_adaptive" = 1 (default) / 0: the two-stage search watches itself — candidate counts and fallbacks are read
 * back asynchronously; after two consecutive fallbacks (more than 2^22 candidates) the next 64 single-query searches run
 * the single pass alone (what such a corpus would pay anyway, without stage 1 on top), then stage 1 is probed again with
 * two queries.  Results never change.  The channel scales of the byte mirror are taken again (and the
 * mirror rebuilt, 10 ms per 10 M rows) when the table has grown 4x since they were taken. */
int mi_knn_set_option(mi_knn* t, const char* key, int value);
/* Of the most recent single-query search of this shard (waits for it): how many rows stage 2 re-evaluated, and whether the
 * single pass had to answer instead (then `candidates` is the count that did not fit).  Both 0 when the search did not
 * go through the prefilter (option off, k > 4096, fewer than 2^18 rows, stage 1 switched off for the moment by the adaptive
 * rule).  Behind a batched call: of the FIRST query of its last group. */
int mi_knn_prefilter_stats(mi_knn* t, uint32_t* candidates, uint32_t* fell_back);
/* The adaptive state (waits for the searches in flight): out = {searches for which stage 1 is still switched off,
 * consecutive fallbacks seen, searches that skipped stage 1 so far, rows the table held when the byte mirror's channel
 * scales were taken}. */
int mi_knn_prefilter_state(mi_knn* t, uint32_t out[4]);
int mi_knn_reserve(mi_knn* t, uint64_t rows); /* capacity hint; keeps contents */
int mi_knn_size(const mi_knn* t, uint64_t* rows);

/* Replaces db.insert("image").content(rows) (server/src/clip.rs:125-137): append n
 * rows of dim f32 (host pointers). */
int mi_knn_append(mi_knn* t, const float* rows, uint64_t n);
/* append rows already on the device (e.g. straight from mi_clip_embed_device) */
int mi_knn_append_device(mi_knn* t, const float* d_rows, uint64_t n, void* stream);
/* append n synthetic rows generated on the device: global rows
 * [first_row, first_row+n) of the seeded corpus of image_search_amd/synth.py */
int mi_knn_append_synthetic(mi_knn* t, uint64_t seed, uint64_t first_row, uint64_t n);
/* copy rows [first, first+n) back to the host (tests, refine's row fetch:
 * server/src/search.rs:43-58) */
int mi_knn_get_rows(mi_knn* t, uint64_t first, uint64_t n, float* out);

/* Persistence of a shard (what the database's storage does for `image.embedding`,
 * server/src/clip.rs:125-137): mi_knn_save writes {32-byte header, rows*dim f32} to `path` (through
 * `path`.tmp + fsync + rename: a crash or a full disk leaves the previous file); mi_knn_load appends a
 * file's rows to the table (an empty table takes the file's id base; a non-empty one accepts only the
 * file that continues its ids).  Streamed in 64 MiB pieces: no host copy of the table on either side. */
int mi_knn_save(mi_knn* t, const char* path);
int mi_knn_load(mi_knn* t, const char* path);

/* Replaces `SELECT id, image_path, vector::distance::knn() FROM image WHERE
 * embedding <|K|> $reference` (server/src/search.rs:70-86; K = 1000 there).
 * For each of nq queries (q: [nq,dim] host f32): the k rows of this shard with the
 * smallest cosine distance 1 - q.x/(|q||x|), sorted by (distance asc, id asc);
 * NaN distances (zero-norm row or query) sort last.  idx: [nq,k] uint64,
 * dist: [nq,k] f32.  Fewer than k rows: the tail is MI_KNN_NO_ID / +inf.
 * Each query is one pass over the table (the reference serves one query per
 * request, server/src/search.rs:20-102). */
int mi_knn_search(mi_knn* t, const float* q, uint32_t nq, uint32_t k, uint64_t* idx, float* dist);
/* Same, queries and results on the device, asynchronous on `stream`.
 * d_idx [nq,k] uint64 and d_dist [nq,k] f32 are what each rank feeds to the
 * all-gather when the table is sharded. */
int mi_knn_search_device(mi_knn* t, const float* d_q, uint32_t nq, uint32_t k, uint64_t* d_idx,
                         float* d_dist, void* stream);
/* Throughput variant: ONE pass over the table serves all nq queries (nq <= 16).  With "prefilter" = 2 and dim 768 the whole
 * group of up to 16 queries shares one pass over the byte mirror on the matrix pipe (the queries cut into three signed 7-bit
 * digits, exact int8 MFMA dot products; option "batch_stage1" = 0: the vector-ALU form, groups of 8 / 4 / 2) and ONE launch
 * of every later kernel of the two-stage search; otherwise passes of 8 / 4 / 2 queries over the fp32 rows (k <= 64).
 * Results identical to nq calls of mi_knn_search_device with nq = 1: same ids, same distance bits.
 * Device memory: a group keeps per-query copies of the two-stage workspaces, allocated by the first group of that size
 * and kept — per query 4 bytes per table row (stage-1 keys: 640 MB for 16 queries over 10 M rows, 6.4 GB over 100 M),
 * 32 MB of candidate rows + keys, and the select / sort buffers; a failed allocation fails that search with MI_ERR_OOM. */
int mi_knn_search_batched_device(mi_knn* t, const float* d_q, uint32_t nq, uint32_t k, uint64_t* d_idx,
                                 float* d_dist, void* stream);

/* ---------------------------------------------- Seam B over several GPUs, ONE process */

/* The reference server is one process holding one database handle (server/src/main.rs:30-35), one search at
 * a time (server/src/search.rs:26).  mi_knn_sharded keeps that shape for a table larger than one GPU:
 * `n_dev` shards on `devices[]` (BASELINE config 5: 8 x 10 M rows) behind one handle.
 *   rows    block-cyclic: global row r (= its id, the insertion ordinal) is in block r / block_rows, blocks are
 *           dealt round-robin to the shards (block_rows = 0 -> 4096; a multiple of 64).  Appends keep ids global.
 *   search  the query goes to every device, the shards scan concurrently on their own streams, the per-shard
 *           top-k lists (12 k bytes per shard and query) are all-gathered — RCCL ncclAllGather over xGMI between
 *           distinct devices; device-to-device / peer copies into the first shard's buffer when a device is listed twice
 *           or librccl is absent — and merged once, on the device.  Result = what ONE mi_knn holding every row returns,
 *           bit for bit.
 * A device may be listed more than once (several shards on one GPU: how a one-GPU box tests n > 1). */
int mi_knn_sharded_create(uint32_t dim, const int* devices, int n_dev, uint32_t block_rows, mi_knn_sharded** out);
void mi_knn_sharded_free(mi_knn_sharded* t);
/* any of the outputs may be NULL; transport: 0 = single shard, 1 = device-to-device / peer copies, 2 = RCCL all-gather
 * (a one-shard table made under MI_KNN_SHARDED_TRANSPORT=rccl reports 2: its one-rank communicator runs the collective) */
int mi_knn_sharded_info(const mi_knn_sharded* t, uint64_t* rows, uint32_t* n_shards, uint32_t* block_rows, int* transport);
/* What the exchange step of server/src/search.rs:70-86's replacement has really executed on this handle so far:
 * out = {searches enqueued, ncclAllGather calls issued (ONE per shard and search: ids and distances travel as one packed
 * record), transport copies issued instead, device merges}. */
int mi_knn_sharded_stats(const mi_knn_sharded* t, uint64_t out[4]);
int mi_knn_sharded_set_option(mi_knn_sharded* t, const char* key, int value); /* mi_knn_set_option on every shard */
int mi_knn_sharded_reserve(mi_knn_sharded* t, uint64_t rows);
int mi_knn_sharded_append(mi_knn_sharded* t, const float* rows, uint64_t n, uint64_t* first_id /* may be NULL */);
/* Rows that are already in device memory, on ANY device of the process (src_device; e.g. straight from
 * mi_clip_embed_device of a replica there): every run goes to its shard by a device-to-device copy (same GPU) or
 * hipMemcpyPeerAsync over xGMI (another GPU), enqueued on `stream` — a stream of src_device, NULL = its null stream — with
 * no trip through the host and no host block; searches enqueued afterwards see the rows.  A failure leaves the table as it was. */
int mi_knn_sharded_append_device(mi_knn_sharded* t, const float* d_rows, uint64_t n, int src_device, void* stream,
                                 uint64_t* first_id /* may be NULL */);
int mi_knn_sharded_append_synthetic(mi_knn_sharded* t, uint64_t seed, uint64_t first_row, uint64_t n);
/* shard s of the table, borrowed (its device, its size, mi_knn_get_rows on local rows, ...); NULL when out of range */
mi_knn* mi_knn_sharded_shard(mi_knn_sharded* t, uint32_t s);
int mi_knn_sharded_get_rows(mi_knn_sharded* t, uint64_t first, uint64_t n, float* out);
int mi_knn_sharded_search(mi_knn_sharded* t, const float* q, uint32_t nq, uint32_t k, uint64_t* idx, float* dist);
/* The same search without the wait: everything (query upload, the shards' scans, exchange, merge, readback) is enqueued
 * and the call returns; idx / dist (caller-owned, must stay valid) are filled when mi_knn_sharded_sync returns, or when 8
 * later searches have been enqueued.  q is copied before the call returns. */
int mi_knn_sharded_search_async(mi_knn_sharded* t, const float* q, uint32_t nq, uint32_t k, uint64_t* idx, float* dist);
int mi_knn_sharded_sync(mi_knn_sharded* t);
/* `<prefix>.g<gen>.<s>of<n>.miknn` per shard + the manifest `<prefix>.shards` (n, block, rows, dim, gen), written last:
 * every save is a new GENERATION of files, the previous one is deleted only after the manifest names the new one, so a
 * crash or an I/O error at any point of a save leaves a complete, loadable table on disk.  load needs an empty table
 * (and leaves it empty on failure) and re-deals the blocks when the shard count or block size differ from the saved ones. */
int mi_knn_sharded_save(mi_knn_sharded* t, const char* prefix);
int mi_knn_sharded_load(mi_knn_sharded* t, const char* prefix);
/* Change the layout of a LIVE table: every row of `src` into the empty `dst` (another shard count, device set or block
 * size), block by block, device to device — a plain copy where source and destination shard share a GPU,
 * hipMemcpyPeerAsync over xGMI where they do not; nothing passes through the host.  src is unchanged. */
int mi_knn_sharded_rebalance(mi_knn_sharded* dst, mi_knn_sharded* src);
/* the placement arithmetic by itself (host-only): global row <-> (shard, local row) */
int mi_knn_sharded_place(uint32_t block_rows, uint32_t n_shards, uint64_t row, uint32_t* shard, uint64_t* local);
int mi_knn_sharded_id(uint32_t block_rows, uint32_t n_shards, uint32_t shard, uint64_t local, uint64_t* row);

/* Deterministic merge of `lists` candidate lists of k (id, dist) entries each —
 * what every rank holds after the all-gather of per-shard results — into the
 * global top-k under the same ordering.  Host-only. */
int mi_knn_merge(const uint64_t* idx_in, const float* dist_in, uint32_t lists, uint32_t k,
                 uint64_t* idx, float* dist);
/* The same merge on the device, asynchronous on `stream`: in = [lists][nq][k] (the rank-major buffer an all-gather of
 * per-shard [nq][k] results leaves on every rank), out = [nq][k].  Each input list must be in result order with its
 * MI_KNN_NO_ID padding at the tail (what every search entry point returns).  Bit-identical to mi_knn_merge. */
int mi_knn_merge_device(int device, const uint64_t* d_idx_in, const float* d_dist_in, uint32_t lists, uint32_t nq, uint32_t k,
                        uint64_t* d_idx, float* d_dist, void* stream);

/* ------------------------------------------- fused flow (BASELINE config 4, one GPU) */

/* The body of the scan loop (server/src/clip.rs:107-137: upload -> forward -> readback -> insert)
 * and the query (server/src/search.rs:70-86) as one pipeline on three HIP streams:
 *   copy stream    upload of chunk i+1 under the forward of chunk i
 *   ingest stream  forward; its last kernel writes the embeddings straight into the table's next
 *                  rows (no readback, no re-upload)
 *   search stream  the scan of a query and the readback of its k results; the host does not
 *                  block on it (results arrive at sync/drain).  Measured: on the device the
 *                  4.5 ms scan does NOT hide under the forward -- with the runtime's default 4
 *                  hardware queues it runs between two forwards, with 8 queues it overlaps and
 *                  slows the forward by as much; the stream buys host-side
 *                  asynchrony, not device time
 * `model` is an image tower, `table` a shard on the same device with dim == the model's output
 * width.  Both are borrowed and must outlive the pipeline; they stay usable through their own
 * entry points (the handles order all work, see Conventions). */
int mi_pipeline_create(mi_clip* model, mi_knn* table, mi_pipeline** out);
/* The same pipeline for ONE process over several GPUs (BASELINE config 5 as the reference's single server would run it:
 * one AppState, scan task and search handler side by side, server/src/main.rs:30-35): models[s] is the tower replica on the
 * device of the table's shard s (n_models == shard count; one handle may serve several shards of its GPU).
 *   ingest  a chunk is cut at the table's block boundaries; every run is uploaded to, and embedded by, the replica on the
 *           GPU that owns its block, whose last kernel writes the rows straight into that shard ("each
 *           replica appends its embeddings to its local shard"); runs of consecutive blocks occupy different GPUs at once,
 *           so a table created with block_rows = the chunk size per GPU keeps every replica busy.
 *   query   mi_knn_sharded_search_async: the shards scan side by side, lists all-gathered and merged on the device.
 * mi_pipeline_ingest / query / sync / drain / stats / free serve both forms (drain delivers everything pending here;
 * mi_pipeline_query_device is for the one-GPU form, whose caller does the exchange). */
int mi_pipeline_create_sharded(mi_clip* const* models, int n_models, mi_knn_sharded* table, mi_pipeline** out);
void mi_pipeline_free(mi_pipeline* p);
/* Enqueue one chunk: nchw = [n,3,H,W] f32, host memory.  The n embeddings become rows
 * [size, size+n) of the table; *first_id (may be NULL) = id of the first one.  Returns when the
 * chunk is queued (at most two chunks are in flight).  With pinned memory (mi_host_alloc) the upload
 * is asynchronous and the buffer may be refilled once the NEXT mi_pipeline_ingest (or
 * mi_pipeline_sync) has returned; ordinary memory is staged before the call returns.
 * n = 0 is a successful no-op, as in mi_clip_embed. */
int mi_pipeline_ingest(mi_pipeline* p, const float* nchw, size_t n, uint64_t* first_id);
/* Enqueue one query (q: [dim] f32 host, copied before the call returns) over every row ingested by
 * the calls made before this one.  idx [k] / dist [k] (host, caller-owned) are filled when
 * mi_pipeline_sync returns (or when 16 later queries have been enqueued).  Same results and
 * ordering as mi_knn_search. */
int mi_pipeline_query(mi_pipeline* p, const float* q, uint32_t k, uint64_t* idx, float* dist);
/* As mi_pipeline_query, the k results left on the device (d_idx [k] uint64, d_dist [k] f32, caller-owned, same device):
 * the per-shard list of a row-sharded table, ready for the all-gather without a trip through the host.  The scan runs
 * on the pipeline's search stream; `consumer_stream` (a hipStream_t; NULL = the device's default stream, e.g. what
 * torch.cuda.current_stream() is unless the caller changed it) is made to wait for it — an event, no host block — so work
 * enqueued on that stream afterwards (the collective) sees the results. */
int mi_pipeline_query_device(mi_pipeline* p, const float* q, uint32_t k, uint64_t* d_idx, float* d_dist, void* consumer_stream);
/* Wait for everything enqueued and deliver the pending query results. */
int mi_pipeline_sync(mi_pipeline* p);
/* Deliver finished queries, oldest first, until at most `leave_pending` are still pending (blocks
 * for those it delivers; the ingest stream is not waited for).  With leave_pending = 1 a caller gets
 * the results of query i-1 while query i scans: the per-shard lists a multi-GPU caller all-gathers. */
int mi_pipeline_drain(mi_pipeline* p, uint32_t leave_pending);
/* Device time spent in the pipeline's forwards and scans as measured by events on their own streams,
 * folded in at sync: out = {forwards, total ms of forwards, scans, total ms of scans}; reset != 0 clears. */
int mi_pipeline_stats(mi_pipeline* p, double out[4], int reset);

/* Page-locked host memory for upload buffers (hipHostMalloc). */
int mi_host_alloc(size_t bytes, void** out);
void mi_host_free(void* p);

/* ------------------------------------------- table `image` with its image_path column */

/* The statements the reference server issues against `image` {id, image_path, embedding} (server/src/search.rs:13-18),
 * so that the Rust side needs nothing between its handlers and this library:
 *   mi_index_existing   SELECT image_path FROM image WHERE image_path IN $paths              server/src/clip.rs:74-83
 *   mi_index_insert     db.insert("image").content(rows)                                     server/src/clip.rs:125-137
 *   mi_index_rows_of    SELECT id .. FROM image WHERE image_path IN $paths (then mi_knn_get_rows on mi_index_table)
 *                                                                                            server/src/search.rs:43-58
 *   mi_index_search     the refine step + `embedding <|K|> $reference`                       server/src/search.rs:20-110
 * Row id = insertion ordinal.  No uniqueness constraint on image_path (the reference's table has none; its scan loop
 * filters with the first statement): a path may own several rows.  media_dir: requests name files "media/<rel>", rows
 * store media_dir + <rel> (server/src/search.rs:35-40, :104-109); "" disables the mapping. */
int mi_index_create(uint32_t dim, int device, const char* media_dir, mi_index** out);
void mi_index_free(mi_index* ix);
mi_knn* mi_index_table(mi_index* ix); /* the embedding shard, borrowed (mi_pipeline_create, mi_knn_get_rows, ...) */
int mi_index_size(mi_index* ix, uint64_t* rows);
int mi_index_media_dir(mi_index* ix, char* buf, size_t cap, size_t* needed); /* as given at creation, or as loaded */
int mi_index_existing(mi_index* ix, const char* const* paths, size_t n, uint8_t* exists /* [n]: 1 = has a row */);
int mi_index_insert(mi_index* ix, const char* const* paths, const float* embeddings, size_t n, uint64_t* first_id);
/* paths for the n rows mi_pipeline_ingest has just written into mi_index_table (embeddings never left the device) */
int mi_index_adopt(mi_index* ix, const char* const* paths, size_t n);
/* ids of every row of the given paths, ascending and unique (table order: average_slices adds in input order);
 * *count = how many there are, at most `cap` are written */
int mi_index_rows_of(mi_index* ix, const char* const* paths, size_t n, uint64_t* ids, size_t cap, size_t* count);
/* image_path of row `id`; web != 0: as sent to the client, relative to "media/" (server/src/search.rs:104-109) */
int mi_index_path(mi_index* ix, uint64_t id, int web, char* buf, size_t cap, size_t* needed);
/* web_search_text behind the text tower: referenced_images are the client's "media/.." names; those found in the table
 * refine the query; idx/dist [k] as mi_knn_search; *n_found (may be NULL) = results before the MI_KNN_NO_ID padding */
int mi_index_search(mi_index* ix, const float* text_embedding, const char* const* referenced_images, size_t n_ref, uint32_t k,
                    uint64_t* idx, float* dist, uint32_t* n_found);
/* `<dir>/embedding.miknn` + `<dir>/image_path.bin`, each through a temporary file, fsync and rename, the path file
 * last: after a crash the directory holds a consistent index (at worst the one before the save). */
int mi_index_save(mi_index* ix, const char* dir);
int mi_index_load(mi_index* ix, const char* dir); /* into an empty index */

/* ------------------------------------------------------------ query refinement */

/* fn average_slices(vectors: &Vec<&Vec<f32>>) -> Vec<f32> (server/src/search.rs:127-150):
 * zero-init, add in input order, divide by (m as f32).  m = 0 -> MI_ERR_INVALID
 * (the reference asserts "Input must not be empty"). */
int mi_average_slices(const float* const* vectors, size_t m, size_t len, float* out);
/* the refine step of web_search_text (server/src/search.rs:28, :60-67):
 * m = 0 -> out = text; else out = average_slices([average_slices(selected), text]). */
int mi_refine(const float* text, const float* const* selected, size_t m, size_t len, float* out);

#ifdef __cplusplus
}
#endif
#endif /* MI355CLIP_H */
