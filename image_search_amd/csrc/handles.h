// handles.h — the opaque handles of include/mi355clip.h as the library's translation units see them,
// plus the few host-side routines that cross files (the fused pipeline and the sharded table are built
// from the same forward / scan launchers as the single-handle entry points).
#pragma once
#include <mutex>
#include <vector>

#include "common.h"

namespace mi {
typedef unsigned short bf16_t;

struct Layer {
    float *ln1w, *ln1b, *ln2w, *ln2b;
    void *wqkv, *wo, *w1, *w2;  // T [N][K]
    float *bqkv, *bo, *b1, *b2;
    // the LayerNorm in front of q/k/v and of fc1 folded into the linear (bf16 image tower, option "ln_fold"):
    // W' = bf16(W diag(gamma)), c[n] = sum_k W'[n][k], b' = W beta + b
    void *wqkv_f = nullptr, *w1_f = nullptr;
    float *cqkv = nullptr, *bqkv_f = nullptr, *c1 = nullptr, *b1_f = nullptr;
};


}  // namespace mi

struct mi_clip {
    int device = 0, precision = 0;
    int image = 0, patch = 0, grid = 0, S = 0, D = 0, L = 0, H = 0, FF = 0, E = 0, Kp = 0;
    float eps = 1e-5f;
    std::vector<void*> allocs;
    float *cls = nullptr, *pos = nullptr, *pre_w = nullptr, *pre_b = nullptr, *post_w = nullptr, *post_b = nullptr,
          *proj = nullptr;
    void* wpatch = nullptr;  // T [D][Kp]
    std::vector<mi::Layer> layers;
    // workspace for `cap` images
    size_t cap = 0;
    std::vector<void*> ws;
    float *d_in = nullptr, *d_in2 = nullptr, *d_out = nullptr, *d_out2 = nullptr;
    // activations: set 0 serves a whole chunk; set 1 exists so that two half-chunks can run as two
    // independent streams (see forward()).
    struct Act {
        float *patch = nullptr, *x = nullptr;
        void *col = nullptr, *y = nullptr, *qkv = nullptr, *h = nullptr;
        mi::bf16_t *delta = nullptr, *delta2 = nullptr;  // bf16 path: out_proj / fc2 outputs, added to x by LayerNorm
        // last layer, CLS rows only (n rows, padded to 256): context, residual, LN output, MLP hidden, deltas
        void *c_ctx = nullptr, *c_y = nullptr, *c_h = nullptr;
        float* c_x = nullptr;
        mi::bf16_t *c_d1 = nullptr, *c_d2 = nullptr;
        // ln_fold: per-row partial sums written by the out_proj / fc2 epilogues [M][D / 32][2] and what the q/k/v / fc1
        // epilogues read, {rstd, -mean * rstd} [M][2]
        float *part = nullptr, *stats = nullptr;
        float* c_stats = nullptr;   // ... of the CLS rows (last layer: the query columns are computed for those rows only)
    } act[4];
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    // a forward's front (patch gather + patch GEMM) may run on a stream of its own, under the PREVIOUS forward's layers
    // (forward(..., front)): ev_embed[p] = the last embed_ln of activation set p has read `patch` (the next front may
    // overwrite col / patch), ev_front[p] = the front of set p has written `patch` (embed_ln may read it)
    hipEvent_t ev_embed[4] = {nullptr, nullptr, nullptr, nullptr}, ev_front[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_embed_set[4] = {false, false, false, false};
    bool x24 = true;          // bf16 image tower: the residual stream as 24-bit floats in two planes (3 bytes per element instead of 4; option "x24", MI_CLIP_X24)
    bool ln_fold = true;      // bf16 image tower without LayerNorm kernels in the layer loop where the geometry allows (option "ln_fold", MI_CLIP_LN_FOLD; forward() in vit.hip)
    bool fold_ready = false;  // the folded weights were built at load (geometry allows it)
    bool ln_center = true;    // fold_ready handles: the common mode of everything written to the residual stream removed at load / in embed_ln (vit.hip: center_writer; MI_CLIP_LN_CENTER)
    // ln_fold's watch on its own precondition (mi_clip_ln_fold_stats): live rows of the residual stream whose mean lies more
    // than 4 standard deviations off zero (mean^2 > 16 var) — there bf16(x), taken BEFORE the mean is subtracted, starts to
    // lose the bits the LayerNorm tower keeps.  Counted on the device by embed_ln_kernel / ln_stats_kernel (an atomic only when a
    // row trips), rows looked at counted on the host.
    unsigned long long* d_fold_offset_rows = nullptr;
    uint64_t fold_rows_checked = 0;
    int parts = 2;  // MI_CLIP_PARTS: sub-chunks run as independent streams
    int n_cu = 256;
    uint8_t* d_rgb = nullptr;
    // text tower (mi_clip_load_text): token table, device copies of the ids and of the EOS rows
    bool text = false;
    int vocab = 0;
    float* tok = nullptr;
    int *d_ids = nullptr, *d_rows = nullptr;
    size_t text_cap = 0;
    // one text query as a captured hipGraph (forward_text_one is ~90 short kernels; launched one by one the host's launch
    // calls cost more than the kernels): 0 = not yet run, 1 = ran once eagerly (function attributes are set), 2 = captured
    int* h_ids_pin = nullptr;      // pinned staging of one query's ids / its embedding: the graph copies from / to them
    float* h_out_pin = nullptr;
    int text_graph_state = 0;
    hipGraph_t text_graph = nullptr;
    hipGraphExec_t text_graph_exec = nullptr;
    // mi_clip_embed_images: two upload buffers for decoded images, the resize intermediate, a copy stream
    uint8_t* d_img_src[2] = {nullptr, nullptr};
    float* d_img_tmp = nullptr;
    size_t img_src_cap = 0, img_tmp_cap = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_used[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    size_t max_batch = 256;
    // options (mi_clip_set_option; the MI_CLIP_* / MI_GEMM_* environment variables only seed them at load)
    bool split_ln = false;    // MI_PRECISION_BF16_SPLIT: LayerNorm outputs as hi + lo bf16 pairs, q/k/v and fc1 run over K = 2D
    int attn_ver = 2;         // bf16 attention for 64 < S <= 288: 2 = 32-query tiles (attn32_kernels.h), 1 = 16-query tiles
    bool q_prescaled = false; // log2(e)/8 folded into W_q / b_q at load (attn_ver 2 in the tower)
    bool attn_shift = false;  // force the shifted (exact maximum) pass of attn32 — test hook
    int qkv_pad = 128;        // elements added to the image tower's qkv row pitch where attn32 runs (vit.hip: qkv_pitch)
    int qkv_layout = 0;       // 0 = token rows [M][3D + pad]; 1 = head-major planes [3][H][Mp][64] (vit.hip: qkv_head_major)
    bool front_overlap = false; // mi_pipeline_ingest / mi_clip_embed: a forward's patch gather + patch GEMM on the copy stream, under the previous forward (option "front_overlap")
    bool store_nt = true;     // persistent GEMM: q|k|v / h / delta stores with the nt cache policy (outputs a later kernel reads; -0.25..-0.45 ms per forward); option "store_nt"
    bool attn_nt = false;     // attn32: K / V LDS-DMA and query loads with the nt cache policy (read once by one CU); option "attn_nt": -8 % alone, +0.15 ms in the tower (its K / V are warm from the GEMM that wrote them): off
    int attn_order = 1;       // attn32: first pair of workgroup b (0 = b; 1 = transposed, an XCD's workgroups spread over all heads)
    bool full_last = false;   // compute the dead rows of the last layer too (A/B against the reference graph)
    bool split_tail = true;   // cut a short last round of GEMM tiles into quadrant tasks
    int gemm_order = 4;       // persistent GEMM tile order: 0 = row-major; np > 0 = column groups of np weight tiles, an XCD's
                              // concurrent tiles a (32 / np) x np patch (N / 256 > np and divisible by it, else row-major)
    bool text_fast = true;    // one text query (n == 1, CLIP-L text geometry, bf16): the skinny-GEMM path (vit.hip forward_text_one)
    bool attn_f32_mfma = true; // fp32 attention for S <= 272 on the matrix pipe (attn_f32_mfma_kernel); 0 = one thread per query (A/B; option "attn_f32_mfma")
    bool text_fuse = true;    // ... with out_proj inside the attention launch (text_attn_out_kernel; option "text_fuse")
    int ln_nt = 0;            // A/B hook (MI_CLIP_LN_NT / option "ln_nt"): bit 0 = LN1 writes the residual stream back non-temporally, bit 1 = LN1's last-use loads non-temporal; measured within noise, off
    bool im2col_rows = true;  // bf16 tower: the LDS-staged patch gather (im2col_rows_kernel); 0 = the 4P-byte-run form (A/B)
    mi::WorkOrder order;          // serialises this handle's enqueued work across caller streams
    std::mutex mu;
};


struct mi_knn {
    int device = 0;
    uint32_t dim = 0;
    uint64_t base = 0, rows = 0, cap = 0;
    // a shard of a block-cyclic table (mi_knn_sharded): ids = base + ((local / block) * n + rank) * block + local % block
    uint32_t cyc_block = 0, cyc_n = 0, cyc_rank = 0;
    float* table = nullptr;
    hipStream_t stream = nullptr;
    int n_cu = 0;
    // search workspace
    float* d_q = nullptr;        // [16][dim]
    uint64_t* d_cand = nullptr;  // per-wave lists
    uint64_t* d_tmp = nullptr;   // merge level output
    uint64_t* d_keys = nullptr;  // final keys (k rounded up to 1024 multiples)
    uint64_t* d_idx = nullptr;
    float* d_dist = nullptr;
    size_t cand_keys = 0, tmp_keys = 0, keys_cap = 0, idx_cap = 0, dist_cap = 0;
    bool select_path = true;       // 64 < k <= 4096 by radix select over all keys (MI_KNN_SELECT=0: the per-wave LDS lists)
    uint32_t* d_keys32 = nullptr;  // one distance key per row (selection path, 64 < k <= 4096)
    uint32_t* d_sel = nullptr;     // 6 x 2048 histogram bins + the collect counter
    size_t keys32_cap = 0, sel_cap = 0;
    // two-stage exact search (mi_knn_set_option "prefilter"): a bf16 copy of the rows + their squared norms, kept up to
    // `mirror_rows` and caught up by the next search; candidate rows / keys of stage 2
    int prefilter = 0;              // 0 off, 1 bf16 mirror, 2 byte mirror
    int coarse_ring = 8;            // rows in flight + 1 per lane group in the byte stage-1 scan (MI_KNN_RING: A/B)
    bool last_prefiltered = false;  // the most recent single-query search went through the two stages
    uint16_t* d_mirror = nullptr;   // bf16 rows (prefilter 1) or byte rows (prefilter 2: dim bytes per row)
    float* d_scale8 = nullptr;      // prefilter 2: per-row scale, per-row bound factor, rho of the query in flight
    float* d_cfac8 = nullptr;
    float* d_rho8 = nullptr;
    float* d_g8 = nullptr;          // prefilter 2: per-dimension scale [dim] (+ [dim] accumulators), fixed once the first rows are mirrored
    bool g8_ready = false;
    size_t scale8_cap = 0, cfac8_cap = 0, rho8_cap = 0, g8_cap = 0;
    float* d_xx = nullptr;
    bool batch_stage1_mfma = true;  // the shared stage 1 of a group of queries on the matrix pipe (option "batch_stage1")
    int8_t* d_digits = nullptr;     // [KNN_GROUP_MAX = 16][3][dim]: the queries of a group as three signed 7-bit digits
    float* d_qs = nullptr;          // [KNN_GROUP_MAX][4]: {digit scale S, |q|, rho, -}
    size_t digits_cap = 0, qs_cap = 0;
    int pref_sample = 1;            // k <= 64 over the byte mirror: the collect threshold from a sample of the stage-1 keys — 1: for groups of queries, 2: for single queries too, 0: never (option "prefilter_sample")
    uint32_t* d_skeys = nullptr;    // [queries of a group][sample rows]: the keys of every 8th tile, compact
    size_t skeys_cap = 0;
    uint64_t mirror_rows = 0;
    size_t mirror_cap = 0, xx_cap = 0;
    uint64_t g8_rows = 0;              // rows the table held when the channel scales were taken (refreshed at 4x)
    // Feedback of the two-stage search, read back asynchronously (a query or two late, never a host block): after two
    // consecutive fallbacks stage 1 is skipped for the next PREF_SKIP single-query searches and then probed again — a corpus the mirror cannot thin out pays the single
    // pass, not stage 1 on top of it ("prefilter_adaptive" = 0 turns this off).
    static constexpr int PREF_RING = 8, PREF_SKIP = 64;
    bool pref_adaptive = true;
    uint32_t* h_pref_ring = nullptr;   // pinned [PREF_RING][2]: {candidates, fell back}
    hipEvent_t pref_ev[PREF_RING] = {};
    bool pref_ev_pending[PREF_RING] = {};
    uint32_t pref_consec = 0, pref_skip_left = 0;
    // behind a skip window exactly two probes go out; until both have reported, later queries keep to the single pass
    // (a caller that enqueues faster than the device answers must not queue dozens of probes behind the first two)
    bool pref_probing = false;
    uint32_t pref_probes_left = 0, pref_reports_due = 0;
    uint64_t pref_seq = 0, pref_skipped = 0;
    uint32_t* d_pref_rows = nullptr;   // [2 * PREF_CAP]: candidate rows, then their exact distance keys
    uint64_t* d_pref_keys = nullptr;   // [4096]: the k best keys of stage 2
    uint32_t* d_pref_flag = nullptr;   // {candidate count, fallback, go}
    size_t pref_rows_cap = 0, pref_keys_cap = 0, pref_flag_cap = 0;
    // Order across caller streams.  `writes`: the last append (a search must see every row counted in
    // `rows`).  `reads`: the last search (searches share the workspace above, and a reallocation of the
    // table must wait for them).  An append only ever writes rows beyond `rows`, so it does not wait
    // for searches in flight: ingest and query streams overlap.
    mi::WorkOrder writes, reads;
    std::mutex mu;
};


// The table row-sharded over several GPUs inside one process (sharded.hip).  Global row r lives in block r / block, block b
// on shard b % n at local block b / n; every shard is an ordinary mi_knn whose kernels emit global ids (IdMap).
typedef struct ncclComm* ncclComm_t;
namespace mi {
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t b) {  // caller has the device selected and nothing in flight that uses the old buffer
        if (b <= cap) return;
        if (p) HIP_CHECK(hipFree(p));
        p = nullptr; cap = 0;
        HIP_CHECK(hipMalloc(&p, b));
        cap = b;
    }
};
struct PinnedBuf2 {
    void* p = nullptr;
    size_t cap = 0;
    void reserve(size_t b) {
        if (b <= cap) return;
        if (p) HIP_CHECK(hipHostFree(p));
        p = nullptr; cap = 0;
        HIP_CHECK(hipHostMalloc(&p, b, hipHostMallocPortable));  // read and written by streams of several devices
        cap = b;
    }
};
// one search in flight on a sharded table: per-shard query / result buffers, the gathered lists and the merged result
// on the first shard's device, pinned copies for the host, and where the caller wants them
struct ShardedSlot {
    // a shard's answer is one packed record [nq*k x u64 id | nq*k x f32 distance] (padded to 16 bytes)
    std::vector<DevBuf> d_q, d_rec;           // [shard]
    std::vector<DevBuf> g_rec;                // [shard] (RCCL receive side: n records) or [0] only (copy transport)
    DevBuf m_rec;                             // merged record, on shard 0's device
    PinnedBuf2 h_q, h_rec;
    std::vector<hipEvent_t> ev;               // [shard]: list of shard s is in place
    hipEvent_t done = nullptr;
    uint32_t nq = 0, k = 0;
    uint64_t* user_idx = nullptr;
    float* user_dist = nullptr;
    bool busy = false;
};
}  // namespace mi

struct mi_knn_sharded {
    static constexpr int N_SLOTS = 8;
    uint32_t dim = 0, block = 0;
    uint64_t rows = 0;
    std::vector<int> devices;
    std::vector<mi_knn*> shard;
    mi::ShardedSlot slots[N_SLOTS];
    int next_slot = 0;
    std::vector<ncclComm_t> comms;
    bool use_rccl = false;
    struct { uint64_t searches = 0, collectives = 0, copies = 0, merges = 0; } stats;   // mi_knn_sharded_stats
    uint64_t generation = 0;                  // of the files last saved or loaded (mi_knn_sharded_save)
    std::vector<hipEvent_t> ev_src;           // [device ordinal]: the peer copies of the last append_device from that device
    std::mutex mu;
    uint32_t n() const { return (uint32_t)shard.size(); }
};

namespace mi {
// vit.hip
hipStream_t clip_own_stream(mi_clip* m);
void clip_ensure_workspace(mi_clip* m, size_t n);   // frees and reallocates: waits for the handle's pending work
void clip_ensure_copy_stream(mi_clip* m);
void clip_forward(mi_clip* m, const float* d_img, size_t n, float* d_out, hipStream_t s, hipStream_t front = nullptr);
// knn.hip
hipStream_t knn_own_stream(mi_knn* t);
void knn_grow(mi_knn* t, uint64_t want_rows);       // may reallocate: waits for the handle's pending work
void knn_search_one(mi_knn* t, const float* d_q, uint32_t k, uint64_t* d_idx, float* d_dist, hipStream_t s);
void knn_search_many(mi_knn* t, const float* d_q, uint32_t nq, uint32_t k, uint64_t* d_idx, float* d_dist, hipStream_t s);  // nq queries in groups that share their passes; results [nq][k]
void knn_truncate(mi_knn* t, uint64_t rows);        // forget the rows behind `rows` (a failed multi-shard append / load rolls back)
// list l of query u: ids at d_idx_in + l * idx_stride + u * k, distances at d_dist_in + l * dist_stride + u * k (elements)
void knn_merge_lists_device(const uint64_t* d_idx_in, const float* d_dist_in, uint32_t lists, uint32_t nq, uint32_t k,
                            size_t idx_stride, size_t dist_stride, uint64_t* d_idx, float* d_dist, hipStream_t s);   // caller has the device selected
// sharded.hip
void sharded_place(const mi_knn_sharded* t, uint64_t r, uint32_t* s, uint64_t* local);
uint64_t sharded_rows_of(const mi_knn_sharded* t, uint64_t total, uint32_t s);  // rows shard s holds when the table holds `total`
void sharded_search_enqueue(mi_knn_sharded* t, const float* q, uint32_t nq, uint32_t k, uint64_t* idx, float* dist);  // t->mu held
void sharded_deliver_all(mi_knn_sharded* t);                                    // t->mu held
}  // namespace mi
