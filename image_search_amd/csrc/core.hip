// core.hip — error state, device selection and the host-only entry points
// (query refinement, preprocessing arithmetic, candidate merge) of libmi355clip.so.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.h"

namespace mi {

static thread_local std::string g_last_error;

void set_last_error(const std::string& m) { g_last_error = m; }

void use_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        fail(MI_ERR_NO_DEVICE, "no HIP device visible (%s): libmi355clip has no CPU fallback",
             e == hipSuccess ? "count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) fail(MI_ERR_NO_DEVICE, "device %d out of range (0..%d)", device, n - 1);
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        fail(MI_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 code objects only", device,
             prop.gcnArchName);
    HIP_CHECK(hipSetDevice(device));
}

}  // namespace mi

using namespace mi;

namespace {

// ordering of results: (distance asc, id asc), NaN last — the same 32-bit monotone
// image of the distance the kernels use (knn_kernels.h dist_to_u32).
inline uint32_t dist_key(float d) {
    uint32_t b;
    std::memcpy(&b, &d, 4);
    if (d != d) return 0xFFFFFFFFu;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

}  // namespace

namespace mi {
// global top-k of `lists` candidate lists of k (id, distance) entries under (distance asc, id asc), NaN last
void merge_lists(const uint64_t* idx_in, const float* dist_in, uint32_t lists, uint32_t k, uint64_t* idx, float* dist) {
    struct Ent { uint32_t key; uint64_t id; float d; };
    std::vector<Ent> all;
    all.reserve((size_t)lists * k);
    for (size_t i = 0; i < (size_t)lists * k; ++i)
        if (idx_in[i] != MI_KNN_NO_ID) all.push_back({dist_key(dist_in[i]), idx_in[i], dist_in[i]});
    const size_t keep = std::min<size_t>(k, all.size());
    std::partial_sort(all.begin(), all.begin() + keep, all.end(), [](const Ent& a, const Ent& b) {
        return a.key < b.key || (a.key == b.key && a.id < b.id);
    });
    for (uint32_t i = 0; i < k; ++i) {
        if (i < keep) { idx[i] = all[i].id; dist[i] = all[i].d; }
        else { idx[i] = MI_KNN_NO_ID; dist[i] = INFINITY; }
    }
}
}  // namespace mi

extern "C" {

const char* mi_last_error(void) { return g_last_error.c_str(); }

int mi_abi_version(void) { return 4; }

int mi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// fn average_slices — server/src/search.rs:127-150
int mi_average_slices(const float* const* vectors, size_t m, size_t len, float* out) {
    return guarded([&] {
        if (m == 0) fail(MI_ERR_INVALID, "Input must not be empty");  // the reference's assert message
        if (!vectors || (!out && len)) fail(MI_ERR_INVALID, "null argument");
        for (size_t v = 0; v < m; ++v)
            if (!vectors[v] && len) fail(MI_ERR_INVALID, "vector %zu is null", v);
        for (size_t i = 0; i < len; ++i) out[i] = 0.0f;
        for (size_t v = 0; v < m; ++v) {
            const float* x = vectors[v];
            for (size_t i = 0; i < len; ++i) out[i] += x[i];
        }
        const float count = (float)m;
        for (size_t i = 0; i < len; ++i) out[i] /= count;
    });
}

// refine step of web_search_text — server/src/search.rs:28, :60-67
int mi_refine(const float* text, const float* const* selected, size_t m, size_t len, float* out) {
    return guarded([&] {
        if ((!text || !out) && len) fail(MI_ERR_INVALID, "null argument");
        if (m == 0) {
            if (len) std::memmove(out, text, len * sizeof(float));
            return;
        }
        std::vector<float> sel(len);
        int rc = mi_average_slices(selected, m, len, sel.data());
        if (rc != MI_OK) fail(rc, "%s", mi_last_error());
        const float* two[2] = {sel.data(), text};
        std::vector<float> res(len);
        rc = mi_average_slices(two, 2, len, res.data());
        if (rc != MI_OK) fail(rc, "%s", mi_last_error());
        if (len) std::memcpy(out, res.data(), len * sizeof(float));
    });
}

// image_prepare_resnet's arithmetic — server/src/clip.rs:158-172
int mi_preprocess_rgb8(const uint8_t* rgb8, size_t n, uint32_t height, uint32_t width, float* chw) {
    return guarded([&] {
        if (n == 0) return;
        if (!rgb8 || !chw) fail(MI_ERR_INVALID, "null argument");
        const float mean[3] = {0.485f, 0.456f, 0.406f};
        const float sd[3] = {0.229f, 0.224f, 0.225f};
        const size_t P = (size_t)height * width;
        for (size_t im = 0; im < n; ++im) {
            const uint8_t* src = rgb8 + im * P * 3;
            float* dst = chw + im * P * 3;
            for (size_t i = 0; i < P; ++i)
                for (int c = 0; c < 3; ++c) {
                    const float v = (float)src[i * 3 + c] / 255.0f;
                    dst[(size_t)c * P + i] = (v - mean[c]) / sd[c];
                }
        }
    });
}

int mi_knn_merge(const uint64_t* idx_in, const float* dist_in, uint32_t lists, uint32_t k, uint64_t* idx,
                 float* dist) {
    return guarded([&] {
        if (k == 0) fail(MI_ERR_INVALID, "k must be >= 1");
        if (!idx || !dist || (lists && (!idx_in || !dist_in))) fail(MI_ERR_INVALID, "null argument");
        merge_lists(idx_in, dist_in, lists, k, idx, dist);
    });
}

// Block-cyclic placement of mi_knn_sharded (host-only arithmetic, also what the shards' kernels apply):
// global row -> (shard, local row) and back.
int mi_knn_sharded_place(uint32_t block_rows, uint32_t n_shards, uint64_t row, uint32_t* shard, uint64_t* local) {
    return guarded([&] {
        if (block_rows == 0 || n_shards == 0 || !shard || !local) fail(MI_ERR_INVALID, "bad argument");
        const uint64_t blk = row / block_rows;
        *shard = (uint32_t)(blk % n_shards);
        *local = (blk / n_shards) * block_rows + row % block_rows;
    });
}

int mi_knn_sharded_id(uint32_t block_rows, uint32_t n_shards, uint32_t shard, uint64_t local, uint64_t* row) {
    return guarded([&] {
        if (block_rows == 0 || n_shards == 0 || shard >= n_shards || !row) fail(MI_ERR_INVALID, "bad argument");
        *row = ((local / block_rows) * n_shards + shard) * block_rows + local % block_rows;
    });
}

}  // extern "C"
