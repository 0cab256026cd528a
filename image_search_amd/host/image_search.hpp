// image_search.hpp — C++ host-side mirror of the reference's interface for the hot path, on top of
// the C ABI (include/mi355clip.h).  The reference is Rust (no toolchain in this image), so the
// compiled-language host layer is C++: same names, argument meaning and error behaviour as
//   clip::clip_vit_large_patch14::Model::{from_file, forward}   clip/src/lib.rs:2-7, server/src/clip.rs:46-48,:118
//   average_slices                                              server/src/search.rs:127-150
//   image_prepare_resnet (arithmetic)                           server/src/clip.rs:158-172
//   the `embedding <|K|> $reference` statement                  server/src/search.rs:70-86
// Where the reference panics (assert!/unwrap) this throws std::runtime_error with the same message.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi355clip.h"

namespace image_search {

inline void check(int rc) {
    if (rc != MI_OK) throw std::runtime_error(std::string("mi355clip: ") + mi_last_error());
}

// fn average_slices(vectors: &Vec<&Vec<f32>>) -> Vec<f32>
inline std::vector<float> average_slices(const std::vector<const std::vector<float>*>& vectors) {
    if (vectors.empty()) throw std::runtime_error("Input must not be empty");
    const size_t len = vectors[0]->size();
    std::vector<const float*> ptrs;
    for (auto v : vectors) {
        if (v->size() != len) throw std::runtime_error("All vectors must have the same length");
        ptrs.push_back(v->data());
    }
    std::vector<float> out(len);
    check(mi_average_slices(ptrs.data(), ptrs.size(), len, out.data()));
    return out;
}

// the extension allow-list of the ingest walk (server/src/clip.rs:60-66, test_matches :181-233)
inline bool is_image_path(const std::string& path) {
    const size_t slash = path.find_last_of("/\\");
    const std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
    const size_t dot = name.rfind('.');
    if (dot == std::string::npos || dot == 0) return false;
    std::string ext = name.substr(dot + 1);
    for (auto& c : ext) c = (char)std::tolower((unsigned char)c);
    for (const char* e : {"jpg", "jpeg", "png", "gif", "bmp", "webp", "tiff"})
        if (ext == e) return true;
    return false;
}

// fn image_prepare_resnet(img) -> Vec<f32>, minus the resize: RGB8 HWC (224x224) -> CHW f32
inline std::vector<float> image_prepare_resnet(const std::vector<uint8_t>& rgb8, uint32_t h = 224, uint32_t w = 224) {
    if (rgb8.size() != (size_t)h * w * 3) throw std::runtime_error("rgb8 buffer is not h*w*3 bytes");
    std::vector<float> out((size_t)h * w * 3);
    check(mi_preprocess_rgb8(rgb8.data(), 1, h, w, out.data()));
    return out;
}

// image_prepare_resnet whole (clip.rs:153-175): one decoded RGB8 image of any size -> CHW f32 [3][224][224]
inline std::vector<float> image_prepare_resnet(const uint8_t* rgb8, uint32_t width, uint32_t height, int device = 0) {
    std::vector<float> out((size_t)3 * 224 * 224);
    check(mi_image_prepare_resnet(device, rgb8, width, height, out.data()));
    return out;
}

namespace clip_vit_large_patch14 {
class Model {
    mi_clip* h_ = nullptr;
    uint32_t info_[8] = {0};
    explicit Model(mi_clip* h) : h_(h) { check(mi_clip_info(h_, info_)); }

   public:
    static Model from_file(const std::string& path, int device, int precision = MI_PRECISION_BF16) {
        mi_clip* h = nullptr;
        check(mi_clip_load(path.c_str(), device, precision, &h));
        return Model(h);
    }
    Model(Model&& o) noexcept { *this = std::move(o); }
    Model& operator=(Model&& o) noexcept { std::swap(h_, o.h_); std::swap(info_, o.info_); return *this; }
    Model(const Model&) = delete;
    ~Model() { mi_clip_free(h_); }
    uint32_t image() const { return info_[0]; }
    uint32_t proj() const { return info_[7]; }
    mi_clip* handle() const { return h_; }
    // [n,3,H,W] f32 NCHW -> [n,proj] f32, flat (what `output.to_data()` + cast gives, clip.rs:120-124)
    std::vector<float> forward(const std::vector<float>& nchw, size_t n) const {
        if (nchw.size() != n * 3 * (size_t)image() * image()) throw std::runtime_error("input is not [n,3,H,W]");
        std::vector<float> out(n * proj());
        check(mi_clip_embed(h_, nchw.data(), n, out.data()));
        return out;
    }
    // one chunk of the scan loop (clip.rs:92-124): decoded RGB8 images of any sizes -> [n,proj] f32;
    // CatmullRom resize_exact + normalisation + tower on the device
    struct Rgb8 { const uint8_t* data; uint32_t width, height; };
    std::vector<float> forward_images(const std::vector<Rgb8>& images) const {
        std::vector<const uint8_t*> p;
        std::vector<uint32_t> w, h;
        for (const auto& im : images) { p.push_back(im.data); w.push_back(im.width); h.push_back(im.height); }
        std::vector<float> out(images.size() * proj());
        check(mi_clip_embed_images(h_, p.data(), w.data(), h.data(), images.size(), out.data()));
        return out;
    }
};
}  // namespace clip_vit_large_patch14

// table `image` + index `mt_pts` (server/src/clip.rs:135-143) as one HBM-resident shard
class EmbeddingTable {
    mi_knn* h_ = nullptr;
    uint32_t dim_;

   public:
    explicit EmbeddingTable(uint32_t dim = 768, int device = 0, uint64_t base = 0) : dim_(dim) {
        check(mi_knn_create(dim, device, &h_));
        check(mi_knn_set_base(h_, base));
    }
    EmbeddingTable(const EmbeddingTable&) = delete;
    ~EmbeddingTable() { mi_knn_free(h_); }
    void insert(const std::vector<float>& rows) { check(mi_knn_append(h_, rows.data(), rows.size() / dim_)); }
    uint64_t size() const { uint64_t n = 0; check(mi_knn_size(h_, &n)); return n; }
    // what the database's storage did for the reference: one file per shard
    void save(const std::string& path) const { check(mi_knn_save(h_, path.c_str())); }
    void load(const std::string& path) { check(mi_knn_load(h_, path.c_str())); }
    // `WHERE embedding <|k|> $reference`: ids and cosine distances, ascending
    std::pair<std::vector<uint64_t>, std::vector<float>> knn(const std::vector<float>& reference, uint32_t k = 1000) const {
        std::vector<uint64_t> idx(k);
        std::vector<float> dist(k);
        check(mi_knn_search(h_, reference.data(), 1, k, idx.data(), dist.data()));
        return {idx, dist};
    }
    mi_knn* handle() const { return h_; }
    // "prefilter" = 2 (bytes) or 1 (bf16): the two-stage exact search, same results from a quarter / a half of the bytes
    void set_option(const std::string& key, int value) { check(mi_knn_set_option(h_, key.c_str(), value)); }
};

// the whole table `image` {id, image_path, embedding} (server/src/search.rs:13-18) and the four statements the
// server issues against it
class ImageIndex {
    mi_index* h_ = nullptr;
    static std::vector<const char*> ptrs(const std::vector<std::string>& v) {
        std::vector<const char*> p;
        for (const auto& s : v) p.push_back(s.c_str());
        return p;
    }

   public:
    explicit ImageIndex(uint32_t dim = 768, int device = 0, const std::string& media_dir = "") {
        check(mi_index_create(dim, device, media_dir.c_str(), &h_));
    }
    ImageIndex(const ImageIndex&) = delete;
    ~ImageIndex() { mi_index_free(h_); }
    mi_index* handle() const { return h_; }
    uint64_t size() const { uint64_t n = 0; check(mi_index_size(h_, &n)); return n; }
    // SELECT image_path FROM image WHERE image_path IN $paths  (server/src/clip.rs:74-83)
    std::vector<bool> existing(const std::vector<std::string>& paths) const {
        std::vector<uint8_t> e(paths.size());
        const auto p = ptrs(paths);
        check(mi_index_existing(h_, p.data(), p.size(), e.data()));
        return std::vector<bool>(e.begin(), e.end());
    }
    // db.insert("image").content(rows)  (server/src/clip.rs:125-137): returns the id of the first row
    uint64_t insert(const std::vector<std::string>& paths, const std::vector<float>& embeddings) {
        uint64_t first = 0;
        const auto p = ptrs(paths);
        check(mi_index_insert(h_, p.data(), embeddings.data(), p.size(), &first));
        return first;
    }
    std::string path(uint64_t id, bool web = false) const {
        size_t need = 0;
        check(mi_index_path(h_, id, web ? 1 : 0, nullptr, 0, &need));
        std::string s(need - 1, '\0');  // `needed` counts the terminating NUL
        check(mi_index_path(h_, id, web ? 1 : 0, &s[0], need, &need));
        return s;
    }
    // web_search_text behind the text embedding (server/src/search.rs:43-110): refine with the referenced images'
    // stored embeddings, then `embedding <|k|> $reference`; (id, distance) ascending, missing results dropped
    std::vector<std::pair<uint64_t, float>> search(const std::vector<float>& text_embedding, const std::vector<std::string>& referenced_images,
                                                   uint32_t k = 1000) const {
        std::vector<uint64_t> idx(k);
        std::vector<float> dist(k);
        uint32_t n = 0;
        const auto p = ptrs(referenced_images);
        check(mi_index_search(h_, text_embedding.data(), p.data(), p.size(), k, idx.data(), dist.data(), &n));
        std::vector<std::pair<uint64_t, float>> out;
        for (uint32_t i = 0; i < n; ++i) out.emplace_back(idx[i], dist[i]);
        return out;
    }
    void save(const std::string& dir) const { check(mi_index_save(h_, dir.c_str())); }
    void load(const std::string& dir) { check(mi_index_load(h_, dir.c_str())); }
};

// the table row-sharded over several GPUs of ONE process: same results as one EmbeddingTable
class ShardedTable {
    mi_knn_sharded* h_ = nullptr;
    uint32_t dim_;

   public:
    ShardedTable(uint32_t dim, const std::vector<int>& devices, uint32_t block_rows = 0) : dim_(dim) {
        check(mi_knn_sharded_create(dim, devices.data(), (int)devices.size(), block_rows, &h_));
    }
    ShardedTable(const ShardedTable&) = delete;
    ~ShardedTable() { mi_knn_sharded_free(h_); }
    uint64_t insert(const std::vector<float>& rows) {
        uint64_t first = 0;
        check(mi_knn_sharded_append(h_, rows.data(), rows.size() / dim_, &first));
        return first;
    }
    uint64_t size() const { uint64_t n = 0; check(mi_knn_sharded_info(h_, &n, nullptr, nullptr, nullptr)); return n; }
    void set_option(const std::string& key, int value) { check(mi_knn_sharded_set_option(h_, key.c_str(), value)); }
    // {searches, ncclAllGather calls, transport copies, device merges} issued so far
    std::vector<uint64_t> stats() const { std::vector<uint64_t> v(4); check(mi_knn_sharded_stats(h_, v.data())); return v; }
    std::pair<std::vector<uint64_t>, std::vector<float>> knn(const std::vector<float>& reference, uint32_t k = 1000) const {
        std::vector<uint64_t> idx(k);
        std::vector<float> dist(k);
        check(mi_knn_sharded_search(h_, reference.data(), 1, k, idx.data(), dist.data()));
        return {idx, dist};
    }
    // the same search without the wait: idx / dist (caller-owned, k entries each) are filled when sync() returns
    void knn_async(const float* reference, uint32_t k, uint64_t* idx, float* dist) { check(mi_knn_sharded_search_async(h_, reference, 1, k, idx, dist)); }
    void sync() { check(mi_knn_sharded_sync(h_)); }
    // rows that are already in device memory on `src_device` (a replica's embeddings): routed to their shards device to device
    uint64_t insert_device(const float* d_rows, uint64_t n, int src_device, void* stream = nullptr) {
        uint64_t first = 0;
        check(mi_knn_sharded_append_device(h_, d_rows, n, src_device, stream, &first));
        return first;
    }
    // every row of `src` into this EMPTY table (another shard count / device set / block size), device to device
    void rebalance_from(ShardedTable& src) { check(mi_knn_sharded_rebalance(h_, src.h_)); }
    void save(const std::string& prefix) const { check(mi_knn_sharded_save(h_, prefix.c_str())); }
    void load(const std::string& prefix) { check(mi_knn_sharded_load(h_, prefix.c_str())); }
    mi_knn_sharded* handle() const { return h_; }
};

// the body of the scan loop and the query on HIP streams (BASELINE config 4)
class Pipeline {
    mi_pipeline* h_ = nullptr;

   public:
    Pipeline(clip_vit_large_patch14::Model& model, EmbeddingTable& table) { check(mi_pipeline_create(model.handle(), table.handle(), &h_)); }
    // ONE process over several GPUs: models[s] is the tower replica on the device of shard s
    Pipeline(const std::vector<clip_vit_large_patch14::Model*>& models, ShardedTable& table) {
        std::vector<mi_clip*> hs;
        for (auto* m : models) hs.push_back(m->handle());
        check(mi_pipeline_create_sharded(hs.data(), (int)hs.size(), table.handle(), &h_));
    }
    Pipeline(const Pipeline&) = delete;
    ~Pipeline() { mi_pipeline_free(h_); }
    // [n,3,H,W] f32 (pinned memory from mi_host_alloc makes the upload asynchronous): returns the id of the first new row
    uint64_t ingest(const float* nchw, size_t n) {
        uint64_t first = 0;
        check(mi_pipeline_ingest(h_, nchw, n, &first));
        return first;
    }
    // results land in idx / dist when sync() (or drain) returns
    void query(const float* q, uint32_t k, uint64_t* idx, float* dist) { check(mi_pipeline_query(h_, q, k, idx, dist)); }
    void sync() { check(mi_pipeline_sync(h_)); }
    // deliver finished queries until at most `leave_pending` remain (the ingest stream is not waited for)
    void drain(uint32_t leave_pending = 0) { check(mi_pipeline_drain(h_, leave_pending)); }
};


}  // namespace image_search
