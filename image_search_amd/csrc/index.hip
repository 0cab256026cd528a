// index.hip — table `image` {id, image_path, embedding} (server/src/search.rs:13-18) behind the C ABI: one embedding
// shard (mi_knn) plus the image_path column, so that the statements the reference server issues need no host-language
// glue above the library:
//   SELECT image_path FROM image WHERE image_path IN $paths            server/src/clip.rs:74-83     mi_index_existing
//   db.insert("image").content(rows)                                   server/src/clip.rs:125-137   mi_index_insert
//   SELECT id, image_path, embedding FROM image WHERE image_path IN $p server/src/search.rs:43-58   mi_index_rows_of
//   refine + SELECT id, image_path, knn() ... <|K|> $reference         server/src/search.rs:20-110  mi_index_search
// Row id = insertion ordinal; like the reference's table there is no uniqueness constraint on image_path (the scan
// loop filters first), a path may own several rows and lookups return all of them in id order.
// Persistence: `<dir>/embedding.miknn` (mi_knn_save) and `<dir>/image_path.bin`, each written to a temporary name,
// fsync'ed and renamed; the path file goes last and carries the row count, so a crash between the two leaves the OLD
// path file beside a NEWER embedding file — load then keeps the rows both files agree on (the reference's database
// commits per chunk; here a crash costs at most the chunks since the last save).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "handles.h"

using namespace mi;

struct mi_index {
    mi_knn* table = nullptr;
    uint32_t dim = 0;
    std::string media_dir;  // the server's media directory: "media/..." in requests maps onto it (search.rs:35-40)
    std::vector<std::string> paths;                            // row id -> image_path
    std::unordered_map<std::string, std::vector<uint64_t>> rows_of;  // image_path -> row ids, ascending
    std::mutex mu;
};

namespace {

void add_path(mi_index* ix, const std::string& p) {
    ix->rows_of[p].push_back(ix->paths.size());
    ix->paths.push_back(p);
}

// "media/x.jpg" as the client names it -> the path the row was stored under (search.rs:35-40: only such names are looked up)
bool to_disk(const mi_index* ix, const char* web, std::string* out) {
    if (std::strncmp(web, "media/", 6) != 0) return false;
    *out = ix->media_dir + (web + 6);
    return true;
}

void write_all(int fd, const void* p, size_t n, const char* what) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t w = ::write(fd, c, n);
        if (w <= 0) fail(MI_ERR_IO, "write to %s failed (disk full?)", what);
        c += w; n -= (size_t)w;
    }
}

}  // namespace

extern "C" {

int mi_index_create(uint32_t dim, int device, const char* media_dir, mi_index** out) {
    mi_index* ix = nullptr;
    const int rc = guarded([&] {
        if (!out) fail(MI_ERR_INVALID, "out is null");
        *out = nullptr;
        ix = new mi_index();
        ix->dim = dim;
        ix->media_dir = media_dir ? media_dir : "";
        const int e = mi_knn_create(dim, device, &ix->table);
        if (e != MI_OK) fail(e, "%s", mi_last_error());
        *out = ix;
    });
    if (rc != MI_OK && ix) { delete ix; }
    return rc;
}

void mi_index_free(mi_index* ix) {
    if (!ix) return;
    mi_knn_free(ix->table);
    delete ix;
}

mi_knn* mi_index_table(mi_index* ix) { return ix ? ix->table : nullptr; }

int mi_index_media_dir(mi_index* ix, char* buf, size_t cap, size_t* needed) {
    return guarded([&] {
        if (!ix) fail(MI_ERR_INVALID, "null index handle");
        std::lock_guard<std::mutex> l(ix->mu);
        if (needed) *needed = ix->media_dir.size() + 1;
        if (buf && cap) {
            const size_t n = std::min(cap - 1, ix->media_dir.size());
            std::memcpy(buf, ix->media_dir.data(), n);
            buf[n] = '\0';
        }
    });
}

int mi_index_size(mi_index* ix, uint64_t* rows) {
    return guarded([&] {
        if (!ix || !rows) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(ix->mu);
        *rows = ix->paths.size();
    });
}

int mi_index_existing(mi_index* ix, const char* const* paths, size_t n, uint8_t* exists) {
    return guarded([&] {
        if (!ix || (n && (!paths || !exists))) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(ix->mu);
        for (size_t i = 0; i < n; ++i) {
            if (!paths[i]) fail(MI_ERR_INVALID, "path %zu is null", i);
            exists[i] = ix->rows_of.count(paths[i]) ? 1 : 0;
        }
    });
}

int mi_index_insert(mi_index* ix, const char* const* paths, const float* embeddings, size_t n, uint64_t* first_id) {
    return guarded([&] {
        if (!ix) fail(MI_ERR_INVALID, "null index handle");
        std::lock_guard<std::mutex> l(ix->mu);
        if (first_id) *first_id = ix->paths.size();
        if (n == 0) return;
        if (!paths || !embeddings) fail(MI_ERR_INVALID, "null argument");
        for (size_t i = 0; i < n; ++i)
            if (!paths[i]) fail(MI_ERR_INVALID, "path %zu is null", i);
        const int e = mi_knn_append(ix->table, embeddings, n);  // the rows first: a failure leaves the path column untouched
        if (e != MI_OK) fail(e, "%s", mi_last_error());
        for (size_t i = 0; i < n; ++i) add_path(ix, paths[i]);
    });
}

// paths whose embeddings came from the fused pipeline (mi_pipeline_ingest wrote the rows into mi_index_table already)
int mi_index_adopt(mi_index* ix, const char* const* paths, size_t n) {
    return guarded([&] {
        if (!ix || (n && !paths)) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(ix->mu);
        uint64_t rows = 0;
        const int e = mi_knn_size(ix->table, &rows);
        if (e != MI_OK) fail(e, "%s", mi_last_error());
        if (ix->paths.size() + n != rows)
            fail(MI_ERR_INVALID, "%zu paths for %llu rows without one (the table holds %llu rows, the path column %zu)", n,
                 (unsigned long long)(rows - ix->paths.size()), (unsigned long long)rows, ix->paths.size());
        for (size_t i = 0; i < n; ++i) {
            if (!paths[i]) fail(MI_ERR_INVALID, "path %zu is null", i);
            add_path(ix, paths[i]);
        }
    });
}

int mi_index_rows_of(mi_index* ix, const char* const* paths, size_t n, uint64_t* ids, size_t cap, size_t* count) {
    return guarded([&] {
        if (!ix || !count || (n && !paths)) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(ix->mu);
        std::vector<uint64_t> found;
        for (size_t i = 0; i < n; ++i) {
            if (!paths[i]) fail(MI_ERR_INVALID, "path %zu is null", i);
            auto it = ix->rows_of.find(paths[i]);
            if (it != ix->rows_of.end()) found.insert(found.end(), it->second.begin(), it->second.end());
        }
        // table (id) order whatever the request order, each row once: average_slices adds in input order (search.rs:139-143)
        std::sort(found.begin(), found.end());
        found.erase(std::unique(found.begin(), found.end()), found.end());
        *count = found.size();
        if (ids) std::memcpy(ids, found.data(), std::min(cap, found.size()) * sizeof(uint64_t));
    });
}

int mi_index_path(mi_index* ix, uint64_t id, int web, char* buf, size_t cap, size_t* needed) {
    return guarded([&] {
        if (!ix) fail(MI_ERR_INVALID, "null index handle");
        std::lock_guard<std::mutex> l(ix->mu);
        if (id >= ix->paths.size()) fail(MI_ERR_INVALID, "id %llu out of range (%zu rows)", (unsigned long long)id, ix->paths.size());
        std::string p = ix->paths[id];
        // search.rs:104-109: what goes back to the client is relative to "media/"
        if (web && !ix->media_dir.empty() && p.compare(0, ix->media_dir.size(), ix->media_dir) == 0) p = "media/" + p.substr(ix->media_dir.size());
        if (needed) *needed = p.size() + 1;
        if (buf && cap) {
            const size_t n = std::min(cap - 1, p.size());
            std::memcpy(buf, p.data(), n);
            buf[n] = '\0';
        }
    });
}

// web_search_text after the text tower (search.rs:20-110): query = text, refined with the marked images that are in
// the table (mean of their embeddings in id order, then mean of that and the text vector); K nearest by cosine distance.
int mi_index_search(mi_index* ix, const float* text_embedding, const char* const* referenced_images, size_t n_ref, uint32_t k,
                    uint64_t* idx, float* dist, uint32_t* n_found) {
    return guarded([&] {
        if (!ix || !text_embedding || !idx || !dist || (n_ref && !referenced_images)) fail(MI_ERR_INVALID, "null argument");
        std::vector<uint64_t> marked;
        {
            std::lock_guard<std::mutex> l(ix->mu);
            for (size_t i = 0; i < n_ref; ++i) {
                std::string disk;
                if (!referenced_images[i] || !to_disk(ix, referenced_images[i], &disk)) continue;  // search.rs:35-40
                auto it = ix->rows_of.find(disk);
                if (it != ix->rows_of.end()) marked.insert(marked.end(), it->second.begin(), it->second.end());
            }
        }
        std::sort(marked.begin(), marked.end());
        marked.erase(std::unique(marked.begin(), marked.end()), marked.end());
        std::vector<float> query(text_embedding, text_embedding + ix->dim);
        if (!marked.empty()) {  // search.rs:59-67
            std::vector<float> sel(marked.size() * ix->dim);
            std::vector<const float*> ptr(marked.size());
            for (size_t i = 0; i < marked.size(); ++i) {
                const int e = mi_knn_get_rows(ix->table, marked[i], 1, &sel[i * ix->dim]);
                if (e != MI_OK) fail(e, "%s", mi_last_error());
                ptr[i] = &sel[i * ix->dim];
            }
            const int e = mi_refine(text_embedding, ptr.data(), ptr.size(), ix->dim, query.data());
            if (e != MI_OK) fail(e, "%s", mi_last_error());
        }
        const int e = mi_knn_search(ix->table, query.data(), 1, k, idx, dist);
        if (e != MI_OK) fail(e, "%s", mi_last_error());
        if (n_found) {
            uint32_t n = 0;
            while (n < k && idx[n] != MI_KNN_NO_ID) ++n;
            *n_found = n;
        }
    });
}

int mi_index_save(mi_index* ix, const char* dir) {
    return guarded([&] {
        if (!ix || !dir) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(ix->mu);
        if (::mkdir(dir, 0777) != 0 && errno != EEXIST) fail(MI_ERR_IO, "cannot create directory %s", dir);
        const std::string d(dir);
        const int e = mi_knn_save(ix->table, (d + "/embedding.miknn").c_str());  // tmp + fsync + rename inside
        if (e != MI_OK) fail(e, "%s", mi_last_error());
        // image_path.bin: "MIPATHv1", u64 rows, u32 media_dir length + bytes, then per row u32 length + bytes
        const std::string tmp = d + "/image_path.bin.tmp", fin = d + "/image_path.bin";
        const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd < 0) fail(MI_ERR_IO, "cannot create %s", tmp.c_str());
        try {
            std::string blob("MIPATHv1");
            const uint64_t rows = ix->paths.size();
            blob.append(reinterpret_cast<const char*>(&rows), 8);
            auto put = [&](const std::string& s) {
                const uint32_t n = (uint32_t)s.size();
                blob.append(reinterpret_cast<const char*>(&n), 4);
                blob.append(s);
                if (blob.size() > (8u << 20)) { write_all(fd, blob.data(), blob.size(), tmp.c_str()); blob.clear(); }
            };
            put(ix->media_dir);
            for (const std::string& p : ix->paths) put(p);
            write_all(fd, blob.data(), blob.size(), tmp.c_str());
            if (::fsync(fd) != 0) fail(MI_ERR_IO, "fsync of %s failed", tmp.c_str());
        } catch (...) {
            ::close(fd);
            throw;
        }
        if (::close(fd) != 0) fail(MI_ERR_IO, "close of %s failed", tmp.c_str());
        if (std::rename(tmp.c_str(), fin.c_str()) != 0) fail(MI_ERR_IO, "cannot rename %s", tmp.c_str());
        const int dfd = ::open(dir, O_RDONLY);  // the renames themselves become durable with the directory
        if (dfd >= 0) { (void)::fsync(dfd); ::close(dfd); }
    });
}

int mi_index_load(mi_index* ix, const char* dir) {
    return guarded([&] {
        if (!ix || !dir) fail(MI_ERR_INVALID, "null argument");
        std::lock_guard<std::mutex> l(ix->mu);
        uint64_t have = 0;
        (void)mi_knn_size(ix->table, &have);
        // (rows the pipeline has written into the table but mi_index_adopt has not named yet count too)
        if (!ix->paths.empty() || have != 0) fail(MI_ERR_INVALID, "mi_index_load needs an empty index (%zu paths, %llu embeddings)",
                                                  ix->paths.size(), (unsigned long long)have);
        const std::string d(dir), pf = d + "/image_path.bin";
        FILE* f = std::fopen(pf.c_str(), "rb");
        if (!f) fail(MI_ERR_IO, "cannot open %s", pf.c_str());
        struct stat sb {};
        const uint64_t file_bytes = ::fstat(fileno(f), &sb) == 0 ? (uint64_t)sb.st_size : 0;
        std::vector<std::string> paths;
        std::string media;
        try {
            char magic[8];
            uint64_t rows = 0;
            if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "MIPATHv1", 8) != 0 || std::fread(&rows, 8, 1, f) != 1)
                fail(MI_ERR_IO, "%s is not a MIPATHv1 file", pf.c_str());
            auto get = [&](std::string* s) {
                uint32_t n = 0;
                if (std::fread(&n, 4, 1, f) != 1 || n > (1u << 20)) fail(MI_ERR_IO, "%s is truncated or corrupt", pf.c_str());
                s->resize(n);
                if (n && std::fread(&(*s)[0], 1, n, f) != n) fail(MI_ERR_IO, "%s is truncated", pf.c_str());
            };
            get(&media);
            // every row costs at least its 4-byte length: a count the file cannot hold is corruption, not an allocation request
            if (rows > file_bytes / 4) fail(MI_ERR_IO, "%s claims %llu rows in %llu bytes", pf.c_str(), (unsigned long long)rows,
                                            (unsigned long long)file_bytes);
            paths.resize(rows);
            for (auto& p : paths) get(&p);
        } catch (...) {
            std::fclose(f);
            throw;
        }
        std::fclose(f);
        const int e = mi_knn_load(ix->table, (d + "/embedding.miknn").c_str());
        if (e != MI_OK) fail(e, "%s", mi_last_error());
        uint64_t rows = 0;
        (void)mi_knn_size(ix->table, &rows);
        // the embedding file is written first: it may be NEWER than the path file (a crash between the two renames);
        // rows without a path cannot be served and are not kept.  The other way round cannot happen.
        if (rows < paths.size()) fail(MI_ERR_IO, "%s: %llu embeddings for %zu paths", dir, (unsigned long long)rows, paths.size());
        if (rows > paths.size()) knn_truncate(ix->table, paths.size());  // under the table's own lock; rows beyond are rewritten by later inserts
        ix->media_dir = media;
        for (const auto& p : paths) add_path(ix, p);
    });
}

}  // extern "C"
