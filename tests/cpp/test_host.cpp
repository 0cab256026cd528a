// The reference's own unit tests, restated in C++ against the drop-in (no GPU needed), plus — when
// run with the argument "gpu" — a small end-to-end use of the C++ mirror on device 0.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../image_search_amd/host/image_search.hpp"

using namespace image_search;
#define EXPECT(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
    {   // server/src/search.rs:156-161  tes_average_vector
        std::vector<float> a{1.0f, 2.0f, 4.0f, 4.0f, 10.0f}, b{1.0f, 1.0f, 2.0f, 4.0f, 0.0f};
        auto r = average_slices({&a, &b});
        EXPECT((r == std::vector<float>{1.0f, 1.5f, 3.0f, 4.0f, 5.0f}));
        bool threw = false;
        try { average_slices({}); } catch (const std::runtime_error& e) { threw = std::string(e.what()) == "Input must not be empty"; }
        EXPECT(threw);
        std::vector<float> c{1.0f};
        threw = false;
        try { average_slices({&a, &c}); } catch (const std::runtime_error&) { threw = true; }
        EXPECT(threw);
    }
    {   // server/src/clip.rs:181-233  test_matches
        EXPECT(!is_image_path("file.txt"));
        EXPECT(is_image_path("file.jpg"));
        EXPECT(is_image_path("file.png"));
        EXPECT(!is_image_path("file.mp4"));
        EXPECT(!is_image_path("file"));
    }
    {   // image_prepare_resnet arithmetic, server/src/clip.rs:164-172
        std::vector<uint8_t> px(224 * 224 * 3);
        for (size_t i = 0; i < px.size(); ++i) px[i] = (uint8_t)(i * 7 + 3);
        auto chw = image_prepare_resnet(px);
        const size_t i = 1234;
        EXPECT(chw[i] == ((float)px[i * 3] / 255.0f - 0.485f) / 0.229f);
        EXPECT(chw[224 * 224 + i] == ((float)px[i * 3 + 1] / 255.0f - 0.456f) / 0.224f);
        EXPECT(chw[2 * 224 * 224 + i] == ((float)px[i * 3 + 2] / 255.0f - 0.406f) / 0.225f);
    }
    if (argc > 1 && std::strcmp(argv[1], "gpu") == 0) {
        EmbeddingTable t(768, 0);
        std::vector<float> rows(5 * 768, 0.0f);
        for (int r = 0; r < 5; ++r) { rows[r * 768 + r] = 1.0f; rows[r * 768 + 5] = 0.1f * r; }
        t.insert(rows);
        EXPECT(t.size() == 5);
        std::vector<float> q(768, 0.0f); q[3] = 1.0f;
        auto res = t.knn(q, 7);
        EXPECT(res.first[0] == 3);
        EXPECT(res.first[5] == MI_KNN_NO_ID && std::isinf(res.second[6]));
        for (int i = 1; i < 5; ++i) EXPECT(res.second[i] >= res.second[i - 1]);
        // the table with its path column: the reference's four statements
        ImageIndex ix(768, 0, "/srv/media/");
        const std::vector<std::string> paths{"/srv/media/a.jpg", "/srv/media/sub/b.png", "/srv/media/c.jpg", "/srv/media/d.jpg", "/srv/media/e.jpg"};
        EXPECT(ix.insert(paths, rows) == 0 && ix.size() == 5);
        const auto ex = ix.existing({"/srv/media/c.jpg", "/srv/media/zzz.jpg"});
        EXPECT(ex[0] && !ex[1]);
        EXPECT(ix.path(1) == "/srv/media/sub/b.png" && ix.path(1, true) == "media/sub/b.png");
        const auto hits = ix.search(q, {}, 1000);          // fewer rows than K: five results, nearest first
        EXPECT(hits.size() == 5 && hits[0].first == 3);
        const auto refined = ix.search(q, {"/srv/media/a.jpg"}, 2);   // reference = mean(mean(selected), text): rows 0 and 3 lead
        EXPECT(refined.size() == 2 && ((refined[0].first == 0 && refined[1].first == 3) || (refined[0].first == 3 && refined[1].first == 0)));
        // three shards on one device (64-row blocks dealt round-robin) give what the single table gives, bit for bit
        std::vector<float> many(200 * 768, 0.0f);
        for (int r = 0; r < 200; ++r) { many[r * 768 + r] = 1.0f; many[r * 768 + 300] = 0.01f * r; many[r * 768 + 3] += 0.001f * (r % 7); }
        EmbeddingTable one(768, 0);
        one.insert(many);
        ShardedTable st(768, {0, 0, 0}, 64);
        EXPECT(st.insert(many) == 0 && st.size() == 200);
        const auto a = one.knn(q, 50), b = st.knn(q, 50);
        EXPECT(a.first == b.first);
        EXPECT(std::memcmp(a.second.data(), b.second.data(), 50 * sizeof(float)) == 0);
        // round 3: searches in flight, a live change of layout (device to device), a crash-safe save / load
        std::vector<uint64_t> ai(50), bi(50);
        std::vector<float> ad(50), bd(50);
        std::vector<float> q2(768, 0.0f); q2[7] = 1.0f;
        st.knn_async(q.data(), 50, ai.data(), ad.data());
        st.knn_async(q2.data(), 50, bi.data(), bd.data());
        st.sync();
        EXPECT(ai == a.first && std::memcmp(ad.data(), a.second.data(), 50 * sizeof(float)) == 0);
        EXPECT(bi[0] == 7);
        ShardedTable two(768, {0, 0}, 128);
        two.rebalance_from(st);
        EXPECT(two.size() == 200);
        const auto c = two.knn(q, 50);
        EXPECT(c.first == a.first && std::memcmp(c.second.data(), a.second.data(), 50 * sizeof(float)) == 0);
        const std::string prefix = std::string(argc > 2 ? argv[2] : "/tmp") + "/cpp_sharded";
        two.save(prefix);
        ShardedTable back(768, {0, 0, 0, 0}, 64);          // another layout: re-dealt on load
        back.load(prefix);
        const auto d = back.knn(q, 50);
        EXPECT(back.size() == 200 && d.first == a.first);
    } else if (mi_device_count() == 0) {
        bool threw = false;
        try { EmbeddingTable t(768, 0); } catch (const std::runtime_error& e) { threw = std::strstr(e.what(), "no CPU fallback") != nullptr; }
        EXPECT(threw);  // loud failure, no fallback
    }
    std::printf("ok\n");
    return 0;
}
