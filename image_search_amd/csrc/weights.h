// weights.h — checkpoint files behind mi_clip_load / mi_clip_load_text (host only).
//
//   SafeTensors  Hugging Face `safetensors` with the CLIPVisionModelWithProjection / CLIPTextModelWithProjection
//                tensor names (F32 / F16 / BF16).
//   BurnMpk      what `-w` of the reference server points at: `vision_model.mpk`, written at build time by
//                burn-import's ModelGen through Burn's NamedMpkFileRecorder (clip/build.rs:75-83, loaded by
//                Model::from_file at server/src/clip.rs:46-48, default path server/src/server_arguments.rs:8-9).
//                Layout (Burn 0.19, restated from the published format; no such file exists offline):
//                MessagePack, named (maps keyed by field name): {"metadata": {...}, "item": <module record>}; a
//                parameter is {"id": str, "param": <tensor>}, a tensor is {"bytes": bin, "shape": [u64..],
//                "dtype": "F32"} (older records: {"value": [f32..], "shape": [..]}).  The field names of the
//                module record are those of the struct burn-import GENERATES from the ONNX graph (conv2d1,
//                linear7, ...), which is not in the reference tree — so tensors are mapped to the Hugging Face names
//                either by name (a record that already uses them) or BY SHAPE AND ORDER: document order is taken to
//                be graph order (patch conv, class / position embedding, pre-LN, then per layer LN1, q, k, v, out,
//                LN2, fc1, fc2, then post-LN, projection), every shape class must hold exactly the number of tensors
//                that order implies, and anything else is refused with the inventory in the message.  Burn keeps
//                Linear weights [d_in, d_out]: they are transposed on read (a [d_out, d_in] fc1/fc2/projection is
//                recognised by its shape and left alone; square q/k/v/out weights are taken as Burn's).
//                The graph the reference imports is an OPSET-16 export (clip/scripts/upgrade_opset.py:9-28): LayerNorm
//                is decomposed there, so its gamma / beta are bare `constantN` parameters between Mul / Add nodes, not
//                `layernormalizationN {gamma, beta}` modules, and the record may carry the decomposition's scalar
//                constants and integer tensors (position ids, shapes).  The reader therefore (a) sets aside — and lists —
//                leaves that cannot be a tower tensor (non-float dtype, fewer than 8 elements, a 1-D length that is
//                neither D nor FF), (b) binds a bias to the matrix of its own module when there is one, and (c) places
//                the remaining [D]-vectors in graph order; both the fused and the decomposed inventory, with the linears
//                as Linear modules or as bare MatMul + Add constants, map to the same names.
//                PARITY UNPINNED: no real burn-import file exists offline; tests use tools/make_synthetic_mpk.py.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace mi {

struct TensorInfo {
    std::string dtype;
    std::vector<int64_t> shape;
    uint64_t begin = 0, end = 0;
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : shape) n *= d;
        return n;
    }
};

// raw little-endian elements -> fp32
inline void to_f32(const void* src, const std::string& dtype, int64_t n, float* out, const char* what) {
    if (dtype == "F32") { std::memcpy(out, src, (size_t)n * 4); return; }
    const uint16_t* raw = static_cast<const uint16_t*>(src);
    if (dtype == "BF16") {
        for (int64_t i = 0; i < n; ++i) { const uint32_t b = (uint32_t)raw[i] << 16; std::memcpy(&out[i], &b, 4); }
        return;
    }
    if (dtype == "F16") {
        for (int64_t i = 0; i < n; ++i) {  // IEEE half -> float
            const uint32_t hbits = raw[i], sign = (hbits & 0x8000u) << 16;
            uint32_t ex = (hbits >> 10) & 0x1f, man = hbits & 0x3ffu, b;
            if (ex == 0) {
                if (man == 0) b = sign;
                else { int sh = 0; while (!(man & 0x400u)) { man <<= 1; ++sh; } man &= 0x3ffu; b = sign | ((uint32_t)(113 - sh) << 23) | (man << 13); }
            } else if (ex == 31) b = sign | 0x7f800000u | (man << 13);
            else b = sign | ((ex + 112) << 23) | (man << 13);
            std::memcpy(&out[i], &b, 4);
        }
        return;
    }
    fail(MI_ERR_UNSUPPORTED, "tensor '%s': dtype %s (F32/F16/BF16 supported)", what, dtype.c_str());
}

struct WeightFile {
    std::map<std::string, std::string> meta;
    virtual ~WeightFile() {}
    virtual bool has(const std::string& name) const = 0;
    virtual const TensorInfo& info(const std::string& name) const = 0;       // shape in PyTorch convention ([out, in])
    virtual std::vector<float> read(const std::string& name, int64_t numel) const = 0;
    virtual std::vector<std::string> names() const = 0;                      // in file order
    virtual std::vector<std::string> skipped_lines() const { return {}; }    // "(set aside) path dtype [shape]" of leaves that map to no tensor
};

// ------------------------------------------------------------------ safetensors
// Minimal JSON reader for the safetensors header (one object of objects).
struct Json {
    const char* p;
    const char* e;
    void ws() { while (p < e && std::isspace((unsigned char)*p)) ++p; }
    void expect(char c) {
        ws();
        if (p >= e || *p != c) fail(MI_ERR_IO, "safetensors header: expected '%c'", c);
        ++p;
    }
    bool peek(char c) { ws(); return p < e && *p == c; }
    std::string str() {
        expect('"');
        std::string s;
        while (p < e && *p != '"') {
            if (*p == '\\' && p + 1 < e) { ++p; }
            s.push_back(*p++);
        }
        expect('"');
        return s;
    }
    int64_t num() {
        ws();
        char* end = nullptr;
        const long long v = std::strtoll(p, &end, 10);
        if (end == p) fail(MI_ERR_IO, "safetensors header: expected a number");
        p = end;
        return v;
    }
    void skip() {  // any value
        ws();
        if (peek('"')) { str(); return; }
        if (peek('{')) { ++p; if (peek('}')) { ++p; return; } do { str(); expect(':'); skip(); } while (peek(',') && ++p); expect('}'); return; }
        if (peek('[')) { ++p; if (peek(']')) { ++p; return; } do { skip(); } while (peek(',') && ++p); expect(']'); return; }
        while (p < e && *p != ',' && *p != '}' && *p != ']') ++p;
    }
};

struct SafeTensors : WeightFile {
    FILE* f = nullptr;
    uint64_t data_start = 0, file_size = 0;
    std::map<std::string, TensorInfo> tensors;
    std::vector<std::string> order;

    explicit SafeTensors(const char* path) {
        f = std::fopen(path, "rb");
        if (!f) fail(MI_ERR_IO, "cannot open weights file '%s'", path);
        std::fseek(f, 0, SEEK_END);
        file_size = (uint64_t)std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        uint64_t hl = 0;
        if (std::fread(&hl, 8, 1, f) != 1 || hl == 0 || hl > file_size - 8 || hl > (256u << 20))
            fail(MI_ERR_IO, "'%s' is not a safetensors file (bad header length)", path);
        std::string h(hl, '\0');
        if (std::fread(&h[0], 1, hl, f) != hl) fail(MI_ERR_IO, "'%s': truncated header", path);
        data_start = 8 + hl;
        Json j{h.data(), h.data() + h.size()};
        j.expect('{');
        if (!j.peek('}')) {
            do {
                const std::string name = j.str();
                j.expect(':');
                if (name == "__metadata__") {
                    j.expect('{');
                    if (!j.peek('}')) do { std::string k = j.str(); j.expect(':'); meta[k] = j.str(); } while (j.peek(',') && ++j.p);
                    j.expect('}');
                    continue;
                }
                TensorInfo t;
                j.expect('{');
                do {
                    const std::string key = j.str();
                    j.expect(':');
                    if (key == "dtype") t.dtype = j.str();
                    else if (key == "shape") {
                        j.expect('[');
                        if (!j.peek(']')) do { t.shape.push_back(j.num()); } while (j.peek(',') && ++j.p);
                        j.expect(']');
                    } else if (key == "data_offsets") {
                        j.expect('['); t.begin = (uint64_t)j.num(); j.expect(','); t.end = (uint64_t)j.num(); j.expect(']');
                    } else j.skip();
                } while (j.peek(',') && ++j.p);
                j.expect('}');
                if (data_start + t.end > file_size || t.begin > t.end)
                    fail(MI_ERR_IO, "tensor '%s': data offsets outside the file", name.c_str());
                tensors[name] = t;
                order.push_back(name);
            } while (j.peek(',') && ++j.p);
        }
        j.expect('}');
    }
    ~SafeTensors() override { if (f) std::fclose(f); }

    const TensorInfo& info(const std::string& name) const override {
        auto it = tensors.find(name);
        if (it == tensors.end()) fail(MI_ERR_IO, "weights file lacks tensor '%s'", name.c_str());
        return it->second;
    }
    bool has(const std::string& name) const override { return tensors.count(name) != 0; }
    std::vector<std::string> names() const override { return order; }

    // tensor as fp32, checked against `numel`
    std::vector<float> read(const std::string& name, int64_t numel) const override {
        const TensorInfo& t = info(name);
        if (t.numel() != numel)
            fail(MI_ERR_IO, "tensor '%s' has %lld elements, expected %lld", name.c_str(), (long long)t.numel(), (long long)numel);
        const size_t esz = t.dtype == "F32" ? 4 : (t.dtype == "F16" || t.dtype == "BF16") ? 2 : 0;
        if (!esz) fail(MI_ERR_UNSUPPORTED, "tensor '%s': dtype %s (F32/F16/BF16 supported)", name.c_str(), t.dtype.c_str());
        if (t.end - t.begin != (uint64_t)numel * esz) fail(MI_ERR_IO, "tensor '%s': byte size mismatch", name.c_str());
        std::vector<float> out((size_t)numel);
        if (fseeko(f, (off_t)(data_start + t.begin), SEEK_SET) != 0) fail(MI_ERR_IO, "seek failed");
        if (esz == 4) {
            if (std::fread(out.data(), 4, (size_t)numel, f) != (size_t)numel) fail(MI_ERR_IO, "'%s': short read", name.c_str());
        } else {
            std::vector<uint16_t> raw((size_t)numel);
            if (std::fread(raw.data(), 2, (size_t)numel, f) != (size_t)numel) fail(MI_ERR_IO, "'%s': short read", name.c_str());
            to_f32(raw.data(), t.dtype, numel, out.data(), name.c_str());
        }
        return out;
    }
};

// ------------------------------------------------------------------ Burn named-MessagePack record
struct BurnMpk : WeightFile {
    struct Raw {  // a tensor as found in the file
        std::string path, dtype;
        std::vector<int64_t> shape;
        const uint8_t* data = nullptr;  // dtype bytes, or (legacy "value" arrays) nullptr with `values` filled
        size_t bytes = 0;
        std::vector<float> values;
    };
    struct Mapped { size_t raw; bool transpose; TensorInfo info; };
    const uint8_t* base = nullptr;
    size_t size = 0;
    std::vector<Raw> raws;
    std::map<std::string, Mapped> mapped;
    std::vector<std::string> order;
    std::string path_;

    explicit BurnMpk(const char* path) : path_(path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) fail(MI_ERR_IO, "cannot open weights file '%s'", path);
        struct stat sb;
        if (fstat(fd, &sb) != 0 || sb.st_size <= 0) { ::close(fd); fail(MI_ERR_IO, "'%s' is empty or unreadable", path); }
        size = (size_t)sb.st_size;
        void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) fail(MI_ERR_IO, "cannot map '%s'", path);
        base = static_cast<const uint8_t*>(p);
        const uint8_t* c = base;
        try {
            walk(c, base + size, "", 0);
            map_names();
        } catch (...) {
            munmap(const_cast<uint8_t*>(base), size);
            base = nullptr;
            throw;
        }
    }
    ~BurnMpk() override { if (base) munmap(const_cast<uint8_t*>(base), size); }

    // ---- MessagePack ------------------------------------------------------------------------------------
    [[noreturn]] void bad(const char* what) const { fail(MI_ERR_IO, "'%s' is not a Burn MessagePack record (%s)", path_.c_str(), what); }
    static uint64_t be(const uint8_t* p, int n) { uint64_t v = 0; for (int i = 0; i < n; ++i) v = (v << 8) | p[i]; return v; }
    void need(const uint8_t* c, const uint8_t* e, size_t n) const { if ((size_t)(e - c) < n) bad("truncated"); }

    struct Val {  // a decoded scalar / header
        enum Kind { NIL, BOOL, INT, FLT, STR, BIN, ARR, MAP, EXT } kind = NIL;
        int64_t i = 0; double f = 0; const uint8_t* p = nullptr; size_t n = 0;  // STR/BIN/EXT: bytes; ARR/MAP: element count
    };
    Val head(const uint8_t*& c, const uint8_t* e) const {
        need(c, e, 1);
        const uint8_t b = *c++;
        Val v;
        auto blob = [&](Val::Kind k, size_t n) { need(c, e, n); v.kind = k; v.p = c; v.n = n; c += n; };
        if (b <= 0x7f) { v.kind = Val::INT; v.i = b; }
        else if (b >= 0xe0) { v.kind = Val::INT; v.i = (int8_t)b; }
        else if ((b & 0xf0) == 0x80) { v.kind = Val::MAP; v.n = b & 0x0f; }
        else if ((b & 0xf0) == 0x90) { v.kind = Val::ARR; v.n = b & 0x0f; }
        else if ((b & 0xe0) == 0xa0) blob(Val::STR, b & 0x1f);
        else switch (b) {
            case 0xc0: v.kind = Val::NIL; break;
            case 0xc2: v.kind = Val::BOOL; v.i = 0; break;
            case 0xc3: v.kind = Val::BOOL; v.i = 1; break;
            case 0xc4: { need(c, e, 1); const size_t n = *c++; blob(Val::BIN, n); } break;
            case 0xc5: { need(c, e, 2); const size_t n = be(c, 2); c += 2; blob(Val::BIN, n); } break;
            case 0xc6: { need(c, e, 4); const size_t n = be(c, 4); c += 4; blob(Val::BIN, n); } break;
            case 0xc7: { need(c, e, 2); const size_t n = *c; c += 2; blob(Val::EXT, n); } break;
            case 0xc8: { need(c, e, 3); const size_t n = be(c, 2); c += 3; blob(Val::EXT, n); } break;
            case 0xc9: { need(c, e, 5); const size_t n = be(c, 4); c += 5; blob(Val::EXT, n); } break;
            case 0xca: { need(c, e, 4); uint32_t u = (uint32_t)be(c, 4); c += 4; float f; std::memcpy(&f, &u, 4); v.kind = Val::FLT; v.f = f; } break;
            case 0xcb: { need(c, e, 8); uint64_t u = be(c, 8); c += 8; double d; std::memcpy(&d, &u, 8); v.kind = Val::FLT; v.f = d; } break;
            case 0xcc: need(c, e, 1); v.kind = Val::INT; v.i = *c; c += 1; break;
            case 0xcd: need(c, e, 2); v.kind = Val::INT; v.i = (int64_t)be(c, 2); c += 2; break;
            case 0xce: need(c, e, 4); v.kind = Val::INT; v.i = (int64_t)be(c, 4); c += 4; break;
            case 0xcf: need(c, e, 8); v.kind = Val::INT; v.i = (int64_t)be(c, 8); c += 8; break;
            case 0xd0: need(c, e, 1); v.kind = Val::INT; v.i = (int8_t)*c; c += 1; break;
            case 0xd1: need(c, e, 2); v.kind = Val::INT; v.i = (int16_t)be(c, 2); c += 2; break;
            case 0xd2: need(c, e, 4); v.kind = Val::INT; v.i = (int32_t)be(c, 4); c += 4; break;
            case 0xd3: need(c, e, 8); v.kind = Val::INT; v.i = (int64_t)be(c, 8); c += 8; break;
            case 0xd4: case 0xd5: case 0xd6: case 0xd7: case 0xd8: { const size_t n = (size_t)1 << (b - 0xd4); need(c, e, 1); c += 1; blob(Val::EXT, n); } break;
            case 0xd9: { need(c, e, 1); const size_t n = *c++; blob(Val::STR, n); } break;
            case 0xda: { need(c, e, 2); const size_t n = be(c, 2); c += 2; blob(Val::STR, n); } break;
            case 0xdb: { need(c, e, 4); const size_t n = be(c, 4); c += 4; blob(Val::STR, n); } break;
            case 0xdc: need(c, e, 2); v.kind = Val::ARR; v.n = be(c, 2); c += 2; break;
            case 0xdd: need(c, e, 4); v.kind = Val::ARR; v.n = be(c, 4); c += 4; break;
            case 0xde: need(c, e, 2); v.kind = Val::MAP; v.n = be(c, 2); c += 2; break;
            case 0xdf: need(c, e, 4); v.kind = Val::MAP; v.n = be(c, 4); c += 4; break;
            default: bad("reserved type byte");
        }
        return v;
    }
    void skip(const uint8_t*& c, const uint8_t* e, int depth) const {
        if (depth > 64) bad("nesting too deep");
        const Val v = head(c, e);
        if (v.kind == Val::ARR) for (size_t i = 0; i < v.n; ++i) skip(c, e, depth + 1);
        else if (v.kind == Val::MAP) for (size_t i = 0; i < 2 * v.n; ++i) skip(c, e, depth + 1);
    }
    // Walks one value.  A map that holds "shape" together with "bytes" (or "value") is a tensor; every other map is a
    // module / parameter level whose keys extend the path ("param" and "item" levels are not part of the name).
    void walk(const uint8_t*& c, const uint8_t* e, const std::string& path, int depth) {
        if (depth > 64) bad("nesting too deep");
        const uint8_t* start = c;
        const Val v = head(c, e);
        if (v.kind == Val::ARR) { for (size_t i = 0; i < v.n; ++i) walk(c, e, path + "." + std::to_string(i), depth + 1); return; }
        if (v.kind != Val::MAP) return;
        // first pass over the keys: is this a tensor?
        {
            const uint8_t* s = c;
            bool has_shape = false, has_data = false;
            for (size_t i = 0; i < v.n; ++i) {
                const Val k = head(s, e);
                if (k.kind == Val::STR) {
                    const std::string key((const char*)k.p, k.n);
                    has_shape |= key == "shape";
                    has_data |= key == "bytes" || key == "value";
                }
                skip(s, e, depth + 1);
            }
            if (has_shape && has_data) { tensor(c, e, v.n, path, depth); return; }
        }
        (void)start;
        for (size_t i = 0; i < v.n; ++i) {
            const Val k = head(c, e);
            std::string key = k.kind == Val::STR ? std::string((const char*)k.p, k.n) : (k.kind == Val::INT ? std::to_string(k.i) : "?");
            if (depth == 0 && key == "metadata") {  // {"float": "f32", "format": ..., "version": ...}
                const uint8_t* s = c;
                const Val mv = head(s, e);
                if (mv.kind == Val::MAP)
                    for (size_t j = 0; j < mv.n; ++j) {
                        const Val mk = head(s, e);
                        const uint8_t* before = s;
                        const Val mvv = head(s, e);
                        if (mk.kind == Val::STR && mvv.kind == Val::STR) meta["burn." + std::string((const char*)mk.p, mk.n)] = std::string((const char*)mvv.p, mvv.n);
                        else { s = before; skip(s, e, depth + 2); }
                    }
                skip(c, e, depth + 1);
                continue;
            }
            const bool transparent = key == "param" || (depth == 0 && key == "item");
            walk(c, e, transparent ? path : (path.empty() ? key : path + "." + key), depth + 1);
        }
    }
    void tensor(const uint8_t*& c, const uint8_t* e, size_t n_keys, const std::string& path, int depth) {
        Raw t;
        t.path = path;
        t.dtype = "F32";
        for (size_t i = 0; i < n_keys; ++i) {
            const Val k = head(c, e);
            const std::string key = k.kind == Val::STR ? std::string((const char*)k.p, k.n) : "";
            if (key == "shape") {
                const Val a = head(c, e);
                if (a.kind != Val::ARR) bad("tensor shape is not an array");
                for (size_t j = 0; j < a.n; ++j) { const Val d = head(c, e); if (d.kind != Val::INT || d.i < 0) bad("bad dimension"); t.shape.push_back(d.i); }
            } else if (key == "bytes") {
                const uint8_t* s = c;
                const Val b = head(c, e);
                if (b.kind == Val::BIN) { t.data = b.p; t.bytes = b.n; }
                else if (b.kind == Val::ARR) {  // a sequence of u8 (serde without serde_bytes): copy out
                    t.values.clear();
                    std::vector<uint8_t> tmp(b.n);
                    for (size_t j = 0; j < b.n; ++j) { const Val x = head(c, e); if (x.kind != Val::INT) bad("bad byte"); tmp[j] = (uint8_t)x.i; }
                    owned.emplace_back(std::move(tmp));
                    t.data = owned.back().data(); t.bytes = owned.back().size();
                } else { c = s; bad("tensor bytes are neither bin nor array"); }
            } else if (key == "value") {
                const Val a = head(c, e);
                if (a.kind != Val::ARR) bad("tensor value is not an array");
                t.values.resize(a.n);
                for (size_t j = 0; j < a.n; ++j) { const Val x = head(c, e); t.values[j] = x.kind == Val::FLT ? (float)x.f : x.kind == Val::INT ? (float)x.i : (bad("bad value"), 0.0f); }
            } else if (key == "dtype") {
                const uint8_t* s = c;
                const Val d = head(c, e);
                if (d.kind == Val::STR) t.dtype = std::string((const char*)d.p, d.n);
                else { c = s; skip(c, e, depth + 1); }  // an enum with payload (quantised): refused when read
            } else skip(c, e, depth + 1);
        }
        for (auto& ch : t.dtype) ch = (char)std::toupper((unsigned char)ch);
        int64_t numel = 1;
        for (auto d : t.shape) numel *= d;
        if (t.values.empty() && t.data) {
            const size_t esz = t.dtype == "F32" ? 4 : (t.dtype == "F16" || t.dtype == "BF16") ? 2 : 0;
            if (esz && t.bytes != (size_t)numel * esz) fail(MI_ERR_IO, "'%s': tensor '%s' holds %zu bytes for %lld %s elements", path_.c_str(), path.c_str(), t.bytes, (long long)numel, t.dtype.c_str());
        } else if ((int64_t)t.values.size() != numel) fail(MI_ERR_IO, "'%s': tensor '%s' holds %zu values for shape of %lld", path_.c_str(), path.c_str(), t.values.size(), (long long)numel);
        raws.push_back(std::move(t));
    }
    std::vector<std::vector<uint8_t>> owned;

    // ---- names ----------------------------------------------------------------------------------------------
    static std::vector<int64_t> squeeze(const std::vector<int64_t>& s) {
        std::vector<int64_t> o;
        for (auto d : s) if (d != 1) o.push_back(d);
        if (o.empty()) o.push_back(1);
        return o;
    }
    std::string inventory() const {
        std::string s;
        for (size_t i = 0; i < raws.size() && i < 40; ++i) {
            s += (i ? ", " : "") + raws[i].path + "[";
            for (size_t j = 0; j < raws[i].shape.size(); ++j) s += (j ? "x" : "") + std::to_string(raws[i].shape[j]);
            s += "]";
        }
        if (raws.size() > 40) s += ", ... (" + std::to_string(raws.size()) + " tensors)";
        return s;
    }
    void put(const std::string& name, size_t raw, bool transpose, std::vector<int64_t> shape) {
        if (mapped.count(name)) fail(MI_ERR_UNSUPPORTED, "'%s': two tensors map to '%s'", path_.c_str(), name.c_str());
        Mapped m{raw, transpose, {}};
        m.info.dtype = raws[raw].dtype;
        m.info.shape = std::move(shape);
        mapped[name] = m;
        order.push_back(name);
    }
    static std::string parent_of(const std::string& path) {
        const size_t dot = path.find_last_of('.');
        return dot == std::string::npos ? std::string() : path.substr(0, dot);
    }
    void map_names() {
        if (raws.empty()) bad("no tensor inside");
        // (1) a record that already carries the Hugging Face names
        bool named = false;
        for (const Raw& r : raws) named |= r.path.find("vision_model.") != std::string::npos || r.path.find("text_model.") != std::string::npos;
        if (named) {
            for (size_t i = 0; i < raws.size(); ++i) put(raws[i].path, i, false, raws[i].shape);
            return;
        }
        // (2) shape, module and order (see the header of this file)
        size_t conv = raws.size();
        for (size_t i = 0; i < raws.size(); ++i)
            if (raws[i].shape.size() == 4 && raws[i].shape[1] == 3 && raws[i].shape[2] == raws[i].shape[3]) {
                if (conv != raws.size()) fail(MI_ERR_UNSUPPORTED, "'%s': more than one [D,3,P,P] tensor — ambiguous: %s", path_.c_str(), inventory().c_str());
                conv = i;
            }
        if (conv == raws.size()) fail(MI_ERR_UNSUPPORTED, "'%s': no [D,3,P,P] patch-embedding tensor — not a CLIP vision tower record: %s", path_.c_str(), inventory().c_str());
        const int64_t D = raws[conv].shape[0];
        // Leaves that cannot be a tensor of the tower are set aside (and listed), not filed under a role: the graph the
        // reference builds is an opset-16 export (clip/scripts/upgrade_opset.py:9-28), whose LayerNorms are decomposed —
        // the record then also carries what the decomposition's Pow / Add / Sqrt / Mul nodes and the attention's scale and
        // the QuickGELU's 1.702 hold (rank-0 / [1] floats), the position ids and Reshape shapes (integers).
        //   * a dtype that is not F32 / F16 / BF16;   * fewer elements than the smallest tower dimension can have (< 8).
        std::vector<char> skip_leaf(raws.size(), 0);
        for (size_t i = 0; i < raws.size(); ++i) {
            int64_t numel = 1;
            for (auto d : raws[i].shape) numel *= d;
            const bool is_float = raws[i].dtype == "F32" || raws[i].dtype == "F16" || raws[i].dtype == "BF16";
            if (i != conv && (!is_float || numel < 8)) { skip_leaf[i] = 1; skipped.push_back(i); }
        }
        std::vector<size_t> vecD, vecF, sq, wide, tall, pos, other;
        int64_t FF = 0;
        for (size_t i = 0; i < raws.size(); ++i) {
            if (i == conv || skip_leaf[i]) continue;
            const std::vector<int64_t> s = squeeze(raws[i].shape);
            if (s.size() == 1 && s[0] == D) vecD.push_back(i);
            else if (s.size() == 1) { vecF.push_back(i); }
            else if (s.size() == 2 && s[0] == D && s[1] == D) sq.push_back(i);
            else if (s.size() == 2 && (s[0] == D || s[1] == D)) {
                const int64_t o = s[0] == D ? s[1] : s[0];
                const int64_t g = (int64_t)std::llround(std::sqrt((double)(o - 1)));
                if (s[1] == D && g * g + 1 == o && o != D) pos.push_back(i);         // [S, D], S - 1 a square: positions
                else other.push_back(i);
            } else fail(MI_ERR_UNSUPPORTED, "'%s': tensor '%s' fits no role in a CLIP vision tower: %s", path_.c_str(), raws[i].path.c_str(), inventory().c_str());
        }
        if (sq.size() % 4 != 0 || sq.empty()) fail(MI_ERR_UNSUPPORTED, "'%s': %zu square [D,D] matrices, expected 4 per layer: %s", path_.c_str(), sq.size(), inventory().c_str());
        const size_t L = sq.size() / 4;
        // fc1 / fc2 / projection among `other`: two of every layer share one FF, the projection is the odd one out
        std::map<int64_t, size_t> count_by_o;
        for (size_t i : other) { const auto s = squeeze(raws[i].shape); ++count_by_o[s[0] == D ? s[1] : s[0]]; }
        int64_t E = 0;
        for (auto& kv : count_by_o) {
            if (kv.second == 2 * L) FF = kv.first;
            else if (kv.second == 1) E = kv.first;
        }
        if (L == 1 && count_by_o.size() == 1 && count_by_o.begin()->second == 3) FF = E = count_by_o.begin()->first;  // FF == E, one layer
        // a 1-D float vector that is neither [D] nor [FF] belongs to nothing in the tower: set aside too
        if (FF) {
            std::vector<size_t> keep;
            for (size_t i : vecF) {
                if (squeeze(raws[i].shape)[0] == FF) keep.push_back(i);
                else { skip_leaf[i] = 1; skipped.push_back(i); }
            }
            vecF.swap(keep);
        }
        // A bias that sits in the same module as a matrix (burn-import's Linear: {weight [in,out], bias [out]}) is BOUND to
        // that matrix, wherever the module stands in the record; every other [D]-vector is a bare constant — the
        // class embedding, the affine parameters of a decomposed LayerNorm (Mul by gamma, then Add beta: graph order), the
        // gamma / beta of a fused LayerNorm module, or the bias of a MatMul + Add pair that was not coalesced into a Linear —
        // and takes its place in graph order among the bare ones.
        std::map<std::string, size_t> matrix_of_module;   // module path -> raw index of its 2-D tensor
        std::map<std::string, int> matrices_in_module;
        for (const std::vector<size_t>* grp : {&sq, &other})
            for (size_t i : *grp) { const std::string par = parent_of(raws[i].path); matrix_of_module[par] = i; ++matrices_in_module[par]; }
        std::map<size_t, size_t> bias_of_matrix;          // matrix raw index -> bias raw index
        auto bind = [&](std::vector<size_t>& vecs, int64_t want_out_or_any) {
            std::vector<size_t> bare;
            for (size_t i : vecs) {
                const std::string par = parent_of(raws[i].path);
                auto it = par.empty() ? matrix_of_module.end() : matrix_of_module.find(par);
                if (it != matrix_of_module.end() && matrices_in_module[par] == 1 && !bias_of_matrix.count(it->second)) {
                    const auto ms = squeeze(raws[it->second].shape);
                    const int64_t len = squeeze(raws[i].shape)[0];
                    if (ms[0] == len || ms[1] == len) { bias_of_matrix[it->second] = i; continue; }
                }
                bare.push_back(i);
            }
            (void)want_out_or_any;
            vecs.swap(bare);
        };
        bind(vecD, D);
        bind(vecF, FF);
        const size_t bound = bias_of_matrix.size();
        const bool all_bound = bound == 6 * L, none_bound = bound == 0;   // q, k, v, out, fc1, fc2 per layer (the projection has no bias)
        const size_t want_bare_D = all_bound ? 4 * L + 5 : 9 * L + 5, want_bare_F = all_bound ? 0 : L;
        if (!FF || !E || other.size() != 2 * L + 1 || pos.size() != 1 || !(all_bound || none_bound) || vecD.size() != want_bare_D || vecF.size() != want_bare_F)
            fail(MI_ERR_UNSUPPORTED,
                 "'%s': the tensor inventory does not match a CLIP vision tower in graph order (layers %zu from the [D,D] count; "
                 "biases bound to a matrix of their module %zu, want 0 or %zu; bare [D]-vectors %zu, want %zu; bare fc1 biases %zu, want %zu; "
                 "fc/projection matrices %zu, want %zu; position tables %zu, want 1; leaves set aside %zu): %s",
                 path_.c_str(), L, bound, 6 * L, vecD.size(), want_bare_D, vecF.size(), want_bare_F, other.size(), 2 * L + 1, pos.size(),
                 skipped.size(), inventory().c_str());
        const std::string v = "vision_model.";
        put(v + "embeddings.patch_embedding.weight", conv, false, raws[conv].shape);
        put(v + "embeddings.position_embedding.weight", pos[0], false, squeeze(raws[pos[0]].shape));
        size_t vd = 0, vf = 0, q = 0, o = 0;
        auto vec = [&](const std::string& name) { put(name, vecD[vd], false, {D}); ++vd; };
        // Burn keeps a Linear weight [d_in, d_out]; PyTorch (and this library) [d_out, d_in]
        auto lin = [&](const std::string& name, size_t raw, int64_t d_out, int64_t d_in) {
            const auto s = squeeze(raws[raw].shape);
            const bool burn = d_out == d_in ? true : (s[0] == d_in && s[1] == d_out);
            if (!burn && !(s[0] == d_out && s[1] == d_in)) fail(MI_ERR_UNSUPPORTED, "'%s': '%s' is not a %lld x %lld matrix either way", path_.c_str(), raws[raw].path.c_str(), (long long)d_out, (long long)d_in);
            put(name, raw, burn, {d_out, d_in});
        };
        // weight + bias of one linear layer: the bias bound to the matrix, or the next bare vector of its length
        auto linear = [&](const std::string& base, size_t raw, int64_t d_out, int64_t d_in) {
            lin(base + ".weight", raw, d_out, d_in);
            auto it = bias_of_matrix.find(raw);
            if (it != bias_of_matrix.end()) {
                if (squeeze(raws[it->second].shape)[0] != d_out) fail(MI_ERR_UNSUPPORTED, "'%s': '%s' is not a bias of %lld elements", path_.c_str(), raws[it->second].path.c_str(), (long long)d_out);
                put(base + ".bias", it->second, false, {d_out});
            } else if (d_out == D) vec(base + ".bias");
            else put(base + ".bias", vecF[vf++], false, {FF});
        };
        vec(v + "embeddings.class_embedding");
        vec(v + "pre_layrnorm.weight");
        vec(v + "pre_layrnorm.bias");
        for (size_t l = 0; l < L; ++l) {
            const std::string p = v + "encoder.layers." + std::to_string(l) + ".";
            vec(p + "layer_norm1.weight"); vec(p + "layer_norm1.bias");
            for (const char* n : {"q_proj", "k_proj", "v_proj", "out_proj"}) linear(p + "self_attn." + n, sq[q++], D, D);
            vec(p + "layer_norm2.weight"); vec(p + "layer_norm2.bias");
            linear(p + "mlp.fc1", other[o++], FF, D);
            linear(p + "mlp.fc2", other[o++], D, FF);
        }
        vec(v + "post_layernorm.weight");
        vec(v + "post_layernorm.bias");
        lin("visual_projection.weight", other[o++], E, D);
    }
    std::vector<size_t> skipped;   // raw indices of the leaves that were set aside (mi_weights_list prints them)
    std::vector<std::string> skipped_lines() const override {
        std::vector<std::string> out;
        for (size_t i : skipped) {
            std::string l = "(set aside) " + raws[i].path + " " + raws[i].dtype + " [";
            for (size_t j = 0; j < raws[i].shape.size(); ++j) l += (j ? "," : "") + std::to_string(raws[i].shape[j]);
            out.push_back(l + "]");
        }
        return out;
    }

    bool has(const std::string& name) const override { return mapped.count(name) != 0; }
    const TensorInfo& info(const std::string& name) const override {
        auto it = mapped.find(name);
        if (it == mapped.end()) fail(MI_ERR_IO, "weights file lacks tensor '%s' (Burn record: %s)", name.c_str(), inventory().c_str());
        return it->second.info;
    }
    std::vector<std::string> names() const override { return order; }
    std::vector<float> read(const std::string& name, int64_t numel) const override {
        auto it = mapped.find(name);
        if (it == mapped.end()) fail(MI_ERR_IO, "weights file lacks tensor '%s'", name.c_str());
        const Raw& r = raws[it->second.raw];
        int64_t n = 1;
        for (auto d : r.shape) n *= d;
        if (n != numel) fail(MI_ERR_IO, "tensor '%s' ('%s') has %lld elements, expected %lld", name.c_str(), r.path.c_str(), (long long)n, (long long)numel);
        std::vector<float> out((size_t)n);
        if (!r.values.empty()) out = r.values;
        else to_f32(r.data, r.dtype, n, out.data(), name.c_str());
        if (it->second.transpose) {
            const int64_t rows = it->second.info.shape[0], cols = it->second.info.shape[1];  // target [rows = out][cols = in]; stored [in][out]
            std::vector<float> tr((size_t)n);
            for (int64_t i = 0; i < cols; ++i)
                for (int64_t o = 0; o < rows; ++o) tr[(size_t)o * cols + i] = out[(size_t)i * rows + o];
            out.swap(tr);
        }
        return out;
    }
};

// by content: safetensors starts with a u64 header length followed by '{'; a Burn record starts with a MessagePack map
inline std::unique_ptr<WeightFile> open_weights(const char* path) {
    FILE* f = std::fopen(path, "rb");
    if (!f) fail(MI_ERR_IO, "cannot open weights file '%s'", path);
    unsigned char b[9] = {0};
    const size_t got = std::fread(b, 1, 9, f);
    std::fclose(f);
    if (got < 9) fail(MI_ERR_IO, "'%s' is too short to be a weights file", path);
    const bool mpk = (b[0] & 0xf0) == 0x80 || b[0] == 0xde || b[0] == 0xdf;
    if (mpk && b[8] != '{') return std::unique_ptr<WeightFile>(new BurnMpk(path));
    return std::unique_ptr<WeightFile>(new SafeTensors(path));
}

}  // namespace mi
