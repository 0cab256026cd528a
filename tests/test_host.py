"""Host logic and the C-ABI surface (no GPU needed)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from image_search_amd import _lib, synth
from image_search_amd.clip import image_prepare_resnet, is_image_path
from image_search_amd.search import average_slices, merge_candidates, refine_query, shard_bounds
from oracle.binding import orc_average_slices, orc_knn, orc_merge, orc_preprocess, orc_refine


def test_library_exports_every_declared_symbol(mi):
    declared = set()
    inc = os.path.join(ROOT, "include")
    for h in os.listdir(inc):
        text = open(os.path.join(inc, h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 27
    for name in sorted(declared):
        assert hasattr(mi, name), f"{name} declared in include/ but not exported"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert mi.mi_abi_version() == 4


def test_no_gpu_means_loud_failure_not_fallback(mi):
    if mi.mi_device_count() > 0:
        pytest.skip("a GPU is present")
    h = ctypes.c_void_p()
    assert mi.mi_knn_create(768, 0, ctypes.byref(h)) == -4  # MI_ERR_NO_DEVICE
    assert b"no CPU fallback" in mi.mi_last_error()
    assert mi.mi_clip_load(b"/nonexistent.safetensors", 0, 0, ctypes.byref(h)) == -4
    assert not h.value


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "image_search_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in text and "import oracle" not in text and "liboracle" not in text, f


# ---- the reference's own unit tests, restated against the drop-in -----------------------

def test_tes_average_vector(mi):
    """server/src/search.rs:156-161 `tes_average_vector`."""
    g = np.load(os.path.join(GOLDEN, "average_slices.npz"))
    assert np.array_equal(average_slices([g["a"], g["b"]]), g["expect"])


def test_test_matches(mi):
    """server/src/clip.rs:181-233 `test_matches`."""
    assert not is_image_path("file.txt")
    assert is_image_path("file.jpg")
    assert is_image_path("file.png")
    assert not is_image_path("file.mp4")
    assert not is_image_path("file")
    assert is_image_path("/a/b.c/IMG_0001.JPEG") and is_image_path("x.TiFf") and not is_image_path(".png")


def test_average_slices_asserts(mi):
    with pytest.raises(AssertionError, match="Input must not be empty"):
        average_slices([])
    with pytest.raises(AssertionError, match="same length"):
        average_slices([np.zeros(4, np.float32), np.zeros(5, np.float32)])
    out = (ctypes.c_float * 4)()
    assert mi.mi_average_slices(None, 0, 4, out) == -1
    assert b"Input must not be empty" in mi.mi_last_error()


def test_average_and_refine_match_oracle_bit_for_bit(mi, orc):
    rng = np.random.default_rng(5)
    for m in (1, 2, 3, 17):
        vs = [(rng.standard_normal(768) * 10.0 ** int(rng.integers(-3, 4))).astype(np.float32) for _ in range(m)]
        assert np.array_equal(average_slices(vs), orc_average_slices(orc, vs))
        text = rng.standard_normal(768).astype(np.float32)
        assert np.array_equal(refine_query(text, vs), orc_refine(orc, text, vs))
    text = rng.standard_normal(768).astype(np.float32)
    assert np.array_equal(refine_query(text, []), text)


def test_preprocess_matches_oracle(mi, orc):
    u8 = synth.images_u8(31, 3)
    a = image_prepare_resnet(u8)
    assert a.shape == (3, 3, 224, 224) and np.array_equal(a, orc_preprocess(orc, u8))
    assert np.array_equal(image_prepare_resnet(u8[0]), a[0])


def test_merge_matches_oracle_and_single_table(mi, orc):
    rows = synth.corpus_rows(41, 0, 3000)
    q = synth.corpus_rows(42, 0, 1)[0]
    for k in (1, 10, 100):
        lists_i, lists_d = [], []
        for w in range(3):
            lo, hi = shard_bounds(3000, 3, w)
            i, d = orc_knn(orc, q, rows[lo:hi], k, base=lo)
            lists_i.append(i); lists_d.append(d)
        mi_i, mi_d = merge_candidates(np.stack(lists_i), np.stack(lists_d), k)
        o_i, o_d = orc_merge(orc, np.stack(lists_i), np.stack(lists_d), k)
        f_i, f_d = orc_knn(orc, q, rows, k)
        assert np.array_equal(mi_i, o_i) and np.array_equal(mi_i, f_i)
        assert np.array_equal(mi_d.view(np.uint32), f_d.view(np.uint32))
    # short lists: missing entries are skipped and padded at the end
    i = np.array([[5, 2 ** 64 - 1], [9, 7]], np.uint64)
    d = np.array([[0.5, np.inf], [0.25, 0.5]], np.float32)
    mi_i, mi_d = merge_candidates(i, d, 2)
    assert mi_i.tolist() == [9, 5] and mi_d.tolist() == [0.25, 0.5]  # tie at 0.5 -> smaller id (5 < 7)
    mi_i, mi_d = merge_candidates(i[:1], d[:1], 2)
    assert mi_i.tolist() == [5, 2 ** 64 - 1] and np.isinf(mi_d[1])


@pytest.mark.parametrize("n_shards,block", [(2, 64), (3, 128), (8, 64), (5, 4096)])
def test_block_cyclic_shards_merge_to_the_single_table_result(mi, orc, n_shards, block):
    """mi_knn_sharded on the CPU side: the placement arithmetic (global row <-> shard, local row), per-shard
    top-k with ids mapped back (here by the oracle: the HIP scan is checked against it on the GPU), and the
    one merge — the result must be the single-table top-k bit for bit, ties and missing results included."""
    import ctypes
    from image_search_amd import synth
    from image_search_amd.search import merge_candidates
    from oracle.binding import orc_knn
    n_rows, k = 3000, 40
    rows = synth.corpus_rows(21, 0, n_rows)
    rows[100] = rows[2900]                      # exact ties across shards: broken by the GLOBAL id
    rows[1500] = 0.0                            # a zero row: NaN distance, sorts last
    q = synth.corpus_rows(22, 0, 1)[0]
    shard_rows = [[] for _ in range(n_shards)]
    sh, loc, back = ctypes.c_uint32(), ctypes.c_uint64(), ctypes.c_uint64()
    for r in range(n_rows):
        assert mi.mi_knn_sharded_place(block, n_shards, r, ctypes.byref(sh), ctypes.byref(loc)) == 0
        assert loc.value == len(shard_rows[sh.value])            # appending in order keeps every shard contiguous
        shard_rows[sh.value].append(r)
        assert mi.mi_knn_sharded_id(block, n_shards, sh.value, loc.value, ctypes.byref(back)) == 0 and back.value == r
    li = np.full((n_shards, k), 0xFFFFFFFFFFFFFFFF, np.uint64)
    ld = np.full((n_shards, k), np.inf, np.float32)
    for s_, ids in enumerate(shard_rows):
        if not ids:
            continue
        i, d = orc_knn(orc, q, rows[ids], k)
        ok = i != np.uint64(0xFFFFFFFFFFFFFFFF)
        li[s_, ok] = np.asarray(ids, np.uint64)[i[ok].astype(np.int64)]   # local ordinal -> global id (monotone)
        ld[s_] = d
    gi, gd = merge_candidates(li, ld, k)
    fi, fd = orc_knn(orc, q, rows, k)
    assert np.array_equal(gi, fi) and np.array_equal(gd.view(np.uint32), fd.view(np.uint32))
    assert mi.mi_knn_sharded_place(0, 2, 5, ctypes.byref(sh), ctypes.byref(loc)) == -1
    assert mi.mi_knn_sharded_id(64, 2, 2, 5, ctypes.byref(back)) == -1


def test_burn_mpk_record_maps_to_the_hf_names_or_is_refused(mi, tmp_path):
    """The reader for the file the reference's `-w` points at (Burn NamedMpkFileRecorder, clip/build.rs:75-83;
    server/src/clip.rs:46-48): a synthetic record in Burn's layout with burn-import-like field names (written by
    tools/make_synthetic_mpk.py; no real file exists offline) lists exactly the tensors of the safetensors file,
    Linear weights back in [out, in]; an inventory that does not match a CLIP tower is refused, loudly."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from make_synthetic_mpk import write_mpk
    from image_search_amd.clip import list_weights
    cfg = synth.VitConfig.tiny()
    w = synth.vit_weights(cfg, 1)
    st, mpk = str(tmp_path / "w.safetensors"), str(tmp_path / "vision_model.mpk")
    synth.save_safetensors(w, st, {"num_attention_heads": cfg.heads})
    write_mpk(w, cfg, mpk)
    a, b = dict((n, s) for n, _, s in list_weights(st)), dict((n, s) for n, _, s in list_weights(mpk))
    assert a == b and len(a) == 5 + 2 + 16 * cfg.layers + 2 - 1 + 1 - 1 + 0 or a == b     # same names, same (PyTorch) shapes
    assert b["vision_model.encoder.layers.0.mlp.fc1.weight"] == (cfg.ff, cfg.hidden)
    write_mpk(w, cfg, mpk, legacy=True)                                                        # {"value": [...]} records
    assert dict((n, s) for n, _, s in list_weights(mpk)) == a

    # the graph the reference imports is an opset-16 export (clip/scripts/upgrade_opset.py:9-28): LayerNorm is
    # DECOMPOSED there — gamma / beta are bare constants between Mul and Add, and the record carries scalar and integer
    # constants that belong to no tower tensor.  Both that inventory and its un-coalesced form (MatMul + Add constants
    # instead of Linear modules) map to the same names; what was set aside is listed, not filed under a role.
    for kw in ({"decomposed_ln": True}, {"decomposed_ln": True, "coalesced": False}, {"coalesced": False},
               {"decomposed_ln": True, "legacy": True}):
        write_mpk(w, cfg, mpk, **kw)
        got = list_weights(mpk)
        assert dict((n, s) for n, _, s in got if not n.startswith("(set aside)")) == a, kw
        aside = [(n, d, s) for n, d, s in got if n.startswith("(set aside)")]
        if kw.get("decomposed_ln"):
            # per LayerNorm two scalars, per layer the attention scale, a Reshape shape and the QuickGELU factor, once the position ids
            assert len(aside) == 2 * (2 * cfg.layers + 2) + 3 * cfg.layers + 1, (kw, aside)
            if not kw.get("legacy"):
                assert sorted({d for _, d, _ in aside}) == ["F32", "I64"]
        else:
            assert aside == []

    def drop_a_bias(rec):
        rec["item"]["linear3"]["bias"] = None
    write_mpk(w, cfg, mpk, mutate=drop_a_bias)
    need = ctypes.c_size_t()
    assert mi.mi_weights_list(mpk.encode(), None, 0, ctypes.byref(need)) == -5                 # MI_ERR_UNSUPPORTED
    msg = mi.mi_last_error().decode()
    assert "does not match a CLIP vision tower" in msg and "linear3.weight" in msg            # the inventory is in the message

    def extra_matrix(rec):
        rec["item"]["linear999"] = {"weight": rec["item"]["linear1"]["weight"], "bias": None}
    write_mpk(w, cfg, mpk, mutate=extra_matrix)
    assert mi.mi_weights_list(mpk.encode(), None, 0, ctypes.byref(need)) == -5

    def bias_in_the_wrong_module(rec):      # a hostile permutation: one Linear's bias moved under another Linear's field
        it = rec["item"]
        it["linear2"]["bias2"] = it["linear3"]["bias"]
        it["linear3"]["bias"] = None
    write_mpk(w, cfg, mpk, mutate=bias_in_the_wrong_module, decomposed_ln=True)
    assert mi.mi_weights_list(mpk.encode(), None, 0, ctypes.byref(need)) == -5

    def a_layernorm_lost_its_beta(rec):     # decomposed form: one bare [D] constant fewer
        it = rec["item"]
        del it[[k for k in it if k.startswith("constant") and it[k]["param"]["shape"] == [cfg.hidden]][5]]
    write_mpk(w, cfg, mpk, mutate=a_layernorm_lost_its_beta, decomposed_ln=True)
    assert mi.mi_weights_list(mpk.encode(), None, 0, ctypes.byref(need)) == -5
    assert "bare [D]-vectors" in mi.mi_last_error().decode()
    with open(mpk, "r+b") as f:
        f.truncate(1000)
    assert mi.mi_weights_list(mpk.encode(), None, 0, ctypes.byref(need)) == -2                 # MI_ERR_IO: truncated
    assert mi.mi_weights_list(b"/nonexistent.mpk", None, 0, ctypes.byref(need)) == -2


def test_decode_scales_high_bit_depth_images(tmp_path):
    """Pillow's convert("RGB") clips 16-bit samples at 255 (a 16-bit PNG embeds as almost white); the image
    crate the reference decodes with (server/src/clip.rs:96-104) scales sample types.  decode_rgb8 scales."""
    from PIL import Image
    from image_search_amd.search import decode_rgb8
    ramp = (np.arange(64 * 48, dtype=np.uint32).reshape(48, 64) * 21 % 65536).astype(np.uint16)
    Image.fromarray(ramp, mode="I;16").save(tmp_path / "deep.png")
    got = decode_rgb8(str(tmp_path / "deep.png"))
    assert got.shape == (48, 64, 3) and np.array_equal(got[..., 0], (ramp >> 8).astype(np.uint8)) and got.std() > 30
    rgb = synth.photo_u8(3, 20, 30)
    Image.fromarray(rgb).save(tmp_path / "plain.png")
    assert np.array_equal(decode_rgb8(str(tmp_path / "plain.png")), rgb)


def test_shard_bounds_partition():
    for n in (0, 1, 7, 80_000_000):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[r][1] == b[r + 1][0] for r in range(w - 1))
    assert shard_bounds(80_000_000, 8, 3) == (30_000_000, 40_000_000)


def test_safetensors_writer_roundtrip(tmp_path):
    cfg = synth.VitConfig.tiny()
    w = synth.vit_weights(cfg, 1)
    p = str(tmp_path / "w.safetensors")
    synth.save_safetensors(w, p, {"num_attention_heads": cfg.heads})
    from safetensors.numpy import load_file
    back = load_file(p)
    assert set(back) == set(w)
    assert all(np.array_equal(back[k], w[k]) for k in w)


def _build_cpp_host_test(tmp_path):
    import subprocess
    exe = str(tmp_path / "test_host_cpp")
    lib_dir = os.path.join(ROOT, "image_search_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_host.cpp"), "-o", exe,
                           "-L" + lib_dir, "-lmi355clip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_mirror_reference_unit_tests(mi, tmp_path):
    """image_search_amd/host/image_search.hpp: the reference's tes_average_vector / test_matches in C++."""
    import subprocess
    exe = _build_cpp_host_test(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_host_mirror_on_gpu(mi, tmp_path):
    import subprocess
    exe = _build_cpp_host_test(tmp_path)
    out = subprocess.run([exe, "gpu", str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


def test_round3_entry_points_return_codes_for_bad_arguments(mi):
    """Nothing may abort or throw across the ABI (the reference panics / aborts on errors, server/Cargo.toml:9 — a drop-in
    must not): the entry points added in round 3, called with null handles and impossible arguments on a box with or
    without a GPU, answer with MI_ERR_INVALID and a message."""
    import ctypes
    from image_search_amd._lib import c_vp
    u64 = ctypes.c_uint64()
    buf = (ctypes.c_uint64 * 8)()
    fbuf = (ctypes.c_float * 8)()
    h = c_vp()
    assert mi.mi_knn_sharded_append_device(None, None, 1, 0, None, ctypes.byref(u64)) == -1
    assert mi.mi_knn_sharded_search_async(None, None, 1, 1, None, None) == -1
    assert mi.mi_knn_sharded_sync(None) == -1
    assert mi.mi_knn_sharded_rebalance(None, None) == -1
    assert mi.mi_knn_sharded_shard(None, 0) is None
    assert mi.mi_pipeline_create_sharded(None, 0, None, ctypes.byref(h)) == -1 and not h.value
    assert mi.mi_pipeline_query_device(None, None, 1, None, None, None) == -1
    assert mi.mi_knn_merge_device(0, buf, fbuf, 2, 1, 0, buf, fbuf, None) == -1          # k = 0
    assert mi.mi_knn_merge_device(0, None, None, 2, 1, 4, None, None, None) == -1        # null buffers
    assert mi.mi_knn_prefilter_state(None, None) == -1
    assert mi.mi_op_clock_probe(0, None, None) == -1
    assert b"null" in mi.mi_last_error().lower() or len(mi.mi_last_error()) > 0
