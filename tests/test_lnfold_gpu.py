"""The bf16 tower WITHOUT LayerNorm kernels in the layer loop (mi_clip_set_option "ln_fold"),
on a real MI355X through the C ABI.

Per-op: the two new epilogues of the persistent GEMM against a numpy restatement — bit-exact where the
accumulation is exact (small integers), within bf16 resolution on random data:
  EPI_LNF      out = act(rstd * (x W'^T) - mean * rstd * c + b')            (q/k/v, fc1)
  EPI_RESID24  x += bf16(x W^T + b) on the 24-bit planes, + per-row partial sums  (out_proj, fc2)
Tower: a 256-wide config and ViT-L/14 against the numpy oracle inside the same bf16 bound as the LayerNorm
tower, and the identities that must hold to the bit (chunking, CLS-only last layer, tile order, grids).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from image_search_amd import ops, synth
from image_search_amd.clip import PRECISION_BF16, Model
from oracle import vit_numpy

pytestmark = pytest.mark.gpu

F32 = np.float32


def bf16_round(a):
    u = np.ascontiguousarray(a, F32).view(np.uint32)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16).view(F32)


def enc24(x):
    """fp32 -> (hi u16, lo u8): b' = bits + 0x8080; hi = bf16(x) to nearest (ties away), 24 bits kept in all."""
    u = np.ascontiguousarray(x, F32).view(np.uint32) + np.uint32(0x8080)
    return (u >> 16).astype(np.uint16), ((u >> 8) & 0xFF).astype(np.uint8)


def dec24(hi, lo):
    return (((hi.astype(np.uint32) << 16) | (lo.astype(np.uint32) << 8)) - np.uint32(0x8000)).view(F32)


def hi_as_float(hi):
    return (hi.astype(np.uint32) << 16).view(F32)


def block_sums(x):
    """The epilogue's order: per 8-column lane group even and odd columns accumulate apart in column order, then
    even + odd; the four lane groups of a 32-column block add as (g0 + g1) + (g2 + g3).  fp32 throughout; the
    squares go through fma(e, e, q) (exact product, one rounding), restated in float64."""
    m, n = x.shape
    x = x.reshape(m, n // 32, 4, 4, 2).astype(F32)           # [row][block][lane group][pair][even/odd]
    s = x[..., 0, :].copy()
    q = (x[..., 0, :].astype(np.float64) ** 2).astype(F32)
    for i in range(1, 4):
        e = x[..., i, :]
        s = (s + e).astype(F32)
        q = (e.astype(np.float64) ** 2 + q.astype(np.float64)).astype(F32)
    S = (s[..., 0] + s[..., 1]).astype(F32)                  # [row][block][lane group]
    Q = (q[..., 0] + q[..., 1]).astype(F32)
    S = ((S[..., 0] + S[..., 1]).astype(F32) + (S[..., 2] + S[..., 3]).astype(F32)).astype(F32)
    Q = ((Q[..., 0] + Q[..., 1]).astype(F32) + (Q[..., 2] + Q[..., 3]).astype(F32)).astype(F32)
    return np.stack([S, Q], -1)                              # [row][block][2]


def stats_of(part, d, eps):
    """ln_stats_kernel's order: lane j of 16 adds blocks 2j and 2j + 1, the 16 lanes add as a balanced tree."""
    m, nb, _ = part.shape
    assert nb <= 32
    lanes = np.zeros((m, 16, 2), F32)
    pair = (part[:, 0::2, :] + part[:, 1::2, :]).astype(F32)
    lanes[:, :nb // 2] = pair
    while lanes.shape[1] > 1:
        lanes = (lanes[:, 0::2] + lanes[:, 1::2]).astype(F32)
    s, q = lanes[:, 0, 0], lanes[:, 0, 1]
    inv = F32(1.0 / d)
    mean = (s * inv).astype(F32)
    var = np.maximum((q * inv).astype(F32) - (mean * mean).astype(F32), F32(0)).astype(F32)
    rstd = (F32(1) / np.sqrt((var + F32(eps)).astype(F32)).astype(F32)).astype(F32)
    return np.stack([rstd, (-mean * rstd).astype(F32)], -1)


# ---- EPI_RESID24 ---------------------------------------------------------------------

@pytest.mark.parametrize("grid", ["3", "8", "256"])
@pytest.mark.parametrize("shape", [(1000, 256, 128), (2304, 512, 256), (700, 1024, 320), (3000, 1024, 128)])
def test_resid24_epilogue_exact_on_integers(built, monkeypatch, grid, shape):
    """Several tiles per workgroup, the quadrant tasks of a short last round, the prefetched old planes and the counted
    queue across tile boundaries: every output bit (both planes, partial sums, statistics) against numpy."""
    monkeypatch.setenv("MI_OP_GRID", grid)
    m, n, k = shape
    rng = np.random.default_rng(11)
    x = rng.integers(-2, 3, (m, k)).astype(F32)
    w = rng.integers(-1, 2, (n, k)).astype(F32)
    b = rng.integers(-3, 4, n).astype(F32)
    res = (rng.integers(-4000, 4001, (m, n)) / 16).astype(F32)   # 12 bits + 4 fraction bits: exact in 24 bits
    delta = x @ w.T + b
    assert np.abs(delta).max() <= 256
    want = (res + delta).astype(F32)
    got, hi, part, stats = ops.linear_resid24(x, w, b, res)
    hi_w, lo_w = enc24(want)
    assert np.array_equal(got.view(np.uint32), dec24(hi_w, lo_w).view(np.uint32)), (grid, shape)
    assert np.array_equal(hi.view(np.uint32), hi_as_float(hi_w).view(np.uint32))
    pw = block_sums(want)
    assert np.array_equal(part.view(np.uint32), pw.view(np.uint32))
    assert np.allclose(stats, stats_of(pw, n, 1e-5), rtol=2e-6, atol=0)


def test_resid24_epilogue_race_screen_at_the_tower_shapes(built):
    """The tower's own out_proj / fc2 shapes (half chunk: 129 row tiles x 4 column tiles, a split last round on all 256
    workgroups), integer operands so that every bit is known, 12 repetitions each: a stale plane, a prefetched row arriving
    late or a store reading a reused register would show as a changed word in some repetition."""
    rng = np.random.default_rng(21)
    for (m, n, k) in ((32896, 1024, 1024), (32896, 1024, 4096)):
        x = rng.integers(-1, 2, (m, k)).astype(F32)
        w = (rng.integers(-1, 2, (n, k)) * (rng.random((n, k)) < 0.05)).astype(F32)   # sparse: |acc| stays far below 256
        b = rng.integers(-3, 4, n).astype(F32)
        res = (rng.integers(-4000, 4001, (m, n)) / 16).astype(F32)
        delta = x @ w.T + b
        assert np.abs(delta).max() <= 256
        want = (res + delta).astype(F32)
        hi_w, lo_w = enc24(want)
        want_bits = dec24(hi_w, lo_w).view(np.uint32)
        pw = block_sums(want).view(np.uint32)
        for rep in range(12):
            got, hi, part, stats = ops.linear_resid24(x, w, b, res)
            assert np.array_equal(got.view(np.uint32), want_bits), (m, n, k, rep)
            assert np.array_equal(part.view(np.uint32), pw), (m, n, k, rep)


def test_resid24_epilogue_random_data_and_rounding(built):
    """Random operands: the delta is the bf16 of an fp32 accumulation (its order is the MFMA's), so compare within one
    bf16 ulp of the delta; the hi plane must be bf16(new x) to nearest and the planes must reproduce x to 24 bits."""
    rng = np.random.default_rng(12)
    m, n, k = 1500, 1024, 1024
    x = rng.standard_normal((m, k)).astype(F32)
    w = (rng.standard_normal((n, k)) * k ** -0.5).astype(F32)
    b = rng.standard_normal(n).astype(F32)
    res = (rng.standard_normal((m, n)) * 3).astype(F32)
    res[5, :] += 40.0                         # a row with a mean far from zero
    res[:, 7] *= 30.0                         # an outlier channel
    got, hi, part, stats = ops.linear_resid24(x, w, b, res)
    old = dec24(*enc24(res)).astype(np.float64)
    delta = bf16_round(x).astype(np.float64) @ bf16_round(w).astype(np.float64).T + b
    ref = old + delta
    assert np.abs(got - ref).max() <= 2.0 ** -8 * np.abs(delta).max() + 2.0 ** -16 * np.abs(ref).max()
    # the hi plane is the round-to-nearest bf16 of the stored value: |hi - x| <= half a bf16 ulp <= 2^-8 |x|, and it IS a bf16
    assert np.all(np.abs(hi.astype(np.float64) - got) <= 2.0 ** -8 * np.abs(got))
    assert not (hi.view(np.uint32) & 0xFFFF).any()
    assert np.array_equal(hi.view(np.uint32), hi_as_float(enc24(got)[0]).view(np.uint32))
    # statistics of the rows as stored
    mean = got.astype(np.float64).mean(1)
    var = got.astype(np.float64).var(1)
    rstd = 1 / np.sqrt(var + 1e-5)
    assert np.allclose(stats[:, 0], rstd, rtol=2e-4)
    assert np.allclose(stats[:, 1], -mean * rstd, rtol=2e-4, atol=2e-4)
    assert np.allclose(part[..., 0].sum(1), got.astype(np.float64).sum(1), rtol=1e-4, atol=1e-2)


# ---- EPI_LNF ---------------------------------------------------------------------------

@pytest.mark.parametrize("grid", ["3", "8", "256"])
def test_lnf_epilogue_exact_on_integers(built, monkeypatch, grid):
    monkeypatch.setenv("MI_OP_GRID", grid)
    monkeypatch.setenv("MI_OP_STORE_NT", "0" if grid == "8" else "1")   # the stores' cache policy is a hint: both exact
    rng = np.random.default_rng(13)
    for (m, n, k) in ((1000, 256, 128), (2304, 512, 256), (700, 3072, 192)):
        x = rng.integers(-2, 3, (m, k)).astype(F32)
        w = rng.integers(-1, 2, (n, k)).astype(F32)
        b = rng.integers(-3, 4, n).astype(F32)
        c = rng.integers(-2, 3, n).astype(F32)
        st = np.stack([rng.integers(1, 3, m), rng.integers(-2, 3, m)], -1).astype(F32)   # {a, b}: small integers
        ref = st[:, :1] * (x @ w.T) + st[:, 1:] * c + b
        assert np.abs(ref).max() <= 256
        assert np.array_equal(ops.linear_lnf(x, w, b, c, st, ops.EPI_LNF), ref), (grid, m, n, k)


@pytest.mark.parametrize("epi", [ops.EPI_LNF, ops.EPI_LNF_QGELU])
def test_lnf_epilogue_is_the_layernorm_then_the_linear(built, epi):
    """The algebra the option rests on: LN(x) W^T + b == rstd (x' W'^T - mean c) + b' with W' = W diag(gamma),
    c = row sums of bf16(W'), b' = W beta + b — on rows with a mean, a spread of scales and an outlier channel."""
    rng = np.random.default_rng(14)
    m, n, k = 900, 1024, 1024
    x = (rng.standard_normal((m, k)) * rng.uniform(0.3, 3.0, (m, 1)) + rng.standard_normal((m, 1))).astype(F32)
    x[:, 11] *= 25.0
    gamma = (1 + 0.1 * rng.standard_normal(k)).astype(F32)
    beta = (0.05 * rng.standard_normal(k)).astype(F32)
    w = (rng.standard_normal((n, k)) * k ** -0.5).astype(F32)
    b = (0.1 * rng.standard_normal(n)).astype(F32)
    x64 = x.astype(np.float64)
    mean, var = x64.mean(1, keepdims=True), x64.var(1, keepdims=True)
    rstd = 1 / np.sqrt(var + 1e-5)
    ref = ((x64 - mean) * rstd * gamma + beta) @ w.astype(np.float64).T + b
    if epi == ops.EPI_LNF_QGELU:
        ref = ref / (1 + np.exp(-1.702 * ref))
    wf = (w * gamma).astype(F32)
    c = bf16_round(wf).astype(np.float64).sum(1).astype(F32)
    bf = (w.astype(np.float64) @ beta + b).astype(F32)
    st = np.concatenate([rstd, -mean * rstd], 1).astype(F32)
    got = ops.linear_lnf(x, wf, bf, c, st, epi)
    # the LayerNorm tower's own error on the same rows: bf16 of the normalised rows against bf16 weights
    y = bf16_round(((x64 - mean) * rstd * gamma + beta).astype(F32)).astype(np.float64)
    base = y @ bf16_round(w).astype(np.float64).T + b
    if epi == ops.EPI_LNF_QGELU:
        base = base / (1 + np.exp(-1.702 * base))
    scale = float(np.sqrt((ref ** 2).mean()))
    e_fold = float(np.sqrt(((got - ref) ** 2).mean())) / scale
    e_base = float(np.sqrt(((bf16_round(base.astype(F32)) - ref) ** 2).mean())) / scale
    print(f"rms error / rms: folded {e_fold:.3e}, LayerNorm-then-linear {e_base:.3e}")
    assert e_fold <= 2.0 * e_base + 1e-4
    assert np.abs(got - ref).max() <= 1.5e-2 * np.abs(ref).max()


# ---- the tower ---------------------------------------------------------------------------

def _model(tmp, cfg, seed=3):
    w = synth.vit_weights(cfg, seed)
    path = str(tmp / f"vit_{cfg.hidden}_{cfg.layers}.safetensors")
    synth.save_safetensors(w, path, {"num_attention_heads": cfg.heads})
    return w, path


@pytest.fixture(scope="module", params=["d256", "d768"])
def mid(built, tmp_path_factory, request):
    # 256: one weight tile per row of out_proj / fc2 (8 sum blocks per row); 768: three (24 blocks; ViT-B's width)
    cfg = (synth.VitConfig(hidden=256, layers=4, heads=4, ff=512, patch=14, image=56, proj=64) if request.param == "d256" else
           synth.VitConfig(hidden=768, layers=3, heads=12, ff=1024, patch=14, image=56, proj=64))
    w, path = _model(tmp_path_factory.mktemp("w"), cfg)
    return cfg, w, path


def test_mid_tower_ln_fold_within_the_bf16_bound_and_chunking(mid, monkeypatch):
    cfg, w, path = mid
    px = synth.preprocess_rgb8(synth.images_u8(202, 40, cfg.image))
    ref = vit_numpy.vit_forward(w, cfg, px[:6], np.float64)
    rms = float(np.sqrt((ref ** 2).mean()))
    m = Model.from_file(path, 0, PRECISION_BF16)
    m.set_option("ln_fold", 0)
    base = m.forward(px)
    m.set_option("ln_fold", 1)
    out = m.forward(px)                       # 40 images: two half-chunk streams
    e_base = float(np.abs(base[:6] - ref).max() / rms)
    e_fold = float(np.abs(out[:6] - ref).max() / rms)
    print(f"max |err| / rms vs the fp64 oracle: LayerNorm tower {e_base:.3e}, ln_fold {e_fold:.3e}")
    assert e_fold < 3e-2, e_fold
    assert np.isfinite(out).all()
    # a row must not depend on its batch: one image, a one-stream batch, and the two-stream batch agree to the bit
    assert np.array_equal(m.forward(px[:1]).view(np.uint32), out[:1].view(np.uint32))
    assert np.array_equal(m.forward(px[:7]).view(np.uint32), out[:7].view(np.uint32))
    # full last layer == CLS rows only, tile orders, no split tail: bytes move differently, bits do not
    for key, val in (("full_last", 1), ("gemm_order", 0), ("split_tail", 0), ("parts", 1)):
        m.set_option(key, val)
        assert np.array_equal(m.forward(px).view(np.uint32), out.view(np.uint32)), key
    # and back: the option switches per forward, the LayerNorm tower's bits are untouched by the detour
    m.set_option("ln_fold", 0)
    for key, val in (("full_last", 0), ("gemm_order", 4), ("split_tail", 1), ("parts", 2)):
        m.set_option(key, val)
    assert np.array_equal(m.forward(px).view(np.uint32), base.view(np.uint32))
    m.close()


def test_ln_fold_counts_the_rows_its_rounding_cannot_serve_and_centering_keeps_them_away(mid, tmp_path, monkeypatch):
    """The LayerNorm-free loop rounds the UN-normalised residual row to bf16; a row whose mean lies r sigma off zero pays about
    r times the LayerNorm tower's rounding.  A constant on every channel of pre_layrnorm.bias plants such a common offset
    without changing the function (every later LayerNorm removes it).
    Weights as read (MI_CLIP_LN_CENTER=0): the library counts, on the device, the live rows with mean^2 > 16 var
    (mi_clip_ln_fold_stats) — none at 0 and 2 sigma, every row looked at at 64 sigma, where the fold's error has left the
    LayerNorm tower's far behind; "ln_fold" = 0, the switch the header names, restores it.
    Default load: the common mode of everything written to the stream is removed (the function does not change: every reader is
    a LayerNorm), so the same checkpoints give the error of the unplanted one and a count of zero."""
    cfg, w, path = mid
    n = 5
    S = (cfg.image // cfg.patch) ** 2 + 1
    px = synth.preprocess_rgb8(synth.images_u8(203, n, cfg.image))
    ref = vit_numpy.vit_forward(w, cfg, px, np.float64)
    rms = float(np.sqrt((ref ** 2).mean()))
    looked_at = n * S * (1 + 2 * (cfg.layers - 1))   # the embedding's statistics + two per LayerNorm-free layer
    name = [k for k in w if k.endswith("pre_layrnorm.bias")][0]
    errs = {}
    for center in (0, 1):
        monkeypatch.setenv("MI_CLIP_LN_CENTER", str(center))
        for c in (0.0, 2.0, 64.0):
            wc = dict(w)
            wc[name] = w[name] + np.float32(c)
            pc = str(tmp_path / f"offset_{int(c)}.safetensors")
            synth.save_safetensors(wc, pc, {"num_attention_heads": cfg.heads})
            m = Model.from_file(pc, 0, PRECISION_BF16)
            assert m.ln_fold_stats() == (0, 0)
            out = m.forward(px)
            bad, seen = m.ln_fold_stats(reset=True)
            assert seen == looked_at, (c, seen, looked_at)
            assert m.ln_fold_stats() == (0, 0)
            m.set_option("ln_fold", 0)
            base = m.forward(px)
            assert m.ln_fold_stats() == (0, 0)            # the LayerNorm tower neither needs nor feeds the counter
            m.close()
            errs[center, c] = (float(np.abs(out - ref).max() / rms), float(np.abs(base - ref).max() / rms), bad)
    print({k: tuple(round(v, 5) if isinstance(v, float) else v for v in e) for k, e in errs.items()})
    for key, (e_fold, e_ln, bad) in errs.items():
        assert e_ln < 3e-2, key                            # the LayerNorm tower does not care, centred weights or not
    assert errs[0, 0.0][2] == 0 and errs[0, 2.0][2] == 0 and errs[0, 2.0][0] < 3e-2
    assert errs[0, 64.0][2] == looked_at                   # weights as read: every row is flagged ...
    assert errs[0, 64.0][0] > 4.0 * errs[0, 64.0][1]       # ... and the fold's error shows why
    for c in (0.0, 2.0, 64.0):                             # the default load: no offset reaches the stream
        assert errs[1, c][2] == 0, c
        assert errs[1, c][0] < 3e-2 and errs[1, c][0] < 2.0 * max(errs[1, c][1], errs[1, 0.0][0]), (c, errs[1, c])


def test_a_nan_in_one_image_stays_a_nan_and_stays_in_that_image(mid):
    """The residual planes carry NaN / Inf patterns through their encode / decode (an integer add on the bit pattern), the row
    statistics of a poisoned row are NaN, and nothing of it may reach another image of the batch."""
    cfg, w, path = mid
    px = synth.preprocess_rgb8(synth.images_u8(203, 40, cfg.image))
    m = Model.from_file(path, 0, PRECISION_BF16)
    clean = m.forward(px)
    bad = px.copy()
    bad[3, 1, 20, 20] = np.nan
    bad[25, 0, 5, 7] = np.inf
    out = m.forward(bad)
    assert np.isnan(out[3]).all() and np.isnan(out[25]).all()
    keep = [i for i in range(40) if i not in (3, 25)]
    assert np.array_equal(out[keep].view(np.uint32), clean[keep].view(np.uint32))
    m.close()


def test_ln_fold_is_refused_where_it_cannot_run(built, tmp_path):
    cfg = synth.VitConfig.tiny()              # hidden 128: no 256-wide tiles
    w, path = _model(tmp_path, cfg)
    from image_search_amd._lib import MiError
    m = Model.from_file(path, 0, PRECISION_BF16)
    with pytest.raises(MiError):
        m.set_option("ln_fold", 1)
    m.set_option("ln_fold", 0)
    m.close()


def test_l14_ln_fold_within_the_bf16_bound(built, tmp_path_factory):
    cfg = synth.VitConfig.vit_l14()
    g = np.load(os.path.join(GOLDEN, "vit_l14.npz"))
    w = synth.vit_weights(cfg, int(g["seed"]))
    path = str(tmp_path_factory.mktemp("w") / "l14.safetensors")
    synth.save_safetensors(w, path, {"num_attention_heads": cfg.heads})
    u8 = synth.images_u8(int(g["image_seed"]), int(g["n_img"]), cfg.image)
    px = synth.preprocess_rgb8(u8)
    ref = g["embeds_f64"]
    rms = float(np.sqrt((ref ** 2).mean()))
    m = Model.from_file(path, 0, PRECISION_BF16)
    m.set_option("ln_fold", 0)
    base = m.forward(px)
    m.set_option("ln_fold", 1)
    out = m.forward(px)
    e_base, e_fold = float(np.abs(base - ref).max() / rms), float(np.abs(out - ref).max() / rms)
    print(f"ViT-L/14 max |err| / rms vs the fp64 golden: LayerNorm tower {e_base:.3e}, ln_fold {e_fold:.3e}")
    assert e_fold < 3e-2, e_fold
    big = synth.preprocess_rgb8(synth.images_u8(79, 40, cfg.image))
    o40 = m.forward(big)                      # two half-chunk streams, split tails on every GEMM
    assert np.array_equal(m.forward(big[:3]).view(np.uint32), o40[:3].view(np.uint32))
    m.set_option("full_last", 1)
    assert np.array_equal(m.forward(big).view(np.uint32), o40.view(np.uint32))
    m.close()
