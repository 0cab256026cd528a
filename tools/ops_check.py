"""Dev harness (GPU): the op-level hooks (linear, LayerNorm, attention) against numpy on a few shapes. Not a test; tests/test_vit_gpu.py is."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import ops, synth

def bf(a):  # round to bf16 (RNE) and back
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint32) << 16
    return r.view(np.float32)

rng = np.random.default_rng(0)
# exact-integer GEMM check: asymmetric operands
for prec in (0, 1):
    M, N, K = 256, 256, 128
    x = rng.integers(-3, 4, (M, K)).astype(np.float32); w = rng.integers(-3, 4, (N, K)).astype(np.float32)
    b = rng.integers(-5, 6, N).astype(np.float32)
    ref = x @ w.T + b
    out = ops.linear(x, w, b, ops.EPI_BIAS, prec)
    print(f"linear int prec={prec}: max err {np.abs(out-ref).max()}  (ref range {np.abs(ref).max()})")
    if np.abs(out-ref).max() > 0:
        bad = np.argwhere(np.abs(out-ref) > 0); print("  first bad", bad[:8].tolist(), "count", len(bad))
    x = rng.standard_normal((300, 1024)).astype(np.float32); w = (rng.standard_normal((384, 1024)) * 0.03).astype(np.float32)
    b = rng.standard_normal(384).astype(np.float32)
    xr, wr = (bf(x), bf(w)) if prec else (x, w)
    ref = xr.astype(np.float64) @ wr.astype(np.float64).T + b
    out = ops.linear(x, w, b, ops.EPI_BIAS, prec)
    print(f"linear rand prec={prec}: max err {np.abs(out-ref).max():.3e}")
    res = rng.standard_normal((300, 384)).astype(np.float32)
    out = ops.linear(x, w, b, ops.EPI_BIAS_RESID, prec, out=res)
    print(f"linear resid prec={prec}: max err {np.abs(out-(ref+res)).max():.3e}")
    out = ops.linear(x, w, b, ops.EPI_BIAS_QGELU, prec)
    g = ref / (1 + np.exp(-1.702 * ref))
    print(f"linear qgelu prec={prec}: max err {np.abs(out-g).max():.3e}")
    # attention
    for S in (17, 257):
        n, H = 2, 2
        D = H * 64
        qkv = rng.standard_normal((n, S, 3 * D)).astype(np.float32)
        qr = bf(qkv) if prec else qkv
        q, k, v = [qr[..., i*D:(i+1)*D].reshape(n, S, H, 64).transpose(0, 2, 1, 3).astype(np.float64) for i in range(3)]
        s = q @ k.transpose(0, 1, 3, 2) * 0.125
        e = np.exp(s - s.max(-1, keepdims=True)); a = e / e.sum(-1, keepdims=True)
        ref = (a @ v).transpose(0, 2, 1, 3).reshape(n, S, D)
        out = ops.attention(qkv, H, prec)
        err = np.abs(out - ref)
        print(f"attention S={S} prec={prec}: max err {err.max():.3e}")
        if err.max() > 0.05:
            bad = np.argwhere(err > 0.05); print("  bad count", len(bad), "first", bad[:6].tolist())
    x = rng.standard_normal((10, 1024)).astype(np.float32) * 3 + 1
    w = rng.standard_normal(1024).astype(np.float32); b = rng.standard_normal(1024).astype(np.float32)
    xd = x.astype(np.float64); d = xd - xd.mean(-1, keepdims=True); ref = d / np.sqrt((d*d).mean(-1, keepdims=True) + 1e-5) * w + b
    print(f"layernorm prec={prec}: max err {np.abs(ops.layernorm(x, w, b, 1e-5, prec) - ref).max():.3e}")
