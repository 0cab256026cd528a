"""numpy-facing wrappers of the op-level entry points (include/mi355clip_ops.h):
one device kernel of the vision tower per call, for per-op parity tests."""
from __future__ import annotations

import numpy as np

from ._lib import check, lib

EPI_STORE_F32, EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID = 0, 1, 2, 3
EPI_LNF, EPI_LNF_QGELU = 4, 5


def _f(a):
    return np.ascontiguousarray(a, np.float32)


def linear(x, w, bias=None, epilogue=EPI_BIAS, precision=0, out=None, device=0):
    x, w = _f(x), _f(w)
    m, k = x.shape
    n = w.shape[0]
    o = _f(out).copy() if out is not None else np.zeros((m, n), np.float32)
    b = _f(bias) if bias is not None else None
    check(lib().mi_op_linear(device, precision, epilogue, x.ctypes.data, w.ctypes.data,
                             b.ctypes.data if b is not None else None, o.ctypes.data, m, n, k))
    return o


def attention(qkv, heads, precision=0, device=0):
    qkv = _f(qkv)
    n, s, d3 = qkv.shape
    ctx = np.empty((n, s, d3 // 3), np.float32)
    check(lib().mi_op_attention(device, precision, qkv.ctypes.data, ctx.ctypes.data, n, s, d3 // 3, heads))
    return ctx


def layernorm(x, w, b, eps=1e-5, precision=0, device=0):
    x, w, b = _f(x), _f(w), _f(b)
    y = np.empty_like(x)
    check(lib().mi_op_layernorm(device, precision, x.ctypes.data, w.ctypes.data, b.ctypes.data, y.ctypes.data,
                                x.shape[0], x.shape[1], eps))
    return y


def linear_lnf(x, w, bias, c, stats, epilogue=EPI_LNF, device=0):
    """act(stats[:, :1] * (x @ w.T) + stats[:, 1:] * c + bias) on the persistent bf16 GEMM (the LayerNorm in
    front of the linear, finished in its epilogue: option "ln_fold")."""
    x, w, bias, c, stats = _f(x), _f(w), _f(bias), _f(c), _f(stats)
    m, k = x.shape
    n = w.shape[0]
    o = np.zeros((m, n), np.float32)
    check(lib().mi_op_linear_lnf(device, epilogue, x.ctypes.data, w.ctypes.data, bias.ctypes.data, c.ctypes.data,
                                 stats.ctypes.data, o.ctypes.data, m, n, k))
    return o


def linear_resid24(x, w, bias, xres, eps=1e-5, device=0):
    """xres += bf16(x @ w.T + bias) on the 24-bit residual planes in the GEMM's epilogue.
    Returns (new xres, hi plane widened, partial sums [m][n/32][2], stats [m][2])."""
    x, w, bias = _f(x), _f(w), _f(bias)
    r = _f(xres).copy()
    m, k = x.shape
    n = w.shape[0]
    hi = np.zeros((m, n), np.float32)
    part = np.zeros((m, n // 32, 2), np.float32)
    stats = np.zeros((m, 2), np.float32)
    check(lib().mi_op_linear_resid24(device, x.ctypes.data, w.ctypes.data, bias.ctypes.data, r.ctypes.data,
                                     hi.ctypes.data, part.ctypes.data, stats.ctypes.data, m, n, k, eps))
    return r, hi, part, stats
