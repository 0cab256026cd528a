"""ctypes binding of oracle/liboracle.so (test infrastructure; see oracle.c)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_F = ctypes.POINTER(ctypes.c_float)


def build_oracle() -> str:
    so = os.path.join(_DIR, "liboracle.so")
    src = os.path.join(_DIR, "oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def load_oracle() -> ctypes.CDLL:
    lib = ctypes.CDLL(build_oracle())
    lib.orc_gen_scale.restype = ctypes.c_float
    lib.orc_gen_scale.argtypes = [ctypes.c_double]
    lib.orc_gen_f32.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_float, _F]
    lib.orc_average_slices.argtypes = [ctypes.POINTER(_F), ctypes.c_size_t, ctypes.c_size_t, _F]
    lib.orc_refine.argtypes = [_F, ctypes.POINTER(_F), ctypes.c_size_t, ctypes.c_size_t, _F]
    lib.orc_preprocess_rgb8.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t, _F]
    lib.orc_cosine_dist.argtypes = [_F, _F, ctypes.c_uint64, ctypes.c_uint32, _F]
    lib.orc_cosine_dist_f64.argtypes = [_F, _F, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double)]
    lib.orc_knn.argtypes = [_F, _F, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64,
                            ctypes.POINTER(ctypes.c_uint64), _F]
    lib.orc_merge.argtypes = [ctypes.POINTER(ctypes.c_uint64), _F, ctypes.c_uint32, ctypes.c_uint32,
                              ctypes.POINTER(ctypes.c_uint64), _F]
    lib.orc_threads.restype = ctypes.c_int
    return lib


def _fp(a):
    return a.ctypes.data_as(_F)


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def orc_gen_f32(lib, seed, first, n, std=1.0):
    out = np.empty(n, np.float32)
    lib.orc_gen_f32(seed, first, n, lib.orc_gen_scale(std), _fp(out))
    return out


def _ptr_array(vecs):
    arr = (_F * len(vecs))()
    for i, v in enumerate(vecs):
        arr[i] = _fp(v)
    return arr


def orc_average_slices(lib, vecs):
    vecs = [_c(v) for v in vecs]
    n = len(vecs[0]) if vecs else 0
    out = np.empty(n, np.float32)
    rc = lib.orc_average_slices(_ptr_array(vecs), len(vecs), n, _fp(out))
    if rc != 0:
        raise ValueError("Input must not be empty")  # the reference's assert, search.rs:128
    return out


def orc_refine(lib, text, selected):
    text = _c(text)
    selected = [_c(v) for v in selected]
    out = np.empty_like(text)
    lib.orc_refine(_fp(text), _ptr_array(selected), len(selected), len(text), _fp(out))
    return out


def orc_preprocess(lib, hwc):
    hwc = _c(hwc, np.uint8)
    n = hwc.shape[0]
    out = np.empty((n, 3, 224, 224), np.float32)
    lib.orc_preprocess_rgb8(hwc.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n, _fp(out))
    return out


def orc_cosine_dist(lib, q, rows):
    q, rows = _c(q), _c(rows)
    out = np.empty(rows.shape[0], np.float32)
    lib.orc_cosine_dist(_fp(q), _fp(rows), rows.shape[0], rows.shape[1], _fp(out))
    return out


def orc_cosine_dist_f64(lib, q, rows):
    q, rows = _c(q), _c(rows)
    out = np.empty(rows.shape[0], np.float64)
    lib.orc_cosine_dist_f64(_fp(q), _fp(rows), rows.shape[0], rows.shape[1],
                            out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out


def orc_knn(lib, q, rows, k, base=0):
    q, rows = _c(q), _c(rows)
    idx = np.empty(k, np.uint64)
    dist = np.empty(k, np.float32)
    rc = lib.orc_knn(_fp(q), _fp(rows), rows.shape[0], rows.shape[1], k, base,
                     idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), _fp(dist))
    if rc != 0:
        raise ValueError(f"orc_knn rc={rc}")
    return idx, dist


def orc_merge(lib, idx_lists, dist_lists, k):
    idx_in = _c(idx_lists, np.uint64).reshape(-1)
    dist_in = _c(dist_lists, np.float32).reshape(-1)
    lists = idx_in.size // k
    idx = np.empty(k, np.uint64)
    dist = np.empty(k, np.float32)
    lib.orc_merge(idx_in.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), _fp(dist_in), lists, k,
                  idx.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), _fp(dist))
    return idx, dist


def orc_resize_catmullrom(lib, rgb8_hwc, nwidth, nheight):
    """image-0.25.8 resize_exact(.., CatmullRom) restated (oracle.c orc_resize_catmullrom_rgb8)."""
    a = np.ascontiguousarray(rgb8_hwc, np.uint8)
    out = np.zeros((nheight, nwidth, 3), np.uint8)
    lib.orc_resize_catmullrom_rgb8.restype = ctypes.c_int
    rc = lib.orc_resize_catmullrom_rgb8(ctypes.c_void_p(a.ctypes.data), ctypes.c_uint32(a.shape[1]), ctypes.c_uint32(a.shape[0]),
                                        ctypes.c_uint32(nwidth), ctypes.c_uint32(nheight), ctypes.c_void_p(out.ctypes.data))
    if rc != 0:
        raise RuntimeError("orc_resize_catmullrom_rgb8 failed")
    return out


def orc_image_prepare_resnet(lib, rgb8_hwc):
    a = np.ascontiguousarray(rgb8_hwc, np.uint8)
    out = np.zeros((3, 224, 224), np.float32)
    lib.orc_image_prepare_resnet.restype = ctypes.c_int
    rc = lib.orc_image_prepare_resnet(ctypes.c_void_p(a.ctypes.data), ctypes.c_uint32(a.shape[1]), ctypes.c_uint32(a.shape[0]),
                                      ctypes.c_void_p(out.ctypes.data))
    if rc != 0:
        raise RuntimeError("orc_image_prepare_resnet failed")
    return out
