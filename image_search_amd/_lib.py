"""ctypes loader for libmi355clip.so (include/mi355clip.h).

There is no fallback: if the library is missing or a call fails, this raises.
Loading the library and resolving symbols needs no GPU; creating a handle does.
"""
from __future__ import annotations

import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libmi355clip.so")

c_f = ctypes.POINTER(ctypes.c_float)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_vp = ctypes.c_void_p

# every symbol include/mi355clip.h declares: (restype, argtypes)
SYMBOLS = {
    "mi_last_error": (ctypes.c_char_p, []),
    "mi_abi_version": (ctypes.c_int, []),
    "mi_device_count": (ctypes.c_int, []),
    "mi_clip_load": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "mi_clip_free": (None, [c_vp]),
    "mi_weights_list": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "mi_clip_set_option": (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_int]),
    "mi_clip_info": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "mi_clip_ln_fold_stats": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]),
    "mi_clip_embed": (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "mi_clip_embed_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t, c_vp, c_vp]),
    "mi_clip_embed_rgb8": (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "mi_clip_load_text": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "mi_clip_embed_text": (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t, c_vp]),
    "mi_preprocess_rgb8": (ctypes.c_int, [c_vp, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, c_vp]),
    "mi_resize_catmullrom_rgb8": (ctypes.c_int, [ctypes.c_int, c_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                                 ctypes.c_uint32, c_vp]),
    "mi_image_prepare_resnet": (ctypes.c_int, [ctypes.c_int, c_vp, ctypes.c_uint32, ctypes.c_uint32, c_vp]),
    "mi_clip_embed_images": (ctypes.c_int, [c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(ctypes.c_uint32),
                                            ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t, c_vp]),
    "mi_knn_create": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(c_vp)]),
    "mi_knn_free": (None, [c_vp]),
    "mi_knn_set_base": (ctypes.c_int, [c_vp, ctypes.c_uint64]),
    "mi_knn_set_option": (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_int]),
    "mi_knn_prefilter_stats": (ctypes.c_int, [c_vp, c_vp, c_vp]),
    "mi_knn_prefilter_state": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_uint32)]),
    "mi_knn_reserve": (ctypes.c_int, [c_vp, ctypes.c_uint64]),
    "mi_knn_size": (ctypes.c_int, [c_vp, c_u64p]),
    "mi_knn_append": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint64]),
    "mi_knn_append_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint64, c_vp]),
    "mi_knn_append_synthetic": (ctypes.c_int, [c_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]),
    "mi_knn_get_rows": (ctypes.c_int, [c_vp, ctypes.c_uint64, ctypes.c_uint64, c_vp]),
    "mi_knn_save": (ctypes.c_int, [c_vp, ctypes.c_char_p]),
    "mi_knn_load": (ctypes.c_int, [c_vp, ctypes.c_char_p]),
    "mi_knn_search": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp]),
    "mi_knn_search_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp, c_vp]),
    "mi_knn_search_batched_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp, c_vp]),
    "mi_knn_sharded_create": (ctypes.c_int, [ctypes.c_uint32, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_uint32,
                                             ctypes.POINTER(c_vp)]),
    "mi_knn_sharded_free": (None, [c_vp]),
    "mi_knn_sharded_info": (ctypes.c_int, [c_vp, c_u64p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32),
                                           ctypes.POINTER(ctypes.c_int)]),
    "mi_knn_sharded_stats": (ctypes.c_int, [c_vp, c_u64p]),
    "mi_knn_sharded_reserve": (ctypes.c_int, [c_vp, ctypes.c_uint64]),
    "mi_knn_sharded_set_option": (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_int]),
    "mi_knn_sharded_append": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint64, c_u64p]),
    "mi_knn_sharded_append_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint64, ctypes.c_int, c_vp, c_u64p]),
    "mi_knn_sharded_shard": (c_vp, [c_vp, ctypes.c_uint32]),
    "mi_knn_sharded_search_async": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp]),
    "mi_knn_sharded_sync": (ctypes.c_int, [c_vp]),
    "mi_knn_sharded_rebalance": (ctypes.c_int, [c_vp, c_vp]),
    "mi_knn_sharded_append_synthetic": (ctypes.c_int, [c_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64]),
    "mi_knn_sharded_get_rows": (ctypes.c_int, [c_vp, ctypes.c_uint64, ctypes.c_uint64, c_vp]),
    "mi_knn_sharded_search": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp]),
    "mi_knn_sharded_save": (ctypes.c_int, [c_vp, ctypes.c_char_p]),
    "mi_knn_sharded_load": (ctypes.c_int, [c_vp, ctypes.c_char_p]),
    "mi_knn_sharded_place": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32),
                                            c_u64p]),
    "mi_knn_sharded_id": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, c_u64p]),
    "mi_index_create": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(c_vp)]),
    "mi_index_free": (None, [c_vp]),
    "mi_index_table": (c_vp, [c_vp]),
    "mi_index_size": (ctypes.c_int, [c_vp, c_u64p]),
    "mi_index_media_dir": (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]),
    "mi_index_existing": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_size_t, c_vp]),
    "mi_index_insert": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_char_p), c_vp, ctypes.c_size_t, c_u64p]),
    "mi_index_adopt": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_size_t]),
    "mi_index_rows_of": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_size_t, c_vp, ctypes.c_size_t,
                                        ctypes.POINTER(ctypes.c_size_t)]),
    "mi_index_path": (ctypes.c_int, [c_vp, ctypes.c_uint64, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                     ctypes.POINTER(ctypes.c_size_t)]),
    "mi_index_search": (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(ctypes.c_char_p), ctypes.c_size_t, ctypes.c_uint32, c_vp, c_vp,
                                       ctypes.POINTER(ctypes.c_uint32)]),
    "mi_index_save": (ctypes.c_int, [c_vp, ctypes.c_char_p]),
    "mi_index_load": (ctypes.c_int, [c_vp, ctypes.c_char_p]),
    "mi_knn_merge": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp]),
    "mi_knn_merge_device": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, c_vp, c_vp, c_vp]),
    "mi_pipeline_query_device": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint32, c_vp, c_vp, c_vp]),
    "mi_pipeline_create": (ctypes.c_int, [c_vp, c_vp, ctypes.POINTER(c_vp)]),
    "mi_pipeline_create_sharded": (ctypes.c_int, [ctypes.POINTER(c_vp), ctypes.c_int, c_vp, ctypes.POINTER(c_vp)]),
    "mi_pipeline_free": (None, [c_vp]),
    "mi_pipeline_ingest": (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t, c_u64p]),
    "mi_pipeline_query": (ctypes.c_int, [c_vp, c_vp, ctypes.c_uint32, c_vp, c_vp]),
    "mi_pipeline_sync": (ctypes.c_int, [c_vp]),
    "mi_pipeline_drain": (ctypes.c_int, [c_vp, ctypes.c_uint32]),
    "mi_pipeline_stats": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.c_int]),
    "mi_host_alloc": (ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(c_vp)]),
    "mi_host_free": (None, [c_vp]),
    "mi_average_slices": (ctypes.c_int, [ctypes.POINTER(c_f), ctypes.c_size_t, ctypes.c_size_t, c_f]),
    "mi_refine": (ctypes.c_int, [c_f, ctypes.POINTER(c_f), ctypes.c_size_t, ctypes.c_size_t, c_f]),
    # include/mi355clip_ops.h (per-op test hooks)
    "mi_op_linear": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, c_vp,
                                    ctypes.c_size_t, ctypes.c_int, ctypes.c_int]),
    "mi_op_attention": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_vp, c_vp, ctypes.c_size_t, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_int]),
    "mi_op_clock_probe": (ctypes.c_int, [ctypes.c_int, c_vp, ctypes.POINTER(ctypes.c_float)]),
    "mi_op_linear_lnf": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        ctypes.c_size_t, ctypes.c_int, ctypes.c_int]),
    "mi_op_linear_resid24": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                            ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_float]),
    "mi_op_layernorm": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t,
                                       ctypes.c_int, ctypes.c_float]),
}

MI_OK = 0
ERR_NAMES = {-1: "MI_ERR_INVALID", -2: "MI_ERR_IO", -3: "MI_ERR_HIP", -4: "MI_ERR_NO_DEVICE",
             -5: "MI_ERR_UNSUPPORTED", -6: "MI_ERR_OOM"}


class MiError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


_lib = None


def lib() -> ctypes.CDLL:
    """Load the HIP library; raises (loudly) if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -m image_search_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64.so.7 and
        # libhsa-runtime64; if ours (from /opt/rocm) is mapped first, torch later binds to a mixed
        # set and reports "No HIP GPUs are available".  Loading torch first makes the dynamic linker
        # resolve our DT_NEEDED libamdhip64.so.7 to the copy already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != MI_OK:
        raise MiError(rc, lib().mi_last_error().decode(errors="replace"))
