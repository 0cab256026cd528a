"""Host-side mirror of the reference's ViT seam on top of the C ABI.

`clip_vit_large_patch14.Model.from_file(path, device)` and `.forward(tensor)` keep
the names and argument meaning of the generated Burn module the reference calls
(/root/reference/clip/src/lib.rs:2-7, call sites server/src/clip.rs:46-48, :118);
`image_prepare_resnet` is server/src/clip.rs:153-175: the `image` crate's CatmullRom
`resize_exact` (image 0.25.8) and the ImageNet normalisation, both on the device.
"""
from __future__ import annotations

import ctypes

import numpy as np

from ._lib import c_vp, check, lib

PRECISION_F32 = 0   # parity path (<= 1e-4 relative vs the fp32 CPU oracle)
PRECISION_BF16 = 1  # throughput path (bf16 MFMA operands, fp32 accumulate/residual)
PRECISION_BF16_SPLIT = 2  # bf16 with LayerNorm outputs split into hi + lo halves (outlier channels), ~1.5x the tower time

EXTENSIONS = ("jpg", "jpeg", "png", "gif", "bmp", "webp", "tiff")


def is_image_path(path: str) -> bool:
    """The extension allow-list of the ingest walk (server/src/clip.rs:60-66;
    the reference's own unit test `test_matches`, clip.rs:181-233)."""
    name = path.replace("\\", "/").rsplit("/", 1)[-1]
    dot = name.rfind(".")
    if dot <= 0:  # no '.', or a dot-file: Rust's Path::extension() is None
        return False
    return name[dot + 1:].lower() in EXTENSIONS


def resize_exact(rgb8_hwc: np.ndarray, nwidth: int, nheight: int, device: int = 0) -> np.ndarray:
    """`DynamicImage::resize_exact(nwidth, nheight, FilterType::CatmullRom)` for an RGB8 image
    [H,W,3] -> [nheight,nwidth,3] u8, on GPU `device` (clip.rs:154)."""
    a = np.ascontiguousarray(rgb8_hwc, np.uint8)
    if a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected [H,W,3] u8, got {a.shape}")
    out = np.empty((nheight, nwidth, 3), np.uint8)
    check(lib().mi_resize_catmullrom_rgb8(device, a.ctypes.data, a.shape[1], a.shape[0], nwidth, nheight, out.ctypes.data))
    return out


def image_prepare_resnet(rgb8_hwc: np.ndarray, device: int = 0) -> np.ndarray:
    """clip.rs:153-175.  One RGB8 image [H,W,3] of any size -> CHW f32 [3,224,224] (resize +
    normalisation on GPU `device`); a batch [n,224,224,3] already at model resolution keeps the
    host arithmetic of clip.rs:158-172."""
    a = np.ascontiguousarray(rgb8_hwc, np.uint8)
    if a.ndim == 3 and a.shape[:2] != (224, 224):
        out = np.empty((3, 224, 224), np.float32)
        check(lib().mi_image_prepare_resnet(device, a.ctypes.data, a.shape[1], a.shape[0], out.ctypes.data))
        return out
    single = a.ndim == 3
    a = a.reshape((-1,) + a.shape[-3:])
    n, h, w, _ = a.shape
    out = np.empty((n, 3, h, w), np.float32)
    check(lib().mi_preprocess_rgb8(a.ctypes.data, n, h, w, out.ctypes.data))
    return out[0] if single else out


def list_weights(path: str):
    """[(name, dtype, shape)] of a safetensors file or a Burn `.mpk` record, as the library names the tensors."""
    need = ctypes.c_size_t()
    check(lib().mi_weights_list(path.encode(), None, 0, ctypes.byref(need)))
    buf = ctypes.create_string_buffer(need.value)
    check(lib().mi_weights_list(path.encode(), buf, need.value, None))
    out = []
    for line in buf.value.decode().splitlines():
        name, dtype, shape = line.rsplit(" ", 2)
        out.append((name, dtype, tuple(int(x) for x in shape.strip("[]").split(",") if x)))
    return out


class Model:
    """clip::clip_vit_large_patch14::Model<B> on one MI355X."""

    def __init__(self, handle: c_vp):
        self._h = handle
        info = (ctypes.c_uint32 * 8)()
        check(lib().mi_clip_info(self._h, info))
        (self.image, self.patch, self.tokens, self.hidden, self.layers, self.heads, self.ff, self.proj) = list(info)

    @classmethod
    def from_file(cls, path: str, device: int = 0, precision: int = PRECISION_F32) -> "Model":
        h = c_vp()
        check(lib().mi_clip_load(path.encode(), device, precision, ctypes.byref(h)))
        return cls(h)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mi_clip_free(self._h)
            self._h = c_vp()

    __del__ = close

    def forward(self, tensor: np.ndarray) -> np.ndarray:
        """[n,3,H,W] f32 NCHW -> [n,proj] f32 (clip.rs:112-124); n == 0 allowed."""
        x = np.ascontiguousarray(tensor, np.float32)
        if x.ndim != 4 or x.shape[1:] != (3, self.image, self.image):
            raise ValueError(f"expected [n,3,{self.image},{self.image}], got {x.shape}")
        out = np.empty((x.shape[0], self.proj), np.float32)
        check(lib().mi_clip_embed(self._h, x.ctypes.data, x.shape[0], out.ctypes.data))
        return out

    def forward_rgb8(self, rgb8_hwc: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(rgb8_hwc, np.uint8)
        if a.ndim != 4 or a.shape[1:] != (self.image, self.image, 3):
            raise ValueError(f"expected [n,{self.image},{self.image},3] u8, got {a.shape}")
        out = np.empty((a.shape[0], self.proj), np.float32)
        check(lib().mi_clip_embed_rgb8(self._h, a.ctypes.data, a.shape[0], out.ctypes.data))
        return out

    def forward_images(self, images) -> np.ndarray:
        """One chunk of the scan loop (clip.rs:92-124): decoded RGB8 images [H_i,W_i,3] of any
        sizes -> [n,proj] f32; resize, normalisation and the tower all on the device."""
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        for im in imgs:
            if im.ndim != 3 or im.shape[2] != 3:
                raise ValueError(f"expected [H,W,3] u8, got {im.shape}")
        n = len(imgs)
        out = np.empty((n, self.proj), np.float32)
        ptrs = (c_vp * max(n, 1))(*[im.ctypes.data for im in imgs])
        ws = (ctypes.c_uint32 * max(n, 1))(*[im.shape[1] for im in imgs])
        hs = (ctypes.c_uint32 * max(n, 1))(*[im.shape[0] for im in imgs])
        check(lib().mi_clip_embed_images(self._h, ptrs, ws, hs, n, out.ctypes.data))
        return out

    def forward_device(self, d_nchw: int, n: int, d_out: int, stream: int = 0):
        check(lib().mi_clip_embed_device(self._h, d_nchw, n, d_out, stream))

    def set_option(self, key: str, value: int):
        """max_batch / parts / full_last / split_tail (include/mi355clip.h: mi_clip_set_option)."""
        check(lib().mi_clip_set_option(self._h, key.encode(), int(value)))

    def ln_fold_stats(self, reset: bool = False):
        """(rows whose mean lies more than 4 sigma off zero, rows looked at) of the LayerNorm-free layer loop since load /
        the last reset: non-zero first value on real weights = set option "ln_fold" to 0 (mi_clip_ln_fold_stats)."""
        out = (ctypes.c_uint64 * 2)()
        check(lib().mi_clip_ln_fold_stats(self._h, out, 1 if reset else 0))
        return int(out[0]), int(out[1])


class TextModel:
    """The CLIP text tower on one MI355X: what `clip(state, text)` obtains from embed_anything
    (server/src/clip.rs:19-23, :35-40).  `embed(input_ids)` takes tokenizer output
    [n, positions] (BOS .. EOS, padded) and returns [n, proj] f32."""

    def __init__(self, handle: c_vp):
        self._h = handle
        info = (ctypes.c_uint32 * 8)()
        check(lib().mi_clip_info(self._h, info))
        (_, _, self.positions, self.hidden, self.layers, self.heads, self.ff, self.proj) = list(info)

    @classmethod
    def from_file(cls, path: str, device: int = 0, precision: int = PRECISION_F32) -> "TextModel":
        h = c_vp()
        check(lib().mi_clip_load_text(path.encode(), device, precision, ctypes.byref(h)))
        return cls(h)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().mi_clip_free(self._h)
            self._h = c_vp()

    __del__ = close

    def set_option(self, key: str, value: int):
        """text_fast = 0 sends a single query through the batched kernels too (A/B; include/mi355clip.h)."""
        check(lib().mi_clip_set_option(self._h, key.encode(), int(value)))

    def embed(self, input_ids: np.ndarray) -> np.ndarray:
        ids = np.ascontiguousarray(input_ids, np.int32)
        if ids.ndim != 2 or ids.shape[1] != self.positions:
            raise ValueError(f"expected [n,{self.positions}] token ids, got {ids.shape}")
        out = np.empty((ids.shape[0], self.proj), np.float32)
        check(lib().mi_clip_embed_text(self._h, ids.ctypes.data, ids.shape[0], out.ctypes.data))
        return out


class clip_vit_large_patch14:  # noqa: N801 — the reference's module name
    Model = Model
