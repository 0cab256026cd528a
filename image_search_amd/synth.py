"""Seeded synthetic inputs for the hot path.

The reference ships no weights, images or corpus (its model is a build-time
Hugging Face download, /root/reference/clip/build.rs:9-11), so every run here
uses a counter-based generator that is integer-only up to one fp32 multiply:
the same bits come out of this numpy code, the C oracle (oracle/oracle.c
orc_gen_f32) and the HIP kernel (csrc/knn.hip gen_rows_kernel) on any machine.

value(seed, i) = (f0+f1+f2+f3 - 131070) * scale, f* = the four 16-bit fields
of h = mix64(i + mix64(seed + 0x9E3779B97F4A7C15)); an Irwin-Hall(4) bell with
standard deviation `std` when scale = std / sqrt(1431655765).
"""
from __future__ import annotations

import math

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_SUM_VAR = 1431655765.0  # 4 * (65536**2 - 1) / 12


def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def gen_scale(std: float) -> np.float32:
    return np.float32(std / math.sqrt(_SUM_VAR))


def gen_f32(seed: int, first: int, n: int, std: float = 1.0) -> np.ndarray:
    """n values starting at counter `first` of stream `seed`, fp32."""
    with np.errstate(over="ignore"):
        key = _mix64(np.array([seed], dtype=np.uint64) + _GOLDEN)[0]
        out = np.empty(n, dtype=np.float32)
        scale = gen_scale(std)
        step = 1 << 22
        for s in range(0, n, step):
            m = min(step, n - s)
            i = np.arange(first + s, first + s + m, dtype=np.uint64)
            h = _mix64(i + key)
            f = ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF))
                 + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + (h >> np.uint64(48)))
            out[s:s + m] = (f.astype(np.int64) - 131070).astype(np.float32) * scale
    return out


def corpus_rows(seed: int, first_row: int, n_rows: int, dim: int = 768) -> np.ndarray:
    """Rows [first_row, first_row+n_rows) of the synthetic [N,dim] embedding table
    (counter = row*dim + col, std 1; raw, not normalised — the reference stores
    raw embeddings, server/src/clip.rs:120-137)."""
    return gen_f32(seed, first_row * dim, n_rows * dim, 1.0).reshape(n_rows, dim)


def images_u8(seed: int, n: int, size: int = 224) -> np.ndarray:
    """n uniform-random RGB8 images, HWC interleaved (what `to_rgb8().as_raw()` is,
    server/src/clip.rs:155-156)."""
    with np.errstate(over="ignore"):
        key = _mix64(np.array([seed], dtype=np.uint64) + _GOLDEN)[0]
        i = np.arange(n * size * size * 3, dtype=np.uint64)
        return (_mix64(i + key) >> np.uint64(56)).astype(np.uint8).reshape(n, size, size, 3)


def photo_u8(seed: int, height: int, width: int) -> np.ndarray:
    """A seeded photo-like RGB8 image [height,width,3]: smooth structure at several scales plus
    per-pixel noise from the counter generator, so that a resampler sees both gradients and
    high-frequency content (stand-in for a decoded JPEG, server/src/clip.rs:92-104)."""
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    base = np.stack([128 + 70 * np.sin(xx / 17.0 + seed) + 40 * np.cos(yy / 23.0),
                     255.0 * xx / max(width - 1, 1),
                     255.0 * yy / max(height - 1, 1) * (0.5 + 0.5 * np.sin(xx / 5.0))], -1)
    with np.errstate(over="ignore"):
        key = _mix64(np.array([seed ^ 0x5EED], dtype=np.uint64) + _GOLDEN)[0]
        i = np.arange(height * width * 3, dtype=np.uint64)
        noise = ((_mix64(i + key) >> np.uint64(58)).astype(np.float64) - 32.0).reshape(height, width, 3)
    return np.clip(base + noise, 0, 255).astype(np.uint8)


def scenes_u8(seed: int, n: int, size: int = 224) -> np.ndarray:
    """n seeded images that DIFFER from one another the way photos do (uniform noise images all look alike
    to a ViT: every patch has the same statistics): per image a random palette, 2-4 oriented sinusoidal
    gratings of random frequency and phase, a soft blob and mild pixel noise.  [n,size,size,3] u8."""
    out = np.empty((n, size, size, 3), np.uint8)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / np.float32(size)
    for i in range(n):
        p = gen_f32(seed * 7919 + 13, i * 64, 64, 1.0).astype(np.float32)      # this image's parameters
        img = np.zeros((size, size, 3), np.float32)
        base = 128 + 60 * p[0:3]
        for g in range(2 + int(abs(p[3]) * 1.5) % 3):
            fx, fy, ph = 2 + 14 * abs(p[4 + 4 * g]), 2 + 14 * abs(p[5 + 4 * g]), 6.28 * p[6 + 4 * g]
            img += (np.sin(6.2832 * (fx * xx + fy * yy) + ph)[..., None] * (40 * p[20 + 3 * g:23 + 3 * g])).astype(np.float32)
        cx, cy, rad = 0.5 + 0.3 * p[40], 0.5 + 0.3 * p[41], 0.08 + 0.1 * abs(p[42])
        blob = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * rad * rad))
        img += blob[..., None] * (90 * p[43:46])
        with np.errstate(over="ignore"):
            key = _mix64(np.array([seed * 1000003 + i], dtype=np.uint64) + _GOLDEN)[0]
            k = np.arange(size * size * 3, dtype=np.uint64)
            noise = ((_mix64(k + key) >> np.uint64(59)).astype(np.float32) - 16.0).reshape(size, size, 3)
        out[i] = np.clip(base + img + noise, 0, 255).astype(np.uint8)
    return out


def plant_outlier_channels(weights: dict, dims=(7, 301, 666, 900), factor: float = 50.0, compensate: bool = False) -> dict:
    """Trained CLIP ViT-L towers carry a few channels two orders of magnitude above the rest; random-init
    weights do not.  Scales the LayerNorm gains AND biases of `dims` in every layer_norm1/2 by `factor`, so that
    the GEMM inputs see such channels.  compensate=True divides the matching input columns of q/k/v and fc1 by
    the same factor: the function is unchanged (up to rounding), only the intermediate magnitudes differ —
    separates "bf16 cannot represent outliers" (it can: a float keeps its relative precision) from "the planted
    channels change the function" (they make the softmax sharply peaked).  Returns a new dict."""
    out = dict(weights)
    d = list(dims)
    f = np.float32(factor)
    for name, w in weights.items():
        if name.endswith("layer_norm1.weight") or name.endswith("layer_norm2.weight") or \
                name.endswith("layer_norm1.bias") or name.endswith("layer_norm2.bias"):
            w = w.copy()
            w[d] *= f
            out[name] = w
        elif compensate and (name.endswith("q_proj.weight") or name.endswith("k_proj.weight") or
                             name.endswith("v_proj.weight") or name.endswith("fc1.weight")):
            w = w.copy()
            w[:, d] /= f
            out[name] = w
    return out


def preprocess_rgb8(hwc: np.ndarray) -> np.ndarray:
    """image_prepare_resnet's arithmetic, server/src/clip.rs:158-172: p/255, minus
    ImageNet mean, divided by ImageNet std, planar CHW f32 (fp32 ops throughout)."""
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)
    x = hwc.astype(np.float32) / np.float32(255.0)
    x = (x - mean) / std
    return np.ascontiguousarray(np.moveaxis(x, -1, -3))


class VitConfig:
    """Shape of a CLIP vision tower + projection.  `vit_l14()` is the reference's
    model (Xenova/clip-vit-large-patch14, clip/build.rs:10-11); `tiny()` is a
    millisecond-sized config of the same graph for unit tests."""

    def __init__(self, hidden=1024, layers=24, heads=16, ff=4096, patch=14, image=224, proj=768, eps=1e-5):
        self.hidden, self.layers, self.heads, self.ff = hidden, layers, heads, ff
        self.patch, self.image, self.proj, self.eps = patch, image, proj, eps
        assert hidden % heads == 0 and image % patch == 0
        self.head_dim = hidden // heads
        self.grid = image // patch
        self.tokens = self.grid * self.grid + 1

    @staticmethod
    def vit_l14():
        return VitConfig()

    @staticmethod
    def tiny():
        return VitConfig(hidden=128, layers=2, heads=2, ff=512, patch=14, image=56, proj=64)

    def tensor_specs(self):
        """(HF state-dict name, shape, std, offset) in file order; std mirrors HF's
        CLIP init scales (transformers modeling_clip.py:412-443) so activations stay
        well-conditioned; LayerNorm affine and biases are non-trivial on purpose so
        those code paths are exercised."""
        D, L, FF, P, E = self.hidden, self.layers, self.ff, self.patch, self.proj
        f = 1.0
        in_std = D ** -0.5 * (2 * L) ** -0.5 * f
        out_std = D ** -0.5 * f
        fc_std = (2 * D) ** -0.5 * f
        v = "vision_model."
        specs = [
            (v + "embeddings.class_embedding", (D,), D ** -0.5, 0.0),
            (v + "embeddings.patch_embedding.weight", (D, 3, P, P), 0.02, 0.0),
            (v + "embeddings.position_embedding.weight", (self.tokens, D), 0.02, 0.0),
            (v + "pre_layrnorm.weight", (D,), 0.05, 1.0),
            (v + "pre_layrnorm.bias", (D,), 0.02, 0.0),
        ]
        for i in range(L):
            p = f"{v}encoder.layers.{i}."
            specs += [
                (p + "layer_norm1.weight", (D,), 0.05, 1.0), (p + "layer_norm1.bias", (D,), 0.02, 0.0),
                (p + "self_attn.q_proj.weight", (D, D), in_std, 0.0), (p + "self_attn.q_proj.bias", (D,), 0.02, 0.0),
                (p + "self_attn.k_proj.weight", (D, D), in_std, 0.0), (p + "self_attn.k_proj.bias", (D,), 0.02, 0.0),
                (p + "self_attn.v_proj.weight", (D, D), in_std, 0.0), (p + "self_attn.v_proj.bias", (D,), 0.02, 0.0),
                (p + "self_attn.out_proj.weight", (D, D), out_std, 0.0), (p + "self_attn.out_proj.bias", (D,), 0.02, 0.0),
                (p + "layer_norm2.weight", (D,), 0.05, 1.0), (p + "layer_norm2.bias", (D,), 0.02, 0.0),
                (p + "mlp.fc1.weight", (FF, D), fc_std, 0.0), (p + "mlp.fc1.bias", (FF,), 0.02, 0.0),
                (p + "mlp.fc2.weight", (D, FF), in_std, 0.0), (p + "mlp.fc2.bias", (D,), 0.02, 0.0),
            ]
        specs += [
            (v + "post_layernorm.weight", (D,), 0.05, 1.0),
            (v + "post_layernorm.bias", (D,), 0.02, 0.0),
            ("visual_projection.weight", (E, D), D ** -0.5, 0.0),
        ]
        return specs


class TextConfig:
    """Shape of the CLIP text tower + projection the reference reaches through embed_anything
    (`Embedder::from_pretrained_hf("Clip", "openai/clip-vit-large-patch14")`, server/src/clip.rs:35-40;
    used by `clip()`, :19-23).  `clip_l14()` is that model's text side; `tiny()` a unit-test size."""

    def __init__(self, hidden=768, layers=12, heads=12, ff=3072, vocab=49408, positions=77, proj=768, eps=1e-5):
        self.hidden, self.layers, self.heads, self.ff = hidden, layers, heads, ff
        self.vocab, self.positions, self.proj, self.eps = vocab, positions, proj, eps
        assert hidden % heads == 0
        self.head_dim = hidden // heads
        self.tokens = positions

    @staticmethod
    def clip_l14():
        return TextConfig()

    @staticmethod
    def tiny():
        return TextConfig(hidden=128, layers=2, heads=2, ff=512, vocab=1000, positions=20, proj=64)

    def tensor_specs(self):
        D, L, FF, E = self.hidden, self.layers, self.ff, self.proj
        in_std = D ** -0.5 * (2 * L) ** -0.5
        out_std = D ** -0.5
        fc_std = (2 * D) ** -0.5
        t = "text_model."
        specs = [
            (t + "embeddings.token_embedding.weight", (self.vocab, D), 0.02, 0.0),
            (t + "embeddings.position_embedding.weight", (self.positions, D), 0.01, 0.0),
        ]
        for i in range(L):
            p = f"{t}encoder.layers.{i}."
            specs += [
                (p + "layer_norm1.weight", (D,), 0.05, 1.0), (p + "layer_norm1.bias", (D,), 0.02, 0.0),
                (p + "self_attn.q_proj.weight", (D, D), in_std, 0.0), (p + "self_attn.q_proj.bias", (D,), 0.02, 0.0),
                (p + "self_attn.k_proj.weight", (D, D), in_std, 0.0), (p + "self_attn.k_proj.bias", (D,), 0.02, 0.0),
                (p + "self_attn.v_proj.weight", (D, D), in_std, 0.0), (p + "self_attn.v_proj.bias", (D,), 0.02, 0.0),
                (p + "self_attn.out_proj.weight", (D, D), out_std, 0.0), (p + "self_attn.out_proj.bias", (D,), 0.02, 0.0),
                (p + "layer_norm2.weight", (D,), 0.05, 1.0), (p + "layer_norm2.bias", (D,), 0.02, 0.0),
                (p + "mlp.fc1.weight", (FF, D), fc_std, 0.0), (p + "mlp.fc1.bias", (FF,), 0.02, 0.0),
                (p + "mlp.fc2.weight", (D, FF), in_std, 0.0), (p + "mlp.fc2.bias", (D,), 0.02, 0.0),
            ]
        specs += [
            (t + "final_layer_norm.weight", (D,), 0.05, 1.0),
            (t + "final_layer_norm.bias", (D,), 0.02, 0.0),
            ("text_projection.weight", (E, D), D ** -0.5, 0.0),
        ]
        return specs


def token_ids(cfg: TextConfig, seed: int, n: int) -> np.ndarray:
    """n seeded token sequences [n, positions] int32 shaped like a CLIP tokenizer's output:
    BOS (vocab-2), a random number of word tokens, EOS (vocab-1, the largest id, so that
    `argmax` finds it — modeling_clip.py's eos_token_id == 2 branch), EOS padding."""
    with np.errstate(over="ignore"):
        key = _mix64(np.array([seed ^ 0x7E87], dtype=np.uint64) + _GOLDEN)[0]
        r = _mix64(np.arange(n * (cfg.positions + 1), dtype=np.uint64) + key).reshape(n, cfg.positions + 1)
    ids = np.full((n, cfg.positions), cfg.vocab - 1, np.int32)
    ids[:, 0] = cfg.vocab - 2
    for i in range(n):
        words = 1 + int(r[i, 0] % np.uint64(cfg.positions - 2))
        ids[i, 1:1 + words] = (r[i, 1:1 + words] % np.uint64(cfg.vocab - 2)).astype(np.int32)
    return ids


def vit_weights(cfg, seed: int = 0) -> dict:
    """Seed-reproducible weights: tensor t of the spec list is stream seed*4096+t."""
    out = {}
    for t, (name, shape, std, offset) in enumerate(cfg.tensor_specs()):
        n = int(np.prod(shape))
        w = gen_f32(seed * 4096 + t + 1, 0, n, std)
        if offset:
            w = w + np.float32(offset)
        out[name] = w.reshape(shape)
    return out


def save_safetensors(weights: dict, path: str, metadata: dict | None = None) -> None:
    """Write a Hugging Face `safetensors` file (8-byte LE header length, JSON header,
    raw little-endian data) without needing the safetensors package."""
    import json
    header, off = {}, 0
    if metadata:
        header["__metadata__"] = {k: str(v) for k, v in metadata.items()}
    for name, w in weights.items():
        nbytes = int(w.size) * 4
        header[name] = {"dtype": "F32", "shape": list(w.shape), "data_offsets": [off, off + nbytes]}
        off += nbytes
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(len(hj).to_bytes(8, "little"))
        f.write(hj)
        for w in weights.values():
            f.write(np.ascontiguousarray(w, dtype="<f4").tobytes())
