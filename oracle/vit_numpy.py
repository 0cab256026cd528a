"""CPU restatement (numpy) of the CLIP vision tower the reference runs.

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never by image_search_amd/.

PARITY UNPINNED: the reference's model is `clip::clip_vit_large_patch14::Model`
(/root/reference/clip/src/lib.rs:2-7), Rust source generated at build time by
burn-import 0.19.1 from Xenova/clip-vit-large-patch14 onnx/vision_model.onnx
(/root/reference/clip/build.rs:9-11, :75-78) — neither the ONNX file, the
generated code nor burn is in this image, and the reference holds no golden
embedding.  This file restates the published architecture of that graph
(transformers models/clip/modeling_clip.py, the class the ONNX was exported
from) and is pinned against that library's CLIPVisionModelWithProjection on
seeded weights by oracle/make_golden.py (fixtures in tests/golden/).

Call sites restated: server/src/clip.rs:112-124 ([n,3,224,224] f32 NCHW in,
[n,768] f32 row-major out, no L2 normalisation).
"""
from __future__ import annotations

import numpy as np


def _layer_norm(x, w, b, eps):
    # opset-16 export keeps LayerNorm decomposed (clip/scripts/upgrade_opset.py:9,23):
    # ReduceMean -> Sub -> Pow -> ReduceMean -> Add(eps) -> Sqrt -> Div -> Mul -> Add
    mean = x.mean(axis=-1, keepdims=True, dtype=x.dtype)
    d = x - mean
    var = (d * d).mean(axis=-1, keepdims=True, dtype=x.dtype)
    return d / np.sqrt(var + x.dtype.type(eps)) * w + b


def _quick_gelu(x):
    # transformers/activations.py:117-123: x * sigmoid(1.702 x)
    return x / (x.dtype.type(1) + np.exp(x.dtype.type(-1.702) * x))


def _encoder(h, W, prefix, cfg, dtype, causal=False):
    """modeling_clip.py:353-383 x layers: pre-LN residual blocks; `causal` adds the text tower's
    mask (modeling_clip.py: _create_4d_causal_attention_mask — key j visible to query i iff j <= i)."""
    n, S, D = h.shape
    H, dh = cfg.heads, cfg.head_dim
    scale = dtype(dh ** -0.5)
    for i in range(cfg.layers):
        p = f"{prefix}encoder.layers.{i}."
        y = _layer_norm(h, W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], cfg.eps)
        # modeling_clip.py:280-335 — q/k/v/out projections with bias
        q = y @ W[p + "self_attn.q_proj.weight"].T + W[p + "self_attn.q_proj.bias"]
        k = y @ W[p + "self_attn.k_proj.weight"].T + W[p + "self_attn.k_proj.bias"]
        vv = y @ W[p + "self_attn.v_proj.weight"].T + W[p + "self_attn.v_proj.bias"]
        q = q.reshape(n, S, H, dh).transpose(0, 2, 1, 3)
        k = k.reshape(n, S, H, dh).transpose(0, 2, 1, 3)
        vv = vv.reshape(n, S, H, dh).transpose(0, 2, 1, 3)
        # modeling_clip.py:259-277 — softmax(q k^T * scale) v, softmax in fp32
        s = (q @ k.transpose(0, 1, 3, 2)) * scale
        if causal:
            s = np.where(np.tril(np.ones((S, S), bool)), s, dtype(-np.inf))
        s = s - s.max(axis=-1, keepdims=True)
        e = np.exp(s)
        a = e / e.sum(axis=-1, keepdims=True, dtype=dtype)
        ctx = (a @ vv).transpose(0, 2, 1, 3).reshape(n, S, D)
        h = h + (ctx @ W[p + "self_attn.out_proj.weight"].T + W[p + "self_attn.out_proj.bias"])
        y = _layer_norm(h, W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], cfg.eps)
        # modeling_clip.py:338-350 — fc1 -> QuickGELU -> fc2
        y = _quick_gelu(y @ W[p + "mlp.fc1.weight"].T + W[p + "mlp.fc1.bias"])
        h = h + (y @ W[p + "mlp.fc2.weight"].T + W[p + "mlp.fc2.bias"])
    return h


def vit_forward(weights: dict, cfg, pixels: np.ndarray, dtype=np.float32, return_hidden=False):
    """pixels [n,3,H,W] -> image_embeds [n,proj].  dtype float32 is the parity
    oracle; float64 gives the real-number answer used to size the noise floor."""
    W = {k: v.astype(dtype) for k, v in weights.items()}
    v = "vision_model."
    n = pixels.shape[0]
    D, P, G, H, dh = cfg.hidden, cfg.patch, cfg.grid, cfg.heads, cfg.head_dim
    x = pixels.astype(dtype)
    # modeling_clip.py:138-218 — conv (stride = kernel = patch, no bias) as a matmul
    # over [c,py,px]-ordered patches, CLS prepended, learned positions added.
    pt = x.reshape(n, 3, G, P, G, P).transpose(0, 2, 4, 1, 3, 5).reshape(n, G * G, 3 * P * P)
    pe = pt @ W[v + "embeddings.patch_embedding.weight"].reshape(D, 3 * P * P).T
    cls = np.broadcast_to(W[v + "embeddings.class_embedding"], (n, 1, D))
    h = np.concatenate([cls, pe], axis=1) + W[v + "embeddings.position_embedding.weight"]
    # modeling_clip.py:641-651 — pre_layrnorm -> encoder -> CLS -> post_layernorm
    h = _layer_norm(h, W[v + "pre_layrnorm.weight"], W[v + "pre_layrnorm.bias"], cfg.eps)
    h = _encoder(h, W, v, cfg, dtype)
    pooled = _layer_norm(h[:, 0, :], W[v + "post_layernorm.weight"], W[v + "post_layernorm.bias"], cfg.eps)
    # modeling_clip.py:944-950 — bias-free projection
    out = pooled @ W["visual_projection.weight"].T
    if return_hidden:
        return out, h
    return out


def text_forward(weights: dict, cfg, input_ids: np.ndarray, dtype=np.float32):
    """input_ids [n,positions] int -> text_embeds [n,proj]: the CLIP text tower the reference
    calls through embed_anything (server/src/clip.rs:19-23, :35-40), restated from
    modeling_clip.py (CLIPTextTransformer / CLIPTextModelWithProjection): token + position
    embeddings, causal pre-LN encoder, final LayerNorm, the row of the EOS token — located as
    `input_ids.argmax(-1)`, OpenAI's and candle's rule (EOS is the largest id) — and the
    bias-free text projection.  No L2 normalisation (server/src/clip.rs:22 takes the dense vector as is)."""
    W = {k: v.astype(dtype) for k, v in weights.items()}
    t = "text_model."
    ids = np.asarray(input_ids)
    n, S = ids.shape
    h = W[t + "embeddings.token_embedding.weight"][ids] + W[t + "embeddings.position_embedding.weight"][:S]
    h = _encoder(h, W, t, cfg, dtype, causal=True)
    h = _layer_norm(h, W[t + "final_layer_norm.weight"], W[t + "final_layer_norm.bias"], cfg.eps)
    pooled = h[np.arange(n), ids.argmax(axis=-1)]
    return pooled @ W["text_projection.weight"].T
