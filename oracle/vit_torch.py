"""Library-grade CPU baseline (torch, all host cores) of the two kernels of the path.

TEST INFRASTRUCTURE ONLY — used by bench.py's cpu_baseline leg ("kind": "library") and by
tests/test_oracle.py, never by image_search_amd/.  Where oracle/vit_numpy.py and oracle/oracle.c fix every
summation order so that results can be compared bit for bit, this file lets the CPU run as fast as its
libraries allow (oneDNN / MKL GEMMs, fused softmax attention, `topk`): the CPU number someone would
actually deploy, reported beside the port's.  Same graph as vit_numpy.vit_forward (the reference's
`Model::forward`, /root/reference/server/src/clip.rs:118; transformers modeling_clip.py:138-383, :641-651,
:944-950) and the same cosine distance as orc_knn (/root/reference/server/src/search.rs:70-77), in
whatever order the libraries sum.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def load_weights(weights: dict) -> dict:
    return {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)) for k, v in weights.items()}


@torch.no_grad()
def vit_forward(W: dict, cfg, pixels) -> torch.Tensor:
    """pixels [n,3,H,W] f32 -> [n,proj] f32 (no L2 normalisation)."""
    v = "vision_model."
    x = torch.as_tensor(pixels, dtype=torch.float32)
    n = x.shape[0]
    D, H = cfg.hidden, cfg.heads
    pe = F.conv2d(x, W[v + "embeddings.patch_embedding.weight"], stride=cfg.patch).flatten(2).transpose(1, 2)
    cls = W[v + "embeddings.class_embedding"].expand(n, 1, D)
    h = torch.cat([cls, pe], dim=1) + W[v + "embeddings.position_embedding.weight"]
    h = F.layer_norm(h, (D,), W[v + "pre_layrnorm.weight"], W[v + "pre_layrnorm.bias"], cfg.eps)
    S = h.shape[1]
    for i in range(cfg.layers):
        p = f"{v}encoder.layers.{i}."
        y = F.layer_norm(h, (D,), W[p + "layer_norm1.weight"], W[p + "layer_norm1.bias"], cfg.eps)
        q = F.linear(y, W[p + "self_attn.q_proj.weight"], W[p + "self_attn.q_proj.bias"]).view(n, S, H, -1).transpose(1, 2)
        k = F.linear(y, W[p + "self_attn.k_proj.weight"], W[p + "self_attn.k_proj.bias"]).view(n, S, H, -1).transpose(1, 2)
        vv = F.linear(y, W[p + "self_attn.v_proj.weight"], W[p + "self_attn.v_proj.bias"]).view(n, S, H, -1).transpose(1, 2)
        ctx = F.scaled_dot_product_attention(q, k, vv).transpose(1, 2).reshape(n, S, D)
        h = h + F.linear(ctx, W[p + "self_attn.out_proj.weight"], W[p + "self_attn.out_proj.bias"])
        y = F.layer_norm(h, (D,), W[p + "layer_norm2.weight"], W[p + "layer_norm2.bias"], cfg.eps)
        y = F.linear(y, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"])
        y = y * torch.sigmoid(1.702 * y)
        h = h + F.linear(y, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"])
    pooled = F.layer_norm(h[:, 0, :], (D,), W[v + "post_layernorm.weight"], W[v + "post_layernorm.bias"], cfg.eps)
    return F.linear(pooled, W["visual_projection.weight"])


@torch.no_grad()
def knn(rows: torch.Tensor, norms: torch.Tensor, q: torch.Tensor, k: int):
    """cosine top-k of one query over [N,dim] rows; `norms` = the rows' Euclidean norms (kept beside the table,
    as an index would).  Returns (ids, distances) ascending by distance."""
    d = 1.0 - torch.mv(rows, q) / (norms * q.norm())
    dist, idx = torch.topk(d, k, largest=False, sorted=True)
    return idx, dist
