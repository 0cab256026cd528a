"""Writes a synthetic Burn `NamedMpkFileRecorder` record of a CLIP vision tower (what the reference's build step
produces as `vision_model.mpk`, clip/build.rs:75-83) from seeded weights, for the tests of the .mpk reader.

Layout (Burn 0.19, restated from its published format — NOT produced by Burn itself, no Rust toolchain here):
    {"metadata": {"float": "f32", "int": "i32", "format": "burn_core::record::file::NamedMpkFileRecorder<...>",
                  "version": "0.19.1", "settings": "..."},
     "item": {<field>: <module>, ...}}
    parameter = {"id": "<uuid>", "param": {"bytes": bin, "shape": [..], "dtype": "F32"}}
Field names imitate what burn-import generates from an ONNX graph (conv2d1, constant1, layernormalization1, linear1, ...)
in graph order; Linear weights are stored Burn's way, [d_in, d_out]; the embeddings keep their ONNX shapes
([1,1,D], [1,S,D]).  Three inventories (see burn_record): fused LayerNorm modules, the DECOMPOSED opset-16 form the
reference's build produces, and that form with un-coalesced MatMul + Add linears.  `legacy=True` writes tensors as {"value": [floats], "shape": [..]} (older Burn records).

    python tools/make_synthetic_mpk.py out.mpk [tiny|l14] [seed] [decomposed] [uncoalesced]
"""
import os
import sys

import msgpack
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth  # noqa: E402


def _tensor(a, legacy):
    a = np.asarray(a)  # rank 0 stays rank 0 (shape [])
    if a.dtype == np.int64:  # integer leaves (position ids, Reshape shapes): never a tower tensor
        if legacy:
            return {"value": [int(x) for x in a.reshape(-1)], "shape": list(a.shape)}
        return {"bytes": a.tobytes(), "shape": list(a.shape), "dtype": "I64"}
    a = a.astype(np.float32)
    if legacy:
        return {"value": [float(x) for x in a.reshape(-1)], "shape": list(a.shape)}
    return {"bytes": np.ascontiguousarray(a).tobytes(), "shape": list(a.shape), "dtype": "F32"}


def _param(a, n, legacy):
    return {"id": f"{n:032x}", "param": _tensor(a, legacy)}


def burn_record(weights: dict, cfg, legacy: bool = False, decomposed_ln: bool = False, coalesced: bool = True) -> dict:
    """`decomposed_ln=False`: LayerNorm as `layernormalizationN {gamma, beta, epsilon}` modules (an opset >= 17 graph).
    `decomposed_ln=True`: the inventory of the graph the reference really imports — opset 16
    (/root/reference/clip/scripts/upgrade_opset.py:9-28), LayerNorm as ReduceMean / Sub / Pow / ReduceMean / Add / Sqrt /
    Div / Mul / Add: gamma and beta are bare `constantN` [D] parameters in Mul-then-Add order, next to the scalar constants
    of the decomposition (the Pow exponent and epsilon as rank-0 F32), the attention scale ([1] F32), the QuickGELU factor
    (rank-0), the position ids (I64 [1,S]) and Reshape shapes (I64 [4]); the class embedding keeps its ONNX shape [D].
    `coalesced=False`: MatMul + Add pairs were NOT turned into Linear modules: every weight is a bare `constantN [in,out]`
    followed by its bias `constantN [out]`."""
    item, n = {}, [0]
    counters = {"linear": 0, "layernormalization": 0, "constant": 0, "embedding": 0}

    def field(kind):
        counters[kind] += 1
        return f"{kind}{counters[kind]}"

    def P(a):
        n[0] += 1
        return _param(a, n[0], legacy)

    def const(a):
        item[field("constant")] = P(a)

    v = "vision_model."
    D = cfg.hidden
    item["conv2d1"] = {"weight": P(weights[v + "embeddings.patch_embedding.weight"]), "bias": None,
                       "stride": [cfg.patch, cfg.patch], "kernel_size": [cfg.patch, cfg.patch], "dilation": [1, 1], "groups": 1,
                       "padding": {"Valid": None}}
    pos = weights[v + "embeddings.position_embedding.weight"]
    if decomposed_ln:
        const(weights[v + "embeddings.class_embedding"])                     # [D]
        const(np.arange(pos.shape[0], dtype=np.int64).reshape(1, -1))          # position_ids
        item[field("embedding")] = {"weight": P(pos)}                          # Gather over the table
    else:
        const(weights[v + "embeddings.class_embedding"].reshape(1, 1, D))
        const(pos.reshape(1, -1, D))

    def ln(prefix):
        if decomposed_ln:
            const(np.float32(2.0))                                             # Pow exponent, rank 0
            const(np.float32(1e-5))                                            # epsilon, rank 0
            const(weights[prefix + ".weight"])                                 # Mul
            const(weights[prefix + ".bias"])                                   # Add
        else:
            item[field("layernormalization")] = {"gamma": P(weights[prefix + ".weight"]), "beta": P(weights[prefix + ".bias"]), "epsilon": 1e-5}

    def linear(prefix, bias=True):
        if coalesced:
            item[field("linear")] = {"weight": P(weights[prefix + ".weight"].T), "bias": P(weights[prefix + ".bias"]) if bias else None}
        else:
            const(weights[prefix + ".weight"].T)                               # MatMul operand [in, out]
            if bias:
                const(weights[prefix + ".bias"])                               # Add

    ln(v + "pre_layrnorm")
    for layer in range(cfg.layers):
        p = f"{v}encoder.layers.{layer}."
        ln(p + "layer_norm1")
        for name in ("q_proj", "k_proj", "v_proj", "out_proj"):
            linear(p + "self_attn." + name)
            if decomposed_ln and name == "q_proj":
                const(np.array([0.125], np.float32))                           # the attention scale, [1]
            if decomposed_ln and name == "v_proj":
                const(np.array([-1, cfg.tokens, cfg.heads, D // cfg.heads], np.int64))   # a Reshape shape
        ln(p + "layer_norm2")
        linear(p + "mlp.fc1")
        if decomposed_ln:
            const(np.float32(1.702))                                           # QuickGELU
        linear(p + "mlp.fc2")
    ln(v + "post_layernorm")
    linear("visual_projection", bias=False)
    return {"metadata": {"float": "f32", "int": "i32",
                         "format": "burn_core::record::file::NamedMpkFileRecorder<burn_core::record::settings::FullPrecisionSettings>",
                         "version": "0.19.1", "settings": "FullPrecisionSettings"},
            "item": item}


def write_mpk(weights: dict, cfg, path: str, legacy: bool = False, mutate=None, decomposed_ln: bool = False, coalesced: bool = True):
    rec = burn_record(weights, cfg, legacy, decomposed_ln, coalesced)
    if mutate:
        mutate(rec)
    with open(path, "wb") as f:
        f.write(msgpack.packb(rec, use_bin_type=True))


if __name__ == "__main__":
    out = sys.argv[1]
    cfg = synth.VitConfig.vit_l14() if (len(sys.argv) > 2 and sys.argv[2] == "l14") else synth.VitConfig.tiny()
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    write_mpk(synth.vit_weights(cfg, seed), cfg, out, decomposed_ln="decomposed" in sys.argv, coalesced="uncoalesced" not in sys.argv)
    print(out, os.path.getsize(out), "bytes")
