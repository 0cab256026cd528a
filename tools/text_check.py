"""Latency of the text tower (one query = 77 token rows through 12 layers), fp32 and bf16, vs the numpy oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_search_amd import synth
from image_search_amd.clip import PRECISION_BF16, PRECISION_F32, TextModel
from oracle import vit_numpy

cfg = synth.TextConfig.clip_l14()
w = synth.vit_weights(cfg, 3)
path = "/tmp/text_l14.safetensors"
synth.save_safetensors(w, path, {"num_attention_heads": cfg.heads})
for pname, prec in (("fp32", PRECISION_F32), ("bf16", PRECISION_BF16)):
    m = TextModel.from_file(path, 0, prec)
    for n in (1, 8, 64):
        ids = synth.token_ids(cfg, 11, n)
        m.embed(ids)
        t0 = time.perf_counter()
        for _ in range(20):
            out = m.embed(ids)
        dt = (time.perf_counter() - t0) / 20
        print(f"{pname} n={n}: {dt*1e3:.3f} ms per call (host pointers in and out) = {n/dt:.0f} queries/s")
    if prec == PRECISION_BF16:
        m16 = m
        ids = synth.token_ids(cfg, 11, 1)
        for fast in (0, 1):
            m.set_option("text_fast", fast)
            m.embed(ids)
            t0 = time.perf_counter()
            for _ in range(50):
                m.embed(ids)
            print(f"bf16 n=1 text_fast={fast}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per call")
        for fuse in (0, 1, 0, 1):
            m.set_option("text_fuse", fuse)
            m.embed(ids); m.embed(ids); m.embed(ids)
            t0 = time.perf_counter()
            for _ in range(200):
                m.embed(ids)
            print(f"bf16 n=1 text_fuse={fuse}: {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per call ({'73' if fuse else '85'} launches in the graph)")
    else:
        m.close()
m = TextModel.from_file(path)
ids = synth.token_ids(cfg, 11, 1)
t0 = time.perf_counter()
ref = vit_numpy.text_forward(w, cfg, ids, np.float32)
print(f"numpy oracle, 1 query: {(time.perf_counter()-t0)*1e3:.1f} ms; max|diff| fp32 {np.abs(ref - m.embed(ids)).max():.2e}, bf16 {np.abs(ref - m16.embed(ids)).max():.2e} (rms {np.sqrt((ref**2).mean()):.2f})")
