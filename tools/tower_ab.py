"""Dev harness (GPU): A/B of the bf16 ViT-L/14 tower at b = 256 inside ONE process and one gpurun call.

  python tools/tower_ab.py ln_fold=0 ln_fold=1 [ln_fold=1,parts=1 ...] [--rounds 3] [--reps 10]

Every argument is one variant: comma-separated `option=value` pairs applied with mi_clip_set_option on top of
the defaults.  The variants are timed round-robin (box drift hits them all alike), device-resident input,
HIP events around `reps` forwards on the caller's stream; the embeddings of every variant are compared bit
for bit with the first one's.  Prints one JSON line.
"""
import json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if "--lib" in sys.argv:   # a probe build of the library (e.g. another cache policy compiled in): A/B across processes on one box
    from image_search_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from image_search_amd import synth
from image_search_amd.clip import Model, PRECISION_BF16

DEFAULTS = {"ln_fold": 1, "ln_nt": 0, "split_tail": 1, "parts": 2, "x24": 1, "gemm_order": 4, "qkv_pad": 128, "attn_order": 1, "qkv_layout": 0, "attn_nt": 0, "store_nt": 1}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib_note = sys.argv[sys.argv.index("--lib") + 1] if "--lib" in sys.argv else "image_search_amd/libmi355clip.so"
    def flag(name, dflt):
        return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt
    args = [a for a in args if "=" in a] or ["ln_fold=0", "ln_fold=1"]
    rounds, reps, n = flag("--rounds", 3), flag("--reps", 10), flag("--n", 256)
    cfg = synth.VitConfig.vit_l14()
    path = os.path.join(tempfile.gettempdir(), f"tower_ab_{os.getpid()}.safetensors")
    synth.save_safetensors(synth.vit_weights(cfg, 0), path, {"num_attention_heads": cfg.heads})
    m = Model.from_file(path, 0, PRECISION_BF16)
    os.unlink(path)
    px = synth.preprocess_rgb8(synth.images_u8(100, n, cfg.image))
    d_in = torch.from_numpy(px).cuda()
    d_out = torch.empty((n, 768), dtype=torch.float32, device="cuda")
    ts = torch.cuda.Stream()   # a real stream: the NULL stream would send the forward to the handle's own one
    torch.cuda.synchronize()
    torch.cuda.set_stream(ts)
    st = ts.cuda_stream

    def apply(v):
        opts = dict(DEFAULTS)
        opts.update({k: int(x) for k, x in (kv.split("=") for kv in v.split(","))})
        for k, x in opts.items():
            m.set_option(k, x)

    def run(k):
        for _ in range(k):
            m.forward_device(d_in.data_ptr(), n, d_out.data_ptr(), st)

    times = {v: [] for v in args}
    outs = {}
    for v in args:  # warm every variant (workspace, function attributes, streams)
        print("warm", v, file=sys.stderr, flush=True)
        apply(v); run(2); torch.cuda.synchronize()
        outs[v] = d_out.cpu().numpy().copy()
    for _ in range(rounds):
        for v in args:
            apply(v); run(1); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(reps); e1.record(); torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / reps)
    ref = outs[args[0]]
    res = {v: {"ms": [round(t, 3) for t in times[v]], "best": round(min(times[v]), 3),
               "bit_equal_to_first": bool((outs[v].view(np.uint32) == ref.view(np.uint32)).all()),
               "max_diff_to_first_over_rms": float(np.abs(outs[v] - ref).max() / np.sqrt((ref.astype(np.float64) ** 2).mean())),
               "finite": bool(np.isfinite(outs[v]).all())} for v in args}
    print(json.dumps({"n": n, "reps": reps, "lib": lib_note, "variants": res}))


if __name__ == "__main__":
    main()
