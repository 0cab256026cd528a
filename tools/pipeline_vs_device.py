"""Dev harness (GPU): where does the forward's time go INSIDE the fused flow, against the same forward back to back on
device-resident input?  One process, one box, alternating:

  device      mi_clip_embed_device, 256 resident images, K forwards back to back, HIP events on the launch stream
  ingest      mi_pipeline_ingest from a pinned buffer (upload of chunk i+1 under forward i; rows written into the table), no query
  step        ingest + one top-k query per chunk over `rows` table rows (bench.py's step; two-stage search)
  step_small  the same with a 300 k-row table (a query that costs ~0.1 ms: what is left is the query's being there at all)

For the pipeline forms: wall ms per chunk and the per-forward span mi_pipeline_stats reports (events around each forward on the
ingest stream).   python tools/pipeline_vs_device.py [--rows 10000000] [--k 10] [--reps 20] [--rounds 3]
"""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_search_amd import synth
from image_search_amd.clip import Model, PRECISION_BF16
from image_search_amd.search import EmbeddingTable, PinnedBuffer, Pipeline


def main():
    def flag(name, dflt):
        return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt
    rows, k, reps, rounds, n = flag("--rows", 10_000_000), flag("--k", 10), flag("--reps", 20), flag("--rounds", 3), 256
    cfg = synth.VitConfig.vit_l14()
    path = os.path.join(tempfile.gettempdir(), f"pvd_{os.getpid()}.safetensors")
    synth.save_safetensors(synth.vit_weights(cfg, 0), path, {"num_attention_heads": cfg.heads})
    m = Model.from_file(path, 0, PRECISION_BF16)
    os.unlink(path)
    px = synth.preprocess_rgb8(synth.images_u8(100, n, cfg.image))
    pins = [PinnedBuffer(px.shape) for _ in range(2)]
    for p in pins:
        p.array[:] = px
    d_in = torch.from_numpy(px).cuda()
    d_out = torch.empty((n, 768), dtype=torch.float32, device="cuda")
    ts = torch.cuda.Stream()
    qs = synth.corpus_rows(1, 0, 64)
    total = (2 + rounds * (reps + 2)) * 2 * n + 1024
    big = EmbeddingTable(768, 0)
    big.reserve(rows + total)
    big.insert_synthetic(0, 0, rows)
    big.set_option("prefilter", 2)
    small = EmbeddingTable(768, 0)
    small.reserve(300_000 + total)
    small.insert_synthetic(0, 0, 300_000)
    small.set_option("prefilter", 2)
    pipes = {"big": Pipeline(m, big), "small": Pipeline(m, small)}

    def device():
        for _ in range(2):
            m.forward_device(d_in.data_ptr(), n, d_out.data_ptr(), ts.cuda_stream)
        ts.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(ts)
        for _ in range(reps):
            m.forward_device(d_in.data_ptr(), n, d_out.data_ptr(), ts.cuda_stream)
        b.record(ts)
        ts.synchronize()
        return {"ms": a.elapsed_time(b) / reps}

    def pipe_run(which, query):
        pipe = pipes[which]
        for i in range(2):
            pipe.ingest(pins[i & 1].array)
            if query:
                pipe.query(qs[i], k)
        pipe.sync()
        pipe.stats(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps):
            pipe.ingest(pins[i & 1].array)
            if query:
                pipe.query(qs[i % 64], k)
        pipe.sync()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        nf, msf, ns, mss = pipe.stats()
        return {"ms": wall, "forward_span_ms": msf / max(nf, 1), "scan_span_ms": (mss / ns) if ns else None}

    forms = {"device": device, "ingest": lambda: pipe_run("big", False), "step": lambda: pipe_run("big", True),
             "step_small": lambda: pipe_run("small", True)}
    res = {name: [] for name in forms}
    for _ in range(rounds):
        for name, f in forms.items():
            res[name].append({k2: (round(v, 3) if v is not None else None) for k2, v in f().items()})
    print(json.dumps({"rows": rows, "k": k, "reps": reps, "forms": res}, indent=1))


if __name__ == "__main__":
    main()
