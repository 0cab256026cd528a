"""Soak (GPU): the two-stage exact search against the single pass over many queries and corpus shapes, both mirrors.
Prints one JSON object; any mismatch of ids or distance bits is a failure (exit code 1)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_search_amd.search import EmbeddingTable

N = int(os.environ.get("N", 1_000_000))
QUERIES = int(os.environ.get("QUERIES", 60))
gen = torch.Generator(device="cuda"); gen.manual_seed(2026)


def corpus(kind):
    if kind == "iid":
        return torch.randn((N, 768), device="cuda", generator=gen)
    if kind == "heavy tails":      # Student-t with 3 degrees of freedom: a few elements per row dominate the scale
        z = torch.randn((N, 768), device="cuda", generator=gen)
        w = torch.randn((N, 768, 3), device="cuda", generator=gen).square().sum(-1).div_(3).sqrt_()
        return z / w
    if kind == "outlier channels":  # CLIP-like: three fixed dimensions two orders of magnitude above the rest
        x = torch.randn((N, 768), device="cuda", generator=gen)
        x[:, [7, 133, 500]] *= 80.0
        return x
    if kind == "clusters":
        c = torch.randn((200, 768), device="cuda", generator=gen)
        return c[torch.randint(0, 200, (N,), device="cuda", generator=gen)] + 0.05 * torch.randn((N, 768), device="cuda", generator=gen)
    if kind == "unit norm, small offsets":  # rows = a common direction + small noise: all distances within 1e-2
        b = torch.randn((768,), device="cuda", generator=gen)
        return b[None, :] + 0.1 * torch.randn((N, 768), device="cuda", generator=gen)
    raise ValueError(kind)


out = {"rows": N, "queries_per_corpus": QUERIES, "results": []}
ok = True
for kind in ("iid", "heavy tails", "outlier channels", "clusters", "unit norm, small offsets"):
    x = corpus(kind)
    t = EmbeddingTable(768, 0)
    # torch's default stream is the NULL stream, and a NULL stream means "the handle's own stream" to insert_device — which is
    # not ordered behind torch's kernels: wait for the corpus first.  (Through round 3 this wait was missing: the table could
    # receive the PREVIOUS corpus's bytes from the reused allocation; ids and distance bits were compared on whatever the table
    # held, so the zero-mismatch results stand, but the candidate counts of the "clusters" / "unit norm" rows of older
    # profiles are those of stale iid-like data in some runs.)
    torch.cuda.synchronize()
    t.insert_device(x.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for mode in (1, 2):
        mism = 0; cands = []; fb = 0
        for j in range(QUERIES):
            k = (1, 10, 100, 1000)[j % 4]
            if j % 3 == 0:
                q = x[int(torch.randint(0, N, (1,), device="cuda", generator=gen))] + 0.02 * torch.randn((768,), device="cuda", generator=gen)
            else:
                q = torch.randn((768,), device="cuda", generator=gen)
            q = q.cpu().numpy()
            t.set_option("prefilter", 0)
            a = t.knn(q, k)
            t.set_option("prefilter", mode)
            b = t.knn(q, k)
            c, f = t.prefilter_stats()
            cands.append(c); fb += int(f)
            if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))):
                mism += 1
        ok &= mism == 0
        r = {"corpus": kind, "mirror": "bf16" if mode == 1 else "bytes", "mismatches": mism, "fallbacks": fb,
             "candidates_median": int(np.median(cands)), "candidates_max": int(max(cands))}
        print(r, file=sys.stderr, flush=True)
        out["results"].append(r)
    # the batched form (round 3): 8 and 4 queries share one pass over the byte mirror; near-a-stored-row and random queries
    # mixed in one call, every k; must equal the single pass query by query
    t.set_option("prefilter", 0)
    mism = 0; calls = 0
    for j in range(max(4, QUERIES // 4)):
        k = (1, 10, 100, 1000)[j % 4]
        nq = (8, 4, 13, 16)[j % 4]
        qs = torch.randn((nq, 768), device="cuda", generator=gen)
        near = torch.randint(0, N, (nq,), device="cuda", generator=gen)
        pick = torch.arange(nq, device="cuda") % 3 == 0
        qs[pick] = x[near[pick]] + 0.02 * qs[pick]
        qh = qs.cpu().numpy()
        t.set_option("prefilter", 0)
        a = t.knn(qh, k)
        t.set_option("prefilter", 2)
        b = t.knn(qh, k)
        calls += 1
        if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))):
            mism += 1
    ok &= mism == 0
    r = {"corpus": kind, "mirror": "bytes, batched calls of 8 / 4 / 13 / 16 queries", "calls": calls, "mismatches": mism}
    print(r, file=sys.stderr, flush=True)
    out["results"].append(r)
    t.close()
    del x
print(json.dumps(out, indent=1))
sys.exit(0 if ok else 1)
