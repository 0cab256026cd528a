"""Sequence view of the LAST tower forward in a rocprofv3 kernel_trace csv: per kernel its queue, start offset,
duration and the gap to the previous kernel of the same queue; then per-queue and whole-forward summaries.

  python tools/trace_seq.py <rocprof dir> [n_rows_to_print]
"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "mi::" in r["Kernel_Name"] and "knn" not in r["Kernel_Name"] and "gen_f32" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.split("(")[0]
    for k, v in (("gemm_bf16_pp_kernel<2", "fc1"), ("gemm_bf16_pp_kernel<1", "gemm"), ("attn32", "attn"), ("true>", "LN1"), ("false>", "LN2"),
                 ("embed_ln", "embedLN"), ("im2col", "im2col"), ("gemm_bf16_kernel", "patchgemm"), ("head", "head"), ("gather", "gather"), ("scatter", "scatter")):
        if k in n:
            return v
    return n[-20:]


ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows]
starts = [i for i, e in enumerate(ev) if e[2] == "im2col"]
# the last forward = from the second-to-last im2col that is > 5 ms before the end... simply: last two im2col launches
i0 = starts[-2] if len(starts) >= 2 and ev[starts[-1]][0] - ev[starts[-2]][0] < 5e6 else starts[-1]
fw = ev[i0:]
t0, t1 = fw[0][0], max(e[1] for e in fw)
print(f"forward: {len(fw)} kernels, wall {(t1 - t0) / 1e6:.3f} ms")
last_end = {}
nprint = int(sys.argv[2]) if len(sys.argv) > 2 else 60
gaps = collections.defaultdict(list)
for i, (s, e, n, q) in enumerate(fw):
    g = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    gaps[q].append(g)
    if 40 <= i < 40 + nprint:
        print(f"  q{q:>3} +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {g:7.1f}  {n}")
    last_end[q] = max(e, last_end.get(q, 0))
for q, g in gaps.items():
    g2 = sorted(g)
    print(f"queue {q}: {len(g)} kernels, sum positive gaps {sum(x for x in g if x > 0) / 1e3:.3f} ms, median gap {g2[len(g2) // 2]:.1f} us, p90 {g2[int(len(g2) * .9)]:.1f} us")
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in fw:
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:10s} x{c:4d}  total {t / 1e3:8.3f} ms  avg {t / c:8.1f} us")
# union of busy time of R kernels (gemm, fc1, attn)
R = sorted((s, e) for s, e, n, q in fw if n in ("gemm", "fc1", "attn", "patchgemm"))
busy = 0; cs, ce = R[0]
for s, e in R[1:]:
    if s > ce:
        busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"R kernels (gemm, attention): union busy {busy / 1e6:.3f} ms of {(t1 - t0) / 1e6:.3f} ms wall; sum of durations {sum(e - s for s, e in R) / 1e6:.3f} ms")
