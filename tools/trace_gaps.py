"""Timeline analysis of a rocprofv3 kernel_trace csv: busy union vs wall, per-queue gaps, for the ViT forward kernels."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "mi::" in r["Kernel_Name"] and "knn" not in r["Kernel_Name"] and "gen_f32" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into forwards by im2col occurrences (2 per forward with parts=2)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:], r.get("Queue_Id", "")) for r in rows]
starts = [i for i, e in enumerate(ev) if "im2col" in e[2]]
print("kernels", len(ev), "im2col launches", len(starts))
# take the last forward: from the second-to-last pair of im2col
parts = 2 if len(starts) >= 2 and (int(sys.argv[2]) if len(sys.argv) > 2 else 2) == 2 else 1
i0 = starts[-parts]
fw = ev[i0:]
t0, t1 = fw[0][0], max(e[1] for e in fw)
busy = 0; cur_s, cur_e = None, None
for s, e, _, _ in sorted(fw):
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"forward wall {(t1-t0)/1e6:.3f} ms, union busy {busy/1e6:.3f} ms, idle {(t1-t0-busy)/1e6:.3f} ms, sum of kernel durations {sum(e[1]-e[0] for e in fw)/1e6:.3f} ms")
byq = collections.defaultdict(list)
for e in fw: byq[e[3]].append(e)
for q, L in byq.items():
    L.sort()
    gaps = [L[i+1][0]-L[i][1] for i in range(len(L)-1)]
    print(f"queue {q}: {len(L)} kernels, busy {sum(e[1]-e[0] for e in L)/1e6:.3f} ms, sum gaps {sum(g for g in gaps if g>0)/1e6:.3f} ms, median gap {sorted(gaps)[len(gaps)//2]/1e3:.1f} us")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, _ in fw: agg[n][0] += 1; agg[n][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]: print(f"  {n:42s} x{c:4d}  {t/1e6:8.3f} ms  avg {t/c/1e3:8.1f} us")

# ---- which kinds of kernel are in flight together (sweep line over the forward)
def kind(n):
    return "gemm" if "gemm" in n else "attn" if "attn" in n else "ln" if ("ln_kernel" in n or "embed_ln" in n) else "other"
pts = []
for s, e, n, _ in fw:
    pts.append((s, 1, kind(n))); pts.append((e, -1, kind(n)))
pts.sort()
live = collections.Counter(); state_t = collections.Counter(); last = pts[0][0]
for t, d, k in pts:
    key = "+".join(f"{kk}{live[kk] if live[kk] > 1 else ''}" for kk in sorted(live) if live[kk] > 0) or "idle"
    state_t[key] += t - last
    last = t
    live[k] += d
print("in flight together:")
for k, t in sorted(state_t.items(), key=lambda kv: -kv[1]):
    if t > 0: print(f"  {k:24s} {t/1e6:8.3f} ms")
