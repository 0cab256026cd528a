"""Per-shape means of rocprofv3 --pmc passes over tools/probe/gemm_pp_sweep (order of its launches: for each of its four
row counts, for qkv / out / fc1 / fc2: 39 launches).  python tools/pmc_gemm_sweep.py DIR [DIR ...]"""
import collections, csv, glob, sys

M = [32768, 32896, 65536, 65792]
SH = ["qkv", "out", "fc1", "fc2"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "gemm_bf16_pp_kernel" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    pos = {d_: i for i, d_ in enumerate(ids)}
    for r in rows:
        i = pos[int(r["Dispatch_Id"])] // 39
        if i < 16:
            acc[(M[i // 4], SH[i % 4])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (m, sh), cs in sorted(acc.items()):
    if m not in (32896, 65792):
        continue
    print(f"M={m} {sh}")
    for c, v in sorted(cs.items()):
        print(f"  {c:30s} {sum(v) / len(v):16.1f}   (n={len(v)})")
