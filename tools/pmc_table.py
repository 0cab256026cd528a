"""Aggregate a rocprofv3 --pmc counter_collection csv: mean counter value per kernel name."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if pat and pat not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
