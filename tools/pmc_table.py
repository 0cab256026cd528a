"""Per-kernel means of rocprofv3 --pmc counter passes: python tools/pmc_table.py DIR [DIR ...] [--kernel SUBSTR]
Each DIR holds one pass (rocprofv3 --pmc ... --kernel-trace -d DIR --output-format csv)."""
import collections, csv, glob, sys

dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
sub = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--kernel=")), "")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if sub in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"  {c:34s} {sum(v) / len(v):16.1f}   (n={len(v)})")
