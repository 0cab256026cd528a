"""Per-kernel summary of one rocprofv3 counter pass (the committed form of a pass: the raw counter_collection.csv is large):
    python tools/pmc_summary.py gpurun_out/r04c/pmc_f FETCH_SIZE > profiles/r04_pmc_fetch_size.csv
Columns: kernel, launches, mean and max of the counter per launch (FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE counts
64 B per 128-B request on gfx950: x2, applied by tools/pmc_collect.py, not here)."""
import collections, csv, glob, sys

d, counter = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == counter:
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
w = csv.writer(sys.stdout)
w.writerow(["kernel", "launches", f"mean_{counter}_KiB", f"max_{counter}_KiB"])
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    w.writerow([k, len(v), round(sum(v) / len(v), 1), round(max(v), 1)])
