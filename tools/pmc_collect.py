"""Builds profiles/pmc_latest.json (what bench.py reports as roofline.traffic) from two rocprofv3
counter passes of the default bench, collected separately as the guide prescribes:

  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python tools/pmc_collect.py gpurun_out/pmc_f gpurun_out/pmc_w

gfx950 corrections: FETCH_SIZE counts 64 B per 128-B request (x2); both counters are in KiB (x1024)."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
# the single-pass scan is also enqueued behind every two-stage query and returns at once when it is not needed:
# only its real launches (more than half of the largest one's bytes) count
for acc in (fetch, write):
    for k in list(acc):
        if "knn_scan_kernel" in k and acc[k]:
            top = max(acc[k])
            acc[k] = [v for v in acc[k] if v > 0.5 * top] or acc[k]
pk = {}
for k in fetch:
    pk[k] = {"launches": len(fetch[k]),
             "fetch_bytes_per_launch_x2": sum(fetch[k]) / len(fetch[k]) * 1024 * 2,
             "write_bytes_per_launch": (sum(write[k]) / len(write[k]) * 1024) if k in write else None}
scan = next(v for k, v in pk.items() if "knn_scan_kernel" in k)
vit = [k for k in pk if "mi::" in k and "knn" not in k and "gen_f32" not in k]
forwards = next(v["launches"] for k, v in pk.items() if "head_kernel" in k) / 2  # two half-chunk streams per forward
vit_bytes = sum(pk[k]["launches"] * (pk[k]["fetch_bytes_per_launch_x2"] + (pk[k]["write_bytes_per_launch"] or 0)) for k in vit) / forwards
out = {
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (two passes)",
    "correction": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), x1024 B; WRITE_SIZE x1024 B",
    "rows": 10000000, "batch": 256,
    "knn_scan_hbm_bytes": scan["fetch_bytes_per_launch_x2"] + (scan["write_bytes_per_launch"] or 0),
    "knn_algorithmic_bytes": 10000000 * 768 * 4,
    "knn_two_stage_stage1_hbm_bytes": next((v["fetch_bytes_per_launch_x2"] + (v["write_bytes_per_launch"] or 0)
                                            for k, v in pk.items() if "knn_scan_coarse" in k), None),
    "knn_two_stage_stage1_algorithmic_bytes": 10000000 * (768 + 12 + 4),
    "vit_hbm_bytes": vit_bytes,
    "note_vit": "fabric-side requests of all tower kernels of one 256-image forward; includes Infinity-Cache hits (weights, re-read X panels)",
    "per_kernel": pk,
}
with open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "per_kernel"}, indent=1))
