"""The clock the chip holds under each kernel of the tower, in place: GRBM_GUI_ACTIVE / 8 XCDs / the dispatch's duration
(MI355X_MICROARCH.md 'DVFS give-back': within 3 % of the in-kernel clock on long dispatches, reads high on dispatches
shorter than about 0.3 ms — both groups are printed).

    MI_CLIP_PARTS=1 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc_clk -o c --output-format csv -- \
        python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra-configs
    python tools/pmc_clock.py gpurun_out/pmc_clk > profiles/r04_tower_clock_pmc.json"""
import collections, csv, glob, json, sys

d = sys.argv[1]
cc = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(cc)))
dur = {}
if rows and "Start_Timestamp" not in rows[0]:
    kt = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
acc = collections.defaultdict(lambda: {"long": [], "short": [], "us": []})
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    ns = dur.get(r["Dispatch_Id"]) if dur else int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if not ns:
        continue
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    ghz = float(r["Counter_Value"]) / 8.0 / ns
    acc[name]["long" if ns >= 300_000 else "short"].append(ghz)
    acc[name]["us"].append(ns / 1e3)
out = {"source": cc, "recipe": "GRBM_GUI_ACTIVE / 8 / duration per dispatch; 'long' = dispatches of 0.3 ms or more", "kernels": {}}
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]["us"])):
    if "mi::" not in k:
        continue
    m = lambda a: round(sum(a) / len(a), 3) if a else None
    out["kernels"][k] = {"dispatches": len(v["us"]), "avg_us_under_the_counter": round(sum(v["us"]) / len(v["us"]), 1),
                         "GHz_long_dispatches": m(v["long"]), "n_long": len(v["long"]), "GHz_short_dispatches": m(v["short"])}
print(json.dumps(out, indent=1))
