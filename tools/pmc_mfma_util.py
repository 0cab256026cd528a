"""MFMA utilisation of the tower's kernels IN SITU from a rocprofv3 --pmc pass of bench.py (single stream, so kernels do not overlap):

  MI_CLIP_PARTS=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d DIR -o m --output-format csv -- \\
      python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs
  python tools/pmc_mfma_util.py DIR

Per kernel: utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) — the matrix pipes' busy cycles over the
cycles the launch lasted at the clock the chip held (GRBM_GUI_ACTIVE is summed over the 8 XCDs; MI355X_MICROARCH.md: the quotient
over-reads the clock on dispatches under 0.3 ms, which cancels here: both counters are in the same cycles)."""
import collections, csv, glob, sys

f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':58s} {'launches':>8s} {'MFMA busy / launch':>20s} {'cycles / launch':>16s} {'utilisation':>12s}")
tot_busy = tot_cyc = 0.0
for k, cs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    if "mi::" not in k or "knn" in k or "gen_f32" in k:
        continue
    busy, gui = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", []), cs.get("GRBM_GUI_ACTIVE", [])
    if not gui:
        continue
    b, c = sum(busy) / max(len(busy), 1), sum(gui) / len(gui) / 8.0
    tot_busy += sum(busy); tot_cyc += sum(gui) / 8.0
    print(f"{k[:58]:58s} {len(gui):8d} {b:20.0f} {c:16.0f} {b / (c * 1024) if c else 0:12.4f}")
print(f"{'all tower kernels':58s} {'':8s} {'':20s} {'':16s} {tot_busy / (tot_cyc * 1024):12.4f}")
