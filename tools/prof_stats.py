"""Print the rocprofv3 kernel_stats csv (first match under a directory) as a compact table."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
print(f)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(6), f'{float(r["AverageNs"])/1e3:10.1f} us', r["Percentage"].rjust(7))
