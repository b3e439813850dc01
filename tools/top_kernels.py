"""Print the top kernels of a rocprofv3 --stats output directory (kernel_stats csv): name, calls, total ms, average us, percent."""
import csv
import glob
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 15
files = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
if not files:
    print("no kernel_stats.csv under", d)
    sys.exit(0)
rows = list(csv.DictReader(open(files[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f'{r["Name"][:90]:90s} {int(r["Calls"]):6d} {float(r["TotalDurationNs"]) / 1e6:9.2f} ms {float(r["AverageNs"]) / 1e3:9.1f} us {100 * float(r["TotalDurationNs"]) / tot:5.1f} %')
