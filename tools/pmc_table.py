"""Per-kernel means of the counters of a rocprofv3 --pmc run (counter_collection.csv under a directory); optional name filter."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
if not files:
    print("no counter_collection.csv under", d)
    sys.exit(0)
by = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if flt in k:
            by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(by.items()):
    n = max(len(v) for v in c.values())
    m = {name: sum(v) / len(v) for name, v in c.items()}
    line = f"{k[:60]:60s} launches {n:4d} "
    if "SQ_BUSY_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"] > 0:
        # busy cycles are summed over SEs (32) / MFMA busy over SIMDs: the ratio the DESIGN tables use = mfma_busy / (busy_cycles / 32 * 1024)
        line += f' mfma_busy {m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["SQ_BUSY_CYCLES"] / 32.0 * 1024.0):.3f}'
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
        w = m["SQ_WAVE_CYCLES"]
        for name, tag in (("SQ_ACTIVE_INST_VALU", "valu"), ("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"), ("SQ_ACTIVE_INST_LDS", "lds")):
            if name in m:
                line += f" {tag} {m[name] / w:.3f}"
    for name in ("SQ_LDS_BANK_CONFLICT", "FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_INSTS_LDS"):
        if name in m:
            line += f" {name} {m[name]:.4g}"
    print(line)
