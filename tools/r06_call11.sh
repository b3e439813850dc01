#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_c11; mkdir -p $O; cd $R
for b in 8 16 32; do ( timeout 200 python bench.py --workload eloftr --batch $b --no-legs --no-cpu-baseline --no-parity > $O/eloftr_b$b.json.log 2> $O/eloftr_b$b.err; python3 -c "
import json,sys
j=json.loads(open('$O/eloftr_b$b.json.log').read().strip().split('\n')[-1]); print('eloftr batch $b', round(j['value'],1), j['ms_per_step'])" ); done
( timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_full.json.log 2> $O/bench_full.err; python3 - <<'PY'
import json
j=json.loads(open("gpurun_out/r06_c11/bench_full.json.log").read().strip().split("\n")[-1])
print("headline", round(j["value"],1), j["kernel_time_ms_per_step"], j["roofline"]["frac"])
for k,v in j.get("legs",{}).items(): print(k, v.get("value"), v.get("unit"), v.get("ms_per_step"), (v.get("roofline") or {}).get("frac") if isinstance(v.get("roofline"),dict) else v.get("roofline_frac"))
PY
)
