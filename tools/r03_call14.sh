#!/bin/bash
mkdir -p gpurun_out/r03n
for b in 64 96 128; do
  timeout 200 python bench.py --batch $b --no-parity --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/r03n/bench_splg_b$b.json.log
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03n/bench_*.json.log')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value'],1), round(j['ms_per_step'],2))
    except Exception as e: print(f, 'ERR', e)
P
