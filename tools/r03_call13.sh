#!/bin/bash
mkdir -p gpurun_out/r03m
timeout 600 python -m pytest tests/test_gpu_dust3r.py -x -q -m gpu -k "nn_argmax or mast3r" 2>&1 < /dev/null | tail -8 > gpurun_out/r03m/pytest.log
tail -4 gpurun_out/r03m/pytest.log
timeout 300 python bench.py --workload mast3r --batch 8 --no-parity --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/r03m/bench_mast3r_b8.json.log
timeout 300 python bench.py --workload mast3r --no-parity --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/r03m/bench_mast3r_b16.json.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03m/bench_*.json.log')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value'],1), round(j['ms_per_step'],2), {k:v for k,v in j['config'].items() if 'match' in k or 'ms' in k})
    except Exception as e: print(f, 'ERR', e)
P
