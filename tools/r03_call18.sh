#!/bin/bash
mkdir -p gpurun_out/r03q
timeout 600 python -m pytest tests/test_gpu_dust3r.py -x -q -m gpu 2>&1 < /dev/null | tail -25 > gpurun_out/r03q/pytest.log
tail -6 gpurun_out/r03q/pytest.log
timeout 300 python bench.py --workload dust3r --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/r03q/bench_dust3r.json.log
timeout 300 python bench.py --workload dust3r --arith fp16 --no-parity --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/r03q/bench_dust3r_fp16.json.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03q/bench_*.json.log')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value'],1), round(j['ms_per_step'],2), (j.get('parity') or {}).get('status'), (j.get('roofline') or {}).get('class_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
P
