#!/bin/bash
mkdir -p gpurun_out/r03r
IMCUI_DUST3R_LN_EPILOGUE=1 timeout 600 python -m pytest tests/test_gpu_dust3r.py -x -q -m gpu 2>&1 < /dev/null | tail -40 > gpurun_out/r03r/pytest_fold.log
tail -8 gpurun_out/r03r/pytest_fold.log
for e in 0 1; do
  IMCUI_DUST3R_LN_EPILOGUE=$e timeout 300 python bench.py --workload dust3r --no-parity --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/r03r/bench_dust3r_ln$e.json.log
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03r/bench_*.json.log')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value'],1), round(j['ms_per_step'],2), (j.get('roofline') or {}).get('class_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
P
