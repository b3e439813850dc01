#!/bin/bash
mkdir -p gpurun_out/r03p
IMCUI_ATTN_VARIANT=5 timeout 600 python -m pytest tests/test_gpu_lightglue.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 < /dev/null | tail -12 > gpurun_out/r03p/pytest_var5.log
tail -4 gpurun_out/r03p/pytest_var5.log
for v in 0 5 0 5; do
  IMCUI_ATTN_VARIANT=$v timeout 200 python bench.py --no-parity --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 >> gpurun_out/r03p/bench_splg_var$v.json.log
done
IMCUI_ATTN_VARIANT=5 timeout 300 python bench.py --workload dust3r --no-parity --no-cpu-baseline 2>/dev/null < /dev/null | tail -1 > gpurun_out/r03p/bench_dust3r_var5.json.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03p/bench_*.json.log')):
    for l in open(f).read().strip().splitlines():
        try:
            j=json.loads(l); r=j.get('roofline') or {}; print(f.split('/')[-1], round(j['value'],1), round(j['ms_per_step'],2), r.get('avg_launch_ms'), r.get('class_ms_per_step'))
        except Exception as e: print(f, 'ERR', e)
P
