#!/bin/bash
# DPT head fusions (relu on load, double residual, out_conv before the x2) + batch sweep of the DUSt3R bench
mkdir -p gpurun_out/r03k
timeout 900 python -m pytest tests/test_gpu_dust3r.py tests/test_gpu_loftr.py tests/test_gpu_eloftr.py tests/test_gpu_superpoint.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03k/pytest.log
tail -5 gpurun_out/r03k/pytest.log
python bench.py --workload dust3r --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03k/bench_dust3r_b8.json.log
IMCUI_DUST3R_HEAD_UNFUSED=1 python bench.py --workload dust3r --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03k/bench_dust3r_b8_head_unfused.json.log
python bench.py --workload dust3r --batch 16 --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03k/bench_dust3r_b16.json.log
python bench.py --workload dust3r --batch 32 --steps 5 --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03k/bench_dust3r_b32.json.log
python bench.py --workload dust3r --batch 16 --arith fp16 --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03k/bench_dust3r_b16_fp16.json.log
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03k/bench_*.json.log')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value'],1), round(j['ms_per_step'],2), (j.get('roofline') or {}).get('class_ms_per_step'))
    except Exception as e: print(f, 'ERR', e)
P
