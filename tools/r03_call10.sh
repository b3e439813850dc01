#!/bin/bash
# 16-row conv tiles: parity tests of everything that runs 3x3 convolutions, then A/B bench legs
mkdir -p gpurun_out/r03j
timeout 1500 python -m pytest tests/test_gpu_superpoint.py tests/test_gpu_loftr.py tests/test_gpu_eloftr.py tests/test_gpu_kernels.py tests/test_gpu_real_images.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03j/pytest.log
tail -5 gpurun_out/r03j/pytest.log
for m in 0 1 2; do
  IMCUI_CONV_TALL=$m python bench.py --no-parity 2>/dev/null | tail -1 > gpurun_out/r03j/bench_splg_tall$m.json.log
  IMCUI_CONV_TALL=$m python bench.py --workload superpoint --no-parity 2>/dev/null | tail -1 > gpurun_out/r03j/bench_sp_tall$m.json.log
done
for m in 0 2; do
  IMCUI_CONV_TALL=$m python bench.py --workload eloftr --no-parity 2>/dev/null | tail -1 > gpurun_out/r03j/bench_eloftr_tall$m.json.log
  IMCUI_CONV_TALL=$m python bench.py --workload loftr --no-parity 2>/dev/null | tail -1 > gpurun_out/r03j/bench_loftr_tall$m.json.log
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03j/bench_*.json.log')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value'],1), j['ms_per_step'], j.get('roofline',{}) and {k:v for k,v in j['roofline'].items() if 'ms' in k})
    except Exception as e: print(f, 'ERR', e)
P
