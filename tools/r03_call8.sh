#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_round3_kernels.py tests/test_gpu_kernels.py tests/test_gpu_lightglue.py tests/test_gpu_dust3r.py tests/test_gpu_superglue.py -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
b() { local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-parity $BARGS > $O/bench_$name.json.log 2>&1; tail -1 $O/bench_$name.json.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$name', round(d['value'], 2), d['unit'], d['roofline'].get('class_ms_per_step') or d.get('kernel_time_ms_per_step'), 'frac', d.get('roofline', {}).get('frac'))
except Exception as e: print('$name', 'FAILED', e)" )
}
BARGS="--workload dust3r"
b dust3r_pipe A=1
b dust3r_rolled IMCUI_WREG_PIPE=0
BARGS=""
b splg_pipe A=1
b splg_rolled IMCUI_WREG_PIPE=0
BARGS="--workload superglue"
b superglue A=1
