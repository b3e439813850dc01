#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider -rP > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR|\[anchor\]" $O/pytest.log | head -20
b() { local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline $BARGS > $O/bench_$name.json.log 2>&1; tail -1 $O/bench_$name.json.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$name', round(d['value'], 2), d['unit'], d['roofline'].get('class_ms_per_step') or d.get('kernel_time_ms_per_step'), 'frac', d.get('roofline', {}).get('frac'), (d.get('parity') or {}).get('status'))
except Exception as e: print('$name', 'FAILED', e)" )
}
BARGS="--workload dust3r"
b dust3r A=1
BARGS="--workload dust3r --arith fp16"
b dust3r_fp16 A=1
BARGS="--workload eloftr"
b eloftr A=1
BARGS=""
b splg A=1
