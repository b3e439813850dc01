#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_dust3r.py -m gpu -q --maxfail=25 -p no:cacheprovider -rP > $O/pytest_dust3r.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed" $O/pytest_dust3r.log | tail -3; grep -E "^FAILED|^ERROR|\[anchor\]" $O/pytest_dust3r.log | head -20
b() { local name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline $BARGS > $O/bench_$name.json.log 2>&1; tail -1 $O/bench_$name.json.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$name', round(d['value'], 2), d['unit'], d['roofline'].get('class_ms_per_step') or d.get('kernel_time_ms_per_step'), 'frac', d.get('roofline', {}).get('frac'), d['config'].get('matching_ms_per_step'))
except Exception as e: print('$name', 'FAILED', e)" )
}
BARGS="--workload dust3r"
b dust3r_fused A=1
b dust3r_unfused IMCUI_DUST3R_QKV_UNFUSED=1
b dust3r_wreg0 IMCUI_GEMM_WREG=0
BARGS="--workload mast3r"
b mast3r A=1
BARGS=""
timeout 400 python bench.py --no-cpu-baseline > $O/bench_splg.json.log 2>&1; tail -1 $O/bench_splg.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SPLG', d['value'], d.get('parity'))"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dust3r -o dust3r -- python $R/bench.py --workload dust3r --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/rocprof_dust3r.log 2>&1
head -16 $O/stats_dust3r/dust3r_kernel_stats.csv | cut -c1-180
find $O -name "*kernel_trace.csv" -delete
