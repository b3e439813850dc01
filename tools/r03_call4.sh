#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
timeout 300 python tools/r03_diag2.py > $O/diag2.log 2>&1; grep -v "float64" $O/diag2.log | tail -12
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -12 $O/pytest.log
b() { local name=$1; shift
  ( env "$@" timeout 200 python bench.py --no-cpu-baseline $BARGS > $O/bench_$name.json.log 2>&1; tail -1 $O/bench_$name.json.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$name', round(d['value'], 1), d['unit'], d.get('kernel_time_ms_per_step'), 'frac', d.get('roofline', {}).get('frac'))
except Exception as e: print('$name', 'FAILED', e)" )
}
BARGS=""
b wreg2_epi A=1
b wreg2_pass IMCUI_LG_ASSIGN_STATS=pass
b wreg1_pass IMCUI_LG_ASSIGN_STATS=pass IMCUI_GEMM_WREG=1
b wreg0_pass IMCUI_LG_ASSIGN_STATS=pass IMCUI_GEMM_WREG=0
cd /tmp && export TMPDIR=/tmp
IMCUI_LG_ASSIGN_STATS=pass timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_splg -o splg -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_splg.log 2>&1
grep -i "lg_\|gemm" $O/stats_splg/splg_kernel_stats.csv | cut -c1-150
find $O -name "*kernel_trace.csv" -delete
