#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
timeout 600 python tools/r03_diag.py > $O/diag.log 2>&1; tail -60 $O/diag.log
for m in 0 1 2; do
  IMCUI_GEMM_WREG=$m timeout 600 python -m pytest tests/test_gpu_superglue.py tests/test_gpu_lightglue.py -m gpu -q -k "replicas" -p no:cacheprovider > $O/replicas_wreg$m.log 2>&1
  echo "WREG=$m: $(tail -1 $O/replicas_wreg$m.log)"
done
b() { local name=$1; shift
  ( env "$@" timeout 200 python bench.py --no-cpu-baseline $BARGS > $O/bench_$name.json.log 2>&1; tail -1 $O/bench_$name.json.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$name', round(d['value'], 1), d['unit'], d.get('kernel_time_ms_per_step'), 'frac', d.get('roofline', {}).get('frac'))
except Exception as e: print('$name', 'FAILED', e)" )
}
BARGS=""
b epi A=1
b pass IMCUI_LG_ASSIGN_STATS=pass
cd /tmp && export TMPDIR=/tmp
IMCUI_LG_ASSIGN_STATS=pass timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_splg -o splg -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_splg.log 2>&1
grep -i "lg_\|gemm" $O/stats_splg/splg_kernel_stats.csv | cut -c1-160
find $O -name "*kernel_trace.csv" -delete
