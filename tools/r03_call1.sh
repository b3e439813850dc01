#!/bin/bash
# Round 3, first GPU call: parity of the new kernels + A/B bench legs + kernel tables.  Writes under gpurun_out/r03a/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc $?"; tail -15 $O/pytest.log
b() { # name, env..., -- args
  local name=$1; shift
  ( env "$@" timeout 200 python bench.py --no-cpu-baseline $BARGS > $O/bench_$name.json.log 2>&1; tail -1 $O/bench_$name.json.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$name', round(d['value'], 1), d['unit'], d.get('kernel_time_ms_per_step'), 'frac', d.get('roofline', {}).get('frac'))
except Exception as e: print('$name', 'FAILED', e)" )
}
BARGS=""
b new A=1
b wreg_off IMCUI_GEMM_WREG=0
b stats_pass IMCUI_LG_ASSIGN_STATS=pass
b attn1 IMCUI_ATTN_VARIANT=1
b attn2 IMCUI_ATTN_VARIANT=2
b attn3 IMCUI_ATTN_VARIANT=3
BARGS="--workload dust3r"
b dust3r_new A=1
b dust3r_wreg_qkv_only IMCUI_GEMM_WREG=1
BARGS="--workload superglue"
b superglue A=1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_splg -o splg -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/rocprof_splg.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dust3r -o dust3r -- python $R/bench.py --workload dust3r --steps 3 --warmup 1 --no-cpu-baseline > $O/rocprof_dust3r.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_SQ -o splg -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_SQ.log 2>&1
echo pmc SQ rc $?
for f in $O/stats_splg/*kernel_stats.csv $O/stats_dust3r/*kernel_stats.csv; do echo "== $f"; head -14 "$f" | cut -c1-200; done
# keep the merged output small: drop the raw traces, keep the stats tables
find $O -name "*kernel_trace.csv" -size +20M -delete
ls $O
