#!/bin/bash
# lab: kernel table of the LoFTR step at ~2000 matches per pair (coarse threshold 3e-5 on the synthetic pair), 8 pairs per step
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_c18
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
IMCUI_BENCH_LOFTR_THR=0.00003 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o lf -- python $R/bench.py --workload loftr --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-legs > $O/rocprof.log 2>&1 < /dev/null
cd $R && python3 tools/top_kernels.py $O/stats 45
tail -1 $O/rocprof.log | cut -c1-300
