#!/bin/bash
# where does the MASt3R matching time go: kernel table of the mast3r workload
mkdir -p gpurun_out/r03l
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03l/prof -o mast3r -- python $GRAFT_REPO_ROOT/bench.py --workload mast3r --batch 8 --steps 3 --warmup 1 --no-parity --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03l/bench.log 2>&1 < /dev/null
cd $GRAFT_REPO_ROOT
grep '^{' gpurun_out/r03l/bench.log | tail -1 | cut -c1-300
f=$(find gpurun_out/r03l/prof -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then head -25 "$f" | cut -c1-200; cp "$f" gpurun_out/r03l/mast3r_kernel_stats.csv; fi
find gpurun_out/r03l/prof -name '*kernel_trace.csv' -delete
