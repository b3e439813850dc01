#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_c2; mkdir -p $O; cd $R
( timeout 300 tools/wreg_lab > $O/wreg_lab.txt 2>&1; echo "wreg_lab rc $?"; grep -E "full kernel|late" $O/wreg_lab.txt )
( timeout 300 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline > $O/bench_splg.json.log 2> $O/bench_splg.err; tail -1 $O/bench_splg.json.log | cut -c1-300 )
