#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_c1; mkdir -p $O; cd $R
( timeout 120 tools/mx_lab > $O/mx_lab.txt 2>&1; echo "mx_lab rc $?" )
( timeout 200 tools/wreg_lab > $O/wreg_lab.txt 2>&1; echo "wreg_lab rc $?" )
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log )
( timeout 300 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline > $O/bench_splg.json.log 2> $O/bench_splg.err; tail -1 $O/bench_splg.json.log | cut -c1-300 )
