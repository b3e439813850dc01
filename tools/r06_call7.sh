#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_c7; mkdir -p $O; cd $R
( timeout 600 python -m pytest tests/test_gpu_lightglue.py -q -p no:cacheprovider -x -k "alone or full_size" > $O/pytest_lg.log 2>&1; tail -4 $O/pytest_lg.log | cut -c1-250 )
( timeout 300 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline > $O/bench_base.json.log 2> $O/bench_base.err; tail -1 $O/bench_base.json.log | cut -c1-200 )
( timeout 300 python bench.py --batch 1 --steps 30 --warmup 3 --no-legs --no-cpu-baseline > $O/bench_b1.json.log 2> $O/bench_b1.err; tail -1 $O/bench_b1.json.log | cut -c1-200 )
( IMCUI_ATTN_SPLIT=0 timeout 300 python bench.py --batch 1 --steps 30 --warmup 3 --no-legs --no-cpu-baseline --no-parity > $O/bench_b1_nosplit.json.log 2> $O/bench_b1_nosplit.err; tail -1 $O/bench_b1_nosplit.json.log | cut -c1-200 )
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-250 )
