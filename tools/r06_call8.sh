#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_c8; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_lightglue.py tests/test_gpu_kernels.py tests/test_gpu_round3_kernels.py tests/test_gpu_superglue.py -q -p no:cacheprovider -x > $O/pytest_a.log 2>&1; tail -4 $O/pytest_a.log | cut -c1-250 )
( timeout 300 python bench.py --batch 1 --steps 30 --warmup 3 --no-legs --no-cpu-baseline > $O/bench_b1.json.log 2> $O/bench_b1.err; tail -1 $O/bench_b1.json.log | cut -c1-200 )
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b1 -o splg -- python $R/bench.py --batch 1 --steps 30 --warmup 3 --no-cpu-baseline --no-parity --no-legs > $O/rocprof_b1.log 2>&1 < /dev/null
python3 $R/tools/top_kernels.py $O/stats_b1 24 | cut -c1-160
cd $R
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_gpu_lightglue.py --deselect tests/test_gpu_kernels.py --deselect tests/test_gpu_round3_kernels.py --deselect tests/test_gpu_superglue.py > $O/pytest_b.log 2>&1; tail -4 $O/pytest_b.log | cut -c1-250 )
