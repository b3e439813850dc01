#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_c10; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_b64 -o splg -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-legs > $O/rocprof.log 2>&1 < /dev/null
python3 $R/tools/top_kernels.py $O/stats_b64 8 | cut -c1-160
