#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_c4; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_attention_mx.py -q -p no:cacheprovider -s > $O/pytest_mx.log 2>&1; grep -E "variant 9|parity|passed|failed" $O/pytest_mx.log | cut -c1-200 | tail -30 )
cd /tmp && export TMPDIR=/tmp
for v in 9 8; do
IMCUI_ATTN_VARIANT=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_v$v -o splg -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-legs > $O/rocprof_v$v.log 2>&1 < /dev/null
echo "== variant $v"; python3 $R/tools/top_kernels.py $O/stats_v$v 12 | cut -c1-150
done
