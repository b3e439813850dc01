#!/bin/bash
# round 6, second session, call 12: match-sparse last FPN stage of LoFTR (parity + A/B timing), single-rank RCCL launch check
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_c12
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_loftr.py -x -q -p no:cacheprovider -s > $O/pytest_loftr.log 2>&1; tail -5 $O/pytest_loftr.log
grep "\[parity\]" $O/pytest_loftr.log | cut -c1-230
timeout 600 python -m pytest tests/test_gpu_rccl_single_rank.py -x -q -p no:cacheprovider > $O/pytest_rccl.log 2>&1; tail -5 $O/pytest_rccl.log
for mode in "" "--fine-dense"; do
  timeout 400 python bench.py --workload loftr --no-legs --no-cpu-baseline $mode > $O/bench_loftr$mode.json.log 2>$O/bench_loftr$mode.err
  tail -1 $O/bench_loftr$mode.json.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('fine_stage'))"
done
IMCUI_LOFTR_FINE_SPARSE=2 timeout 400 python bench.py --workload loftr --size 480 640 --no-legs --no-cpu-baseline > $O/bench_loftr_640_sparse.json.log 2>&1; tail -1 $O/bench_loftr_640_sparse.json.log | cut -c1-200
timeout 400 python bench.py --workload loftr --size 480 640 --no-legs --no-cpu-baseline --fine-dense > $O/bench_loftr_640_dense.json.log 2>&1; tail -1 $O/bench_loftr_640_dense.json.log | cut -c1-200
