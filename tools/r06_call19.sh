#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_c19
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_loftr.py -x -q -p no:cacheprovider > $O/pytest_loftr.log 2>&1; tail -3 $O/pytest_loftr.log
timeout 400 python bench.py --workload loftr --no-legs --no-cpu-baseline > $O/bench_loftr.json.log 2>$O/bench_loftr.err
tail -1 $O/bench_loftr.json.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], json.dumps(d['config'].get('fine_stage'), indent=1))"
LAB_B=8 timeout 600 python tools/loftr_fine_lab.py 2>/dev/null | tee $O/lab.txt
