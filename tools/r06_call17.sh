#!/bin/bash
# round 6, second session: full GPU suite, smoke, driver-shaped bench at HEAD
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06_c17
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench.json.log 2> $O/bench.err; tail -1 $O/bench.json.log | cut -c1-400
tail -1 $O/bench.json.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('HEADLINE', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('kernel_time_ms_per_step'))
for k,v in d.get('legs',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('frac'), v.get('parity'), v.get('fine_stage'), v.get('status'))
"
