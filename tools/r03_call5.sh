#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider -rP > $O/pytest.log 2>&1
echo "pytest rc $?"; grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20
grep "\[parity\]" $O/pytest.log | head -60 > $O/parity_lines.txt; wc -l $O/parity_lines.txt
timeout 400 python bench.py > $O/bench_splg.json.log 2>&1; tail -1 $O/bench_splg.json.log | cut -c1-400; tail -1 $O/bench_splg.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PARITY', d.get('parity'))"
timeout 200 python bench.py --workload nn > $O/bench_nn.json.log 2>&1; tail -1 $O/bench_nn.json.log | cut -c1-1500
timeout 200 python bench.py --adaptive --no-cpu-baseline > $O/bench_splg_adaptive.json.log 2>&1; tail -1 $O/bench_splg_adaptive.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ADAPTIVE', d['value'], d['roofline'].get('frac'), d.get('parity',{}).get('status'))"
timeout 300 python bench.py --workload eloftr > $O/bench_eloftr.json.log 2>&1; tail -1 $O/bench_eloftr.json.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ELOFTR', d['value'], d.get('parity'))"
