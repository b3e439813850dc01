#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06_c3; mkdir -p $O; cd $R
( timeout 900 python -m pytest tests/test_gpu_attention_mx.py -q -p no:cacheprovider -s > $O/pytest_mx.log 2>&1; tail -25 $O/pytest_mx.log | cut -c1-220 )
( IMCUI_ATTN_VARIANT=9 timeout 300 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline > $O/bench_v9.json.log 2> $O/bench_v9.err; tail -1 $O/bench_v9.json.log | cut -c1-200; tail -2 $O/bench_v9.err | cut -c1-300 )
( timeout 300 python bench.py --steps 10 --warmup 3 --no-legs --no-cpu-baseline > $O/bench_base.json.log 2> $O/bench_base.err; tail -1 $O/bench_base.json.log | cut -c1-200 )
python - <<'PY'
import json
for n in ("v9","base"):
    try:
        j=json.loads(open(f"gpurun_out/r06_c3/bench_{n}.json.log").read().strip().split("\n")[-1])
        print(n, j["value"], j["kernel_time_ms_per_step"], j["roofline"]["avg_launch_ms"], j.get("parity",{}).get("max_score_error"))
    except Exception as e: print(n, "ERR", e)
PY
