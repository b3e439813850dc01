#!/bin/bash
# scratch driver of one gpurun call (round 4); not part of the product
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 600 python tools/attn_pv2_audit.py > $O/attention_pv2.txt 2>$O/attention_pv2.err; echo "audit rc $?"; tail -30 $O/attention_pv2.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json.log 2>$O/bench_default.err; echo "bench rc $?"; tail -c 3000 $O/bench_default.json.log; tail -5 $O/bench_default.err
