#!/bin/bash
# scratch driver of one gpurun call (round 4); not part of the product
O=gpurun_out/r04b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; grep -v "^\[parity\]\|^$" $O/pytest.log | tail -25; grep "^\[parity\] LoFTR\|^\[parity\] DUSt3R" $O/pytest.log | head -40
timeout 600 python tools/attn_pv2_audit.py > $O/attention_pv2.txt 2>$O/attention_pv2.err; echo "audit rc $?"; head -20 $O/attention_pv2.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-legs > $O/bench_headline.json.log 2>$O/bench_headline.err; echo "bench rc $?"; tail -c 1500 $O/bench_headline.json.log
