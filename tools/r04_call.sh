#!/bin/bash
for b in 4 8 16; do timeout 300 python bench.py --workload loftr --batch $b --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('loftr B=$b', round(d['value'],2), round(d['ms_per_step'],2), d['roofline']['gemm_ms_per_step'])"; done
for b in 32 64 128 256; do timeout 300 python bench.py --workload superpoint --batch $b --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('superpoint pairs B=$b', round(d['value'],1), round(d['ms_per_step'],2), d['roofline']['conv_ms_per_step'])"; done
for b in 8 16 32; do timeout 300 python bench.py --workload dust3r --batch $b --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('dust3r B=$b', round(d['value'],2), round(d['ms_per_step'],2))"; done
