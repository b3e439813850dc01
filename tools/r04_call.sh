#!/bin/bash
O=gpurun_out/final_r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_jpeg.py -m gpu -q -p no:cacheprovider 2>&1 | tail -4
timeout 300 python tools/jpeg_bench.py > $O/lab_jpeg.txt 2>$O/lab_jpeg.err; cat $O/lab_jpeg.txt; tail -3 $O/lab_jpeg.err
( timeout 300 python bench.py --h2d jpeg --no-cpu-baseline > $O/bench_splg_h2d_jpeg.json.log 2>$O/bench_splg_h2d_jpeg.err; tail -1 $O/bench_splg_h2d_jpeg.json.log | cut -c1-150; tail -3 $O/bench_splg_h2d_jpeg.err )
( timeout 300 python bench.py --h2d raw --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | cut -c1-150 )
( timeout 300 python bench.py --no-legs --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | cut -c1-150 )
