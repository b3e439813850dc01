#!/bin/bash
# scratch driver of one gpurun call (round 4); not part of the product
O=gpurun_out/r04c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_jpeg.py tests/test_gpu_batch_drivers.py tests/test_gpu_dust3r.py::test_dust3r_plugin_output_structure -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
for v in 0 8 0 8; do
  IMCUI_ATTN_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-legs --no-parity --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_attn_v$v.$RANDOM.json
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob('gpurun_out/r04c/bench_attn_v*.json')):
    d = json.loads(open(f).read())
    print(f.split('/')[-1][:14], round(d['value'],1), 'pairs/s', round(d['roofline']['avg_launch_ms'],4), 'ms/attn launch', d['kernel_time_ms_per_step'])
PY
timeout 600 python bench.py --steps 20 --warmup 5 --h2d jpeg --no-cpu-baseline > $O/bench_splg_h2d_jpeg.json.log 2>$O/bench_h2d_jpeg.err; echo "h2d jpeg rc $?"; tail -c 1200 $O/bench_splg_h2d_jpeg.json.log; tail -5 $O/bench_h2d_jpeg.err
timeout 300 python bench.py --steps 20 --warmup 5 --h2d raw --no-cpu-baseline --no-parity 2>/dev/null | tail -1 | cut -c1-200
