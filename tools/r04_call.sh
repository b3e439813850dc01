#!/bin/bash
O=gpurun_out/final_r04; mkdir -p $O
for v in 6; do
  IMCUI_ATTN_VARIANT=$v timeout 1200 python -m pytest tests/test_gpu_lightglue.py tests/test_gpu_real_images.py tests/test_gpu_auc_parity.py tests/test_gpu_superglue.py tests/test_gpu_vs_hf_ports.py tests/test_gpu_match_driver.py tests/test_gpu_dust3r.py -m gpu -q -p no:cacheprovider -s > $O/pytest_attn_v$v.log 2>&1
  echo "variant $v rc $?"; grep -E "passed|failed" $O/pytest_attn_v$v.log | tail -2; grep -E "^FAILED|AssertionError" $O/pytest_attn_v$v.log | cut -c1-260 | head -30
done
