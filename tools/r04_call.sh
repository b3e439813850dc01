#!/bin/bash
# same-box A/B: gemm.hip built with and without -fno-slp-vectorize (packed-f32 code generation in rotary epilogues)
O=gpurun_out/r04f; mkdir -p $O
run() { for w in "" "--workload loftr" "--workload dust3r"; do timeout 300 python bench.py $w --no-cpu-baseline --no-parity --no-legs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['metric'][:40], round(d['value'],2), d.get('kernel_time_ms_per_step') or r.get('class_ms_per_step') or r.get('gemm_ms_per_step'))"; done; }
run default
L=image-matching-webui_amd/imcui_hip/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Iinclude -Iimage-matching-webui_amd/csrc -fno-slp-vectorize -c image-matching-webui_amd/csrc/gemm.hip -o $L/obj/gemm.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $L/obj/*.o -o $L/libimcui_hip.so && echo rebuilt
run noslp
timeout 600 python -m pytest tests/test_gpu_round3_kernels.py tests/test_gpu_loftr.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
