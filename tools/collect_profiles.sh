#!/bin/bash
# Round evidence collector: run on the GPU box (gpurun), writes under gpurun_out/final/.
# rocprofv3 needs cwd=/tmp and TMPDIR=/tmp on this pool; counters are collected one per pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${OUT:-final_r03}
mkdir -p $O
# LG_ONLY=1: only the SuperPoint+LightGlue legs (use after a change that cannot affect LoFTR / SuperGlue / the lab binaries)
# SKIP_LABS=1: everything except the stand-alone lab programs (their output does not depend on the library)
L=${LG_ONLY:-0}
K=${SKIP_LABS:-$L}
cd /tmp && export TMPDIR=/tmp
# QUICK=1: the four bench lines (headline, LoFTR 1024^2 / 640x480, EfficientLoFTR, DUSt3R), their kernel tables and the DUSt3R
# counter passes only (after a change that leaves the other operating points, A/B legs and counters as they were)
if [ "${QUICK:-0}" = 1 ]; then
  ( cd $R && timeout 300 python bench.py > $O/bench_splg.json.log 2>&1; tail -1 $O/bench_splg.json.log | cut -c1-160 )
  ( cd $R && timeout 300 python bench.py --h2d --no-cpu-baseline > $O/bench_splg_h2d.json.log 2>&1; tail -1 $O/bench_splg_h2d.json.log | cut -c1-160 )
  ( cd $R && timeout 200 python bench.py --workload loftr > $O/bench_loftr_1024.json.log 2>&1; tail -1 $O/bench_loftr_1024.json.log | cut -c1-160 )
  ( cd $R && timeout 200 python bench.py --workload loftr --size 480 640 --no-cpu-baseline > $O/bench_loftr_640x480.json.log 2>&1; tail -1 $O/bench_loftr_640x480.json.log | cut -c1-160 )
  ( cd $R && timeout 200 python bench.py --workload eloftr > $O/bench_eloftr_640x480.json.log 2>&1; tail -1 $O/bench_eloftr_640x480.json.log | cut -c1-160 )
  ( cd $R && timeout 300 python bench.py --workload dust3r > $O/bench_dust3r_512.json.log 2>&1; tail -1 $O/bench_dust3r_512.json.log | cut -c1-160 )
  ( cd $R && timeout 300 python bench.py --workload dust3r --arith fp16 --no-cpu-baseline > $O/bench_dust3r_512_fp16.json.log 2>&1; tail -1 $O/bench_dust3r_512_fp16.json.log | cut -c1-160 )
  ( cd $R && timeout 300 python bench.py --workload mast3r --no-cpu-baseline > $O/bench_mast3r_512.json.log 2>&1; tail -1 $O/bench_mast3r_512.json.log | cut -c1-160 )
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_splg -o splg -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/rocprof_splg.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_loftr -o loftr -- python $R/bench.py --workload loftr --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/rocprof_loftr.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_eloftr -o eloftr -- python $R/bench.py --workload eloftr --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/rocprof_eloftr.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dust3r -o dust3r -- python $R/bench.py --workload dust3r --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/rocprof_dust3r.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mast3r -o mast3r -- python $R/bench.py --workload mast3r --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/rocprof_mast3r.log 2>&1 < /dev/null
  ( cd $R && IMCUI_DUST3R_REGRESS_UNFUSED=1 timeout 300 python bench.py --workload dust3r --no-cpu-baseline --no-parity > $O/bench_dust3r_512_regress_unfused.json.log 2>&1; tail -1 $O/bench_dust3r_512_regress_unfused.json.log | cut -c1-160 )
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_dust3r_$c -o dust3r -- python $R/bench.py --workload dust3r --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $O/pmc_dust3r_$c.log 2>&1
    echo pmc dust3r $c rc $?
  done
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_dust3r_SQ -o dust3r -- python $R/bench.py --workload dust3r --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $O/pmc_dust3r_SQ.log 2>&1
  ( cd $R && timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 )
  ls $O
  exit 0
fi
( cd $R && timeout 300 python bench.py > $O/bench_splg.json.log 2>&1; tail -1 $O/bench_splg.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && timeout 200 python bench.py --workload loftr > $O/bench_loftr_1024.json.log 2>&1; tail -1 $O/bench_loftr_1024.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && timeout 200 python bench.py --workload superpoint > $O/bench_superpoint.json.log 2>&1; tail -1 $O/bench_superpoint.json.log | cut -c1-160 )
( cd $R && timeout 200 python bench.py --precision 0 --no-cpu-baseline > $O/bench_splg_f32.json.log 2>&1; tail -1 $O/bench_splg_f32.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && timeout 200 python bench.py --workload superglue > $O/bench_superglue.json.log 2>&1; tail -1 $O/bench_superglue.json.log | cut -c1-160 )
[ $L = 1 ] || timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_superglue -o superglue -- python $R/bench.py --workload superglue --steps 3 --warmup 1 > $O/rocprof_superglue.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_splg -o splg -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > $O/rocprof_splg.log 2>&1
[ $L = 1 ] || timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_loftr -o loftr -- python $R/bench.py --workload loftr --steps 5 --warmup 2 > $O/rocprof_loftr.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o splg -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $O/pmc_$c.log 2>&1
  echo pmc $c rc $?
  [ $L = 1 ] || timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_loftr_$c -o loftr -- python $R/bench.py --workload loftr --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $O/pmc_loftr_$c.log 2>&1
  [ $L = 1 ] || timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_eloftr_$c -o eloftr -- python $R/bench.py --workload eloftr --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $O/pmc_eloftr_$c.log 2>&1
done
# matrix-pipe occupancy and stall breakdown (one pass: 7 of the 8 SQ slots)
timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_SQ -o splg -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $O/pmc_SQ.log 2>&1
echo pmc SQ rc $?
[ $K = 1 ] || timeout 60 $R/tools/clock_lab > $O/lab_clock.txt 2>&1
[ $K = 1 ] || timeout 60 $R/tools/overlap_lab > $O/lab_overlap.txt 2>&1
[ $K = 1 ] || timeout 60 $R/tools/launch_lab > $O/lab_launch.txt 2>&1
[ $K = 1 ] || timeout 60 $R/tools/gridsync_lab > $O/lab_gridsync.txt 2>&1
( cd $R && timeout 100 python bench.py --adaptive --no-cpu-baseline > $O/bench_splg_adaptive.json.log 2>&1; tail -1 $O/bench_splg_adaptive.json.log | cut -c1-160 )
( cd $R && timeout 100 python bench.py --batch 1 --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_splg_b1.json.log 2>&1; tail -1 $O/bench_splg_b1.json.log | cut -c1-160 )
( cd $R && timeout 100 python bench.py --batch 32 --no-cpu-baseline > $O/bench_splg_b32.json.log 2>&1; tail -1 $O/bench_splg_b32.json.log | cut -c1-160 )
# HIP-graph replay vs eager launches at the latency-bound operating points (reference-default adaptive LightGlue)
for b in 1 4; do
  ( cd $R && timeout 100 python bench.py --batch $b --adaptive --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_splg_adaptive_b${b}_eager.json.log 2>&1 )
  ( cd $R && timeout 100 python bench.py --batch $b --adaptive --graph --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_splg_adaptive_b${b}_graph.json.log 2>&1 )
done
[ $L = 1 ] || ( cd $R && timeout 200 python bench.py --workload loftr --size 480 640 > $O/bench_loftr_640x480.json.log 2>&1; tail -1 $O/bench_loftr_640x480.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && timeout 200 python bench.py --workload eloftr > $O/bench_eloftr_640x480.json.log 2>&1; tail -1 $O/bench_eloftr_640x480.json.log | cut -c1-160 )
[ $L = 1 ] || timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_eloftr -o eloftr -- python $R/bench.py --workload eloftr --steps 3 --warmup 1 > $O/rocprof_eloftr.log 2>&1
[ $L = 1 ] || ( cd $R && timeout 300 python bench.py --workload dust3r > $O/bench_dust3r_512.json.log 2>&1; tail -1 $O/bench_dust3r_512.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && timeout 300 python bench.py --workload dust3r --arith fp16 --no-cpu-baseline > $O/bench_dust3r_512_fp16.json.log 2>&1; tail -1 $O/bench_dust3r_512_fp16.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && timeout 300 python bench.py --workload mast3r --no-cpu-baseline > $O/bench_mast3r_512.json.log 2>&1; tail -1 $O/bench_mast3r_512.json.log | cut -c1-160 )
[ $L = 1 ] || timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dust3r -o dust3r -- python $R/bench.py --workload dust3r --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/rocprof_dust3r.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  [ $L = 1 ] || timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_dust3r_$c -o dust3r -- python $R/bench.py --workload dust3r --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $O/pmc_dust3r_$c.log 2>&1
done
[ $L = 1 ] || timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_dust3r_SQ -o dust3r -- python $R/bench.py --workload dust3r --steps 2 --warmup 1 --no-cpu-baseline --no-parity > $O/pmc_dust3r_SQ.log 2>&1
# the fused FFN kernel: A/B against the three-launch path, and its phase breakdown
( cd $R && IMCUI_LG_FFN_UNFUSED=1 timeout 100 python bench.py --no-cpu-baseline > $O/bench_splg_unfused_ffn.json.log 2>&1; tail -1 $O/bench_splg_unfused_ffn.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && IMCUI_LF_MATCH_4PASS=1 timeout 200 python bench.py --workload loftr > $O/bench_loftr_1024_4pass.json.log 2>&1; tail -1 $O/bench_loftr_1024_4pass.json.log | cut -c1-160 )
# ---- round 3 A/B legs: projection GEMMs on the round-2 kernel, rolled K loop, soft-max partials from the GEMM epilogue, ViT q/k/v round trip
( cd $R && IMCUI_GEMM_WREG=0 timeout 100 python bench.py --no-cpu-baseline --no-parity > $O/bench_splg_wreg_off.json.log 2>&1; tail -1 $O/bench_splg_wreg_off.json.log | cut -c1-160 )
( cd $R && IMCUI_WREG_PIPE=0 timeout 100 python bench.py --no-cpu-baseline --no-parity > $O/bench_splg_wreg_rolled.json.log 2>&1; tail -1 $O/bench_splg_wreg_rolled.json.log | cut -c1-160 )
( cd $R && IMCUI_LG_ASSIGN_STATS=epilogue timeout 100 python bench.py --no-cpu-baseline --no-parity > $O/bench_splg_assign_epilogue.json.log 2>&1; tail -1 $O/bench_splg_assign_epilogue.json.log | cut -c1-160 )
( cd $R && timeout 200 python bench.py --workload nn > $O/bench_nn.json.log 2>&1; tail -1 $O/bench_nn.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && IMCUI_DUST3R_QKV_UNFUSED=1 timeout 300 python bench.py --workload dust3r --no-cpu-baseline --no-parity > $O/bench_dust3r_512_qkv_unfused.json.log 2>&1; tail -1 $O/bench_dust3r_512_qkv_unfused.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && IMCUI_GEMM_WREG=0 timeout 300 python bench.py --workload dust3r --no-cpu-baseline --no-parity > $O/bench_dust3r_512_wreg_off.json.log 2>&1; tail -1 $O/bench_dust3r_512_wreg_off.json.log | cut -c1-160 )
# ---- later in round 3: 16-row tiles of the fused first convolution, DPT head fusions, DUSt3R at 8 pairs per step (the round-2 operating
# point), kernel table + SQ pass of the MASt3R workload (nn_argmax_* kernels)
( cd $R && timeout 300 python bench.py --h2d --no-cpu-baseline > $O/bench_splg_h2d.json.log 2>&1; tail -1 $O/bench_splg_h2d.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && IMCUI_DUST3R_REGRESS_UNFUSED=1 timeout 300 python bench.py --workload dust3r --no-cpu-baseline --no-parity > $O/bench_dust3r_512_regress_unfused.json.log 2>&1; tail -1 $O/bench_dust3r_512_regress_unfused.json.log | cut -c1-160 )
( cd $R && IMCUI_CONV_TALL=0 timeout 100 python bench.py --no-cpu-baseline --no-parity > $O/bench_splg_conv_tall_off.json.log 2>&1; tail -1 $O/bench_splg_conv_tall_off.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && IMCUI_DUST3R_HEAD_UNFUSED=1 timeout 300 python bench.py --workload dust3r --no-cpu-baseline --no-parity > $O/bench_dust3r_512_head_unfused.json.log 2>&1; tail -1 $O/bench_dust3r_512_head_unfused.json.log | cut -c1-160 )
[ $L = 1 ] || ( cd $R && timeout 300 python bench.py --workload dust3r --batch 8 --no-cpu-baseline --no-parity > $O/bench_dust3r_512_b8.json.log 2>&1; tail -1 $O/bench_dust3r_512_b8.json.log | cut -c1-160 )
[ $L = 1 ] || timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_mast3r -o mast3r -- python $R/bench.py --workload mast3r --steps 3 --warmup 1 --no-cpu-baseline --no-parity > $O/rocprof_mast3r.log 2>&1 < /dev/null
[ $L = 1 ] || timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_mast3r_SQ -o mast3r -- python $R/bench.py --workload mast3r --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --no-parity > $O/pmc_mast3r_SQ.log 2>&1 < /dev/null
( cd $R && timeout 100 python tools/ffn_bench.py > $O/lab_ffn_phases.txt 2>&1 )
( cd $R && timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 )
ls $O
