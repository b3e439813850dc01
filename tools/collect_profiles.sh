#!/bin/bash
# Round evidence collector: run on the GPU box (gpurun), writes under gpurun_out/final_r06/; tools/summarize_profiles.py r06_final then
# copies the judged summaries into profiles/ (each stamped with the commit in <out>/HEAD).  rocprofv3 needs cwd=/tmp and TMPDIR=/tmp on
# this pool; counters are collected one per pass (FETCH_SIZE, WRITE_SIZE, one SQ group), never together with a trace domain.
#   HEAD=<commit>  the commit of the snapshot (no .git on the box: pass `HEAD=$(git rev-parse --short HEAD)` on the gpurun command line)
#   PART=1: the driver's line (+ its stderr detail), the stand-alone bench lines the summaries are normalised by, kernel tables
#   PART=2: counter passes (FETCH_SIZE / WRITE_SIZE / SQ) of splg, loftr, eloftr, dust3r, nn -- each at its leg's batch size
#   PART=3: A/B legs, labs, the full GPU test log, smoke          (default: 123)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${OUT:-final2_r06}
mkdir -p $O
echo "${HEAD:-unknown}" > $O/HEAD
P=${PART:-123}
cd /tmp && export TMPDIR=/tmp
b() { ( cd $R && timeout ${T:-300} python bench.py "${@:2}" > $O/$1.json.log 2>$O/$1.err; tail -1 $O/$1.json.log | cut -c1-150 ); }
stats() { timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$1 -o $1 -- python $R/bench.py "${@:2}" --no-cpu-baseline --no-parity --no-legs > $O/rocprof_$1.log 2>&1 < /dev/null; echo "stats $1 rc $?"; }
pmc() { timeout 300 rocprofv3 --kernel-trace --pmc ${@:3} --output-format csv -d $O/pmc_$1 -o $2 -- python $R/bench.py ${WL} --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-legs > $O/pmc_$1.log 2>&1 < /dev/null; echo "pmc $1 rc $?"; }
SQ="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"
if [[ $P == *1* ]]; then
  T=1500 b bench_splg --steps 20 --warmup 5                      # the driver's command: headline + every leg (compact line on stdout, full record on stderr)
  b bench_nn --workload nn --no-legs --no-cpu-baseline            # (the stand-alone lines the traffic summaries are normalised by)
  b bench_loftr_1024 --workload loftr --no-legs --no-cpu-baseline
  b bench_eloftr_640x480 --workload eloftr --no-legs --no-cpu-baseline
  b bench_dust3r_512 --workload dust3r --no-legs --no-cpu-baseline
  b bench_loftr_640x480 --workload loftr --size 480 640 --no-legs --no-cpu-baseline
  b bench_loftr_1024_fine_dense --workload loftr --fine-dense --no-legs --no-cpu-baseline   # round 6: the last FPN stage as dense maps (option loftr_fine_sparse = 0)
  stats splg --steps 5 --warmup 2
  stats nn --workload nn --steps 5 --warmup 2
  stats loftr --workload loftr --steps 5 --warmup 2
  stats eloftr --workload eloftr --steps 3 --warmup 1
  stats dust3r --workload dust3r --steps 3 --warmup 1
  stats splg_b1 --batch 1 --steps 30 --warmup 3
fi
if [[ $P == *2* ]]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    WL="--no-legs" pmc $c splg $c
    WL="--workload nn" pmc nn_$c nn $c
    WL="--workload loftr" pmc loftr_$c loftr $c
    WL="--workload eloftr" pmc eloftr_$c eloftr $c
    WL="--workload dust3r" pmc dust3r_$c dust3r $c
  done
  WL="--no-legs" pmc SQ splg $SQ
  WL="--workload nn" pmc nn_SQ nn $SQ
  WL="--workload loftr" pmc loftr_SQ loftr $SQ
  WL="--workload eloftr" pmc eloftr_SQ eloftr $SQ
  WL="--workload dust3r" pmc dust3r_SQ dust3r $SQ
fi
if [[ $P == *3* ]]; then
  IMCUI_ATTN_VARIANT_CROSS=-1 b bench_splg_attn_cross_off --no-cpu-baseline --no-parity --no-legs   # three products everywhere (the round-4 arithmetic)
  IMCUI_ATTN_VARIANT=9 b bench_splg_attn_v9 --no-cpu-baseline --no-legs                              # round 6: P.V corrections on fp6 MFMA in every block (opt-in)
  IMCUI_ATTN_SPLIT=0 b bench_splg_b1_nosplit --batch 1 --steps 30 --warmup 3 --no-cpu-baseline --no-legs --no-parity   # one pair per step without the key-split launch
  b bench_eloftr_640x480_b8 --workload eloftr --batch 8 --no-legs --no-cpu-baseline                   # (round 5's batch)
  ( cd $R && timeout 120 tools/mx_lab > $O/lab_mx_mfma.txt 2>&1; tail -8 $O/lab_mx_mfma.txt )
  ( cd $R && AUDIT_REDUCED=9 timeout 900 python tools/attn_mix_audit.py > $O/lab_attention_mx_mix.txt 2>/dev/null; tail -9 $O/lab_attention_mx_mix.txt | cut -c1-200 )
  IMCUI_SIMRED=0 b bench_nn_simred_off --workload nn --no-legs --no-cpu-baseline                     # the round-4 tile GEMM with the reducing epilogue
  b bench_splg_h2d --h2d raw --no-cpu-baseline --no-legs
  b bench_splg_h2d_jpeg --h2d jpeg --no-cpu-baseline --no-legs
  b bench_splg_b1 --batch 1 --steps 30 --warmup 3 --no-cpu-baseline --no-legs
  b bench_splg_b32 --batch 32 --no-cpu-baseline --no-legs
  b bench_splg_f32 --precision 0 --no-cpu-baseline --no-legs
  b bench_splg_adaptive_b1_graph --batch 1 --adaptive --graph --steps 40 --warmup 5 --no-cpu-baseline --no-legs
  ( cd $R && timeout 900 python tools/attn_mix_audit.py > $O/lab_attention_mix.txt 2>/dev/null; tail -9 $O/lab_attention_mix.txt | cut -c1-200 )
  ( cd $R && for dbg in 0 1 3; do echo "== IMCUI_SR_DBG=$dbg (bit 0: no per-column part, bit 1: no per-row part; WRONG results, timing only)"; for wl in nn loftr; do IMCUI_SR_DBG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_sr_${wl}_$dbg -o sr -- python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-legs > /dev/null 2>&1 < /dev/null; python3 tools/top_kernels.py $O/stats_sr_${wl}_$dbg 30 | grep simred_kernel; done; done ) > $O/lab_simred_parts.txt 2>&1
  ( cd $R && timeout 300 python tools/jpeg_bench.py > $O/lab_jpeg.txt 2>/dev/null; tail -4 $O/lab_jpeg.txt )
  ( cd $R && timeout 600 python tools/loftr_fine_lab.py > $O/lab_loftr_fine.txt 2>/dev/null; tail -10 $O/lab_loftr_fine.txt )   # round 6: window vs dense evaluation of LoFTR's last FPN stage by match count
  ( cd $R && timeout 200 python tools/ffn_bench.py 262144 20 > $O/lab_ffn_phases.txt 2>/dev/null; tail -8 $O/lab_ffn_phases.txt )
  ( cd $R && timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -s > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log )
  ( cd $R && timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2 )
fi
ls $O | wc -l
