#!/usr/bin/env python
"""Turns gpurun_out/final/ (written by tools/collect_profiles.sh on the GPU box) into the tracked evidence
under profiles/: bench logs, rocprofv3 kernel stats, per-kernel HBM traffic from the PMC passes (FETCH_SIZE is
doubled per MI355X_MICROARCH.md), and the attention traffic record that bench.py reports as roofline.traffic."""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", os.environ.get("OUT", "final2_r06"))
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"
# every summary carries the commit the collection ran at (tools/collect_profiles.sh writes it into <out>/HEAD: VERDICT round 4, item 2 iv)
HEAD = open(os.path.join(F, "HEAD")).read().strip() if os.path.exists(os.path.join(F, "HEAD")) else "unknown"
_dump = json.dump


def _stamped(obj, fh, **kw):
    if isinstance(obj, dict):
        obj = {"head": HEAD, **obj}
    return _dump(obj, fh, **kw)


json.dump = _stamped


def agg(counter, sub=None, stem="splg", name=None):
    """Per-kernel mean of one counter over the launches of a PMC pass (directory pmc_<sub>, counter column `name`)."""
    path = os.path.join(F, f"pmc_{sub or counter}", f"{stem}_counter_collection.csv")
    rows = list(csv.DictReader(open(path)))
    by = collections.defaultdict(list)
    for r in rows:
        if name is not None and r.get("Counter_Name") != name:
            continue
        by[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in by.items()}


bench = json.loads(open(os.path.join(F, "bench_splg.json.log")).read().strip().split("\n")[-1])
B = bench["config"]["pairs_per_step_per_gpu"]
fe, wr = agg("FETCH_SIZE"), agg("WRITE_SIZE")
out = {}
for k in fe:
    if fe[k][1] >= 2:
        out[k] = {"launches": fe[k][1], "fetch_bytes_per_launch": fe[k][0] * 1024 * 2, "write_bytes_per_launch": wr.get(k, (0, 0))[0] * 1024}
json.dump({"note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 1, B={B} pairs; per-launch "
                   "averages in bytes; FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md", "batch_pairs": B, "kernels": out},
          open(os.path.join(P, f"{tag}_pmc_traffic.json"), "w"), indent=1)
akey = next(k for k in out if k.startswith("attn_split_kernel"))
a = out[akey]
sys.path.insert(0, ROOT)
from bench import sources_sha16  # noqa: E402  (hashes of the attention kernel's sources: bench.py flags a later change as `traffic_stale`)

json.dump({"kernel": akey, "source": f"profiles/{tag}_pmc_traffic.json", "batch_pairs": B, "sources_sha16": sources_sha16(),
           "fetch_bytes_per_launch": a["fetch_bytes_per_launch"], "write_bytes_per_launch": a["write_bytes_per_launch"],
           "traffic_bytes_per_launch": a["fetch_bytes_per_launch"] + a["write_bytes_per_launch"],
           "algorithmic_bytes_per_launch": 16777216 * B,
           "note": "Q, K, V^T planes read once + O written once per launch; the XCD-aware grid keeps the K/V of a (sequence, head) in one L2 "
                   "(traffic == compulsory bytes; it was 4.5x that before the remap)"},
          open(os.path.join(P, f"{tag}_attention_traffic.json"), "w"), indent=1)
# LoFTR / EfficientLoFTR: HBM traffic of the GEMM-class kernels per step (same two passes on the dense workloads)
for stem, benchlog in (("loftr", "bench_loftr_1024.json.log"), ("eloftr", "bench_eloftr_640x480.json.log"), ("dust3r", "bench_dust3r_512.json.log"), ("nn", "bench_nn.json.log")):
    if not os.path.exists(os.path.join(F, f"pmc_{stem}_FETCH_SIZE", f"{stem}_counter_collection.csv")):
        continue
    lfe, lwr = agg("FETCH_SIZE", f"{stem}_FETCH_SIZE", stem), agg("WRITE_SIZE", f"{stem}_WRITE_SIZE", stem)
    lb = json.loads(open(os.path.join(F, benchlog)).read().strip().split("\n")[-1])
    lout = {k: {"launches": v[1], "fetch_bytes_per_launch": v[0] * 1024 * 2, "write_bytes_per_launch": lwr.get(k, (0, 0))[0] * 1024}
            for k, v in lfe.items() if v[1] >= 2}
    steps = 3  # bench.py --steps 2 --warmup 1
    tot = sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] for v in lout.values()) / steps
    gem = sum((v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) * v["launches"] for k, v in lout.items()
              if k.startswith("gemm_") or k.startswith("lg_ffn") or k.startswith("conv3x3_") or (stem == "dust3r" and k.startswith("attn_"))) / steps
    json.dump({"note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --workload {stem} --steps 2 --warmup 1; FETCH doubled; "
                       "matrix class = conv3x3_*, gemm_* and the fused MLP kernel, for dust3r also attn_* (what the bench's HIP-event classes time)",
               "pairs_per_step": lb["config"]["pairs_per_step_per_gpu"], "traffic_bytes_per_step_all_kernels": tot,
               "traffic_bytes_per_step_gemm_kernels": gem, "kernels": lout}, open(os.path.join(P, f"{tag}_pmc_traffic_{stem}.json"), "w"), indent=1)
# matrix-pipe occupancy / stall breakdown from the SQ pass
sqp = os.path.join(F, "pmc_SQ", "splg_counter_collection.csv")
if os.path.exists(sqp):
    names = ["SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT"]
    cnt = {n: agg(None, "SQ", "splg", n) for n in names}
    sq = {}
    for k, (busy, n) in cnt["SQ_BUSY_CYCLES"].items():
        g = lambda c: cnt[c].get(k, (0.0, 0))[0]  # noqa: E731
        kc = busy / 32.0  # SQ_BUSY_CYCLES is summed over the 32 shader engines
        wc = max(g("SQ_WAVE_CYCLES"), 1.0)
        sq[k] = {"launches": n, "kernel_cycles": kc, "mfma_busy_cycles_per_simd": g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0,
                 "mfma_busy_frac": g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / max(kc, 1.0), "valu_active_frac_of_wave_cycles": g("SQ_ACTIVE_INST_VALU") / wc,
                 "wait_any_frac": g("SQ_WAIT_ANY") / wc, "wait_inst_frac": g("SQ_WAIT_INST_ANY") / wc, "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT")}
    json.dump({"note": f"rocprofv3 --pmc SQ_* (one pass), bench.py --steps 2 --warmup 1, B={B}; kernel_cycles = SQ_BUSY_CYCLES / 32 shader engines, "
                       "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs; per-launch averages", "kernels": sq},
              open(os.path.join(P, f"{tag}_pmc_sq.json"), "w"), indent=1)
    for k, v in sorted(sq.items(), key=lambda kv: -kv[1]["kernel_cycles"] * kv[1]["launches"])[:6]:
        print(f"{k[:50]:50s} mfma_busy {v['mfma_busy_frac']:.3f} wait_any {v['wait_any_frac']:.2f} wait_inst {v['wait_inst_frac']:.2f}")
# the same SQ pass on the DUSt3R workload; its GEMM launches are grouped by grid size (= shape class: encoder / decoder, N, merged sides)
for wl, wl_note in (("loftr", "bench.py --workload loftr --steps 2 --warmup 1 (1024x1024, 4 pairs per step)"),
                    ("eloftr", "bench.py --workload eloftr --steps 2 --warmup 1 (640x480, 32 pairs per step)"),
                    ("dust3r", "bench.py --workload dust3r --steps 2 --warmup 1 (512x512, 16 pairs per step, 3 x f16 split arithmetic)"),
                    ("mast3r", "bench.py --workload mast3r --batch 2 --steps 1 --warmup 1 (512x512: network + reciprocal matching, nn_argmax_* kernels)"),
                    ("nn", "bench.py --workload nn --steps 2 --warmup 1 (mutual-NN matcher, 5000 x 128-d descriptors, 64 pairs per step)")):
    dsq = os.path.join(F, f"pmc_{wl}_SQ", f"{wl}_counter_collection.csv")
    if not os.path.exists(dsq):
        continue
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(dsq)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if k.startswith("gemm_split_kernel") or k.startswith("gemm_wreg_kernel") or (wl in ("loftr", "eloftr") and k.startswith("conv3x3_")):
            k = f"{k} grid {r['Grid_Size']}"
        by[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dq = {}
    for k, c in by.items():
        g = lambda n: sum(c.get(n, [0.0])) / max(len(c.get(n, [0.0])), 1)  # noqa: E731
        kc = g("SQ_BUSY_CYCLES") / 32.0
        wc = max(g("SQ_WAVE_CYCLES"), 1.0)
        dq[k] = {"launches": len(c["SQ_BUSY_CYCLES"]), "kernel_cycles": kc, "mfma_busy_frac": g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / max(kc, 1.0),
                 "valu_active_frac_of_wave_cycles": g("SQ_ACTIVE_INST_VALU") / wc, "wait_any_frac": g("SQ_WAIT_ANY") / wc,
                 "wait_inst_frac": g("SQ_WAIT_INST_ANY") / wc, "lds_bank_conflict_cycles": g("SQ_LDS_BANK_CONFLICT")}
    json.dump({"note": "rocprofv3 --pmc SQ_* (one pass), " + wl_note + "; kernel_cycles = SQ_BUSY_CYCLES / 32 shader engines, mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs; per-launch "
                       "averages; GEMM launches grouped by grid size (= shape class)", "kernels": dq},
              open(os.path.join(P, f"{tag}_pmc_sq_{wl}.json"), "w"), indent=1)
for opt in ("adaptive", "b1", "adaptive_b1_eager", "adaptive_b1_graph", "adaptive_b4_eager", "adaptive_b4_graph"):  # operating points beside the headline line
    if os.path.exists(os.path.join(F, f"bench_splg_{opt}.json.log")):
        shutil.copy(os.path.join(F, f"bench_splg_{opt}.json.log"), os.path.join(P, f"{tag}_bench_splg_{opt}.json.log"))
for src, dst in [("bench_superglue.json.log", f"{tag}_bench_superglue.json.log"),
                 ("stats_superglue/superglue_kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats_superglue.csv")]:
    if os.path.exists(os.path.join(F, src)):
        shutil.copy(os.path.join(F, src), os.path.join(P, dst))
for src, dst in [("bench_splg.json.log", f"{tag}_bench_splg.json.log"), ("bench_loftr_1024.json.log", f"{tag}_bench_loftr_1024.json.log"),
                 ("bench_splg_f32.json.log", f"{tag}_bench_splg_f32.json.log"),
                 ("bench_superpoint.json.log", f"{tag}_bench_superpoint.json.log"),
                 ("stats_splg/splg_kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats_splg.csv"),
                 ("stats_loftr/loftr_kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats_loftr_1024.csv"),
                 ("bench_loftr_640x480.json.log", f"{tag}_bench_loftr_640x480.json.log"), ("bench_splg_b64.json.log", f"{tag}_bench_splg_b64.json.log"), ("bench_splg_b32.json.log", f"{tag}_bench_splg_b32.json.log"),
                 ("bench_loftr_1024_4pass.json.log", f"{tag}_bench_loftr_1024_4pass.json.log"),
                 ("bench_eloftr_640x480.json.log", f"{tag}_bench_eloftr_640x480.json.log"),
                 ("stats_eloftr/eloftr_kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats_eloftr_640x480.csv"),
                 ("bench_dust3r_512.json.log", f"{tag}_bench_dust3r_512.json.log"),
                 ("bench_dust3r_512_fp16.json.log", f"{tag}_bench_dust3r_512_fp16.json.log"),
                 ("bench_mast3r_512.json.log", f"{tag}_bench_mast3r_512.json.log"),
                 ("stats_dust3r/dust3r_kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats_dust3r_512.csv"),
                 ("bench_splg_unfused_ffn.json.log", f"{tag}_bench_splg_unfused_ffn.json.log"),
                 ("bench_splg_wreg_off.json.log", f"{tag}_bench_splg_wreg_off.json.log"), ("bench_splg_wreg_rolled.json.log", f"{tag}_bench_splg_wreg_rolled.json.log"),
                 ("bench_splg_assign_epilogue.json.log", f"{tag}_bench_splg_assign_epilogue.json.log"), ("bench_nn.json.log", f"{tag}_bench_nn.json.log"),
                 ("bench_dust3r_512_qkv_unfused.json.log", f"{tag}_bench_dust3r_512_qkv_unfused.json.log"),
                 ("bench_dust3r_512_wreg_off.json.log", f"{tag}_bench_dust3r_512_wreg_off.json.log"),
                 ("bench_splg_conv_tall_off.json.log", f"{tag}_bench_splg_conv_tall_off.json.log"),
                 ("bench_splg_h2d.json.log", f"{tag}_bench_splg_h2d.json.log"),
                 ("bench_dust3r_512_regress_unfused.json.log", f"{tag}_bench_dust3r_512_regress_unfused.json.log"),
                 ("bench_dust3r_512_head_unfused.json.log", f"{tag}_bench_dust3r_512_head_unfused.json.log"),
                 ("bench_dust3r_512_b8.json.log", f"{tag}_bench_dust3r_512_b8.json.log"),
                 ("stats_mast3r/mast3r_kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats_mast3r_512.csv"),
                 ("stats_nn/nn_kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats_nn.csv"),
                 ("bench_splg_attn_v0.json.log", f"{tag}_bench_splg_attn_v0.json.log"), ("bench_splg_attn_v6.json.log", f"{tag}_bench_splg_attn_v6.json.log"),
                 ("bench_splg_attn_v7.json.log", f"{tag}_bench_splg_attn_v7.json.log"), ("bench_splg_h2d_jpeg.json.log", f"{tag}_bench_splg_h2d_jpeg.json.log"),
                 ("lab_attention_pv2.txt", f"{tag.split('_')[0]}_lab_attention_pv2.txt"), ("lab_jpeg.txt", f"{tag.split('_')[0]}_lab_jpeg.txt"),
                 ("pytest_gpu.log", f"{tag}_pytest_gpu.log"), ("bench_splg.err", f"{tag}_bench_splg.detail.log"),
                 ("lab_attention_mix.txt", f"{tag.split('_')[0]}_lab_attention_mix.txt"), ("bench_splg_attn_cross_off.json.log", f"{tag}_bench_splg_attn_cross_off.json.log"),
                 ("bench_nn_simred_off.json.log", f"{tag}_bench_nn_simred_off.json.log"), ("lab_simred_parts.txt", f"{tag.split('_')[0]}_lab_simred_parts.txt"),
                 ("bench_splg_adaptive_b1_graph.json.log", f"{tag}_bench_splg_adaptive_b1_graph.json.log"),
                 ("lab_ffn_phases.txt", f"{tag}_lab_ffn_phases.txt"),
                 ("bench_splg_attn_v9.json.log", f"{tag}_bench_splg_attn_v9.json.log"), ("bench_splg_b1_nosplit.json.log", f"{tag}_bench_splg_b1_nosplit.json.log"),
                 ("bench_eloftr_640x480_b8.json.log", f"{tag}_bench_eloftr_640x480_b8.json.log"), ("lab_mx_mfma.txt", f"{tag.split('_')[0]}_lab_mx_mfma.txt"),
                 ("lab_attention_mx_mix.txt", f"{tag.split('_')[0]}_lab_attention_mx_mix.txt"),
                 ("bench_loftr_1024_fine_dense.json.log", f"{tag}_bench_loftr_1024_fine_dense.json.log"), ("lab_loftr_fine.txt", f"{tag}_lab_loftr_fine.txt"),
                 ("stats_splg_b1/splg_b1_kernel_stats.csv", f"{tag}_rocprofv3_kernel_stats_splg_b1.csv")]:
    if not os.path.exists(os.path.join(F, src)):
        continue
    shutil.copy(os.path.join(F, src), os.path.join(P, dst))
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["fetch_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(f"{k[:50]:50s} n={v['launches']:4d} fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB write {v['write_bytes_per_launch'] / 1e6:9.1f} MB")
print("pairs/s", bench["value"], "batch", B)
