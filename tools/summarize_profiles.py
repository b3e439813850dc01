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
F = os.path.join(ROOT, "gpurun_out", "final")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01_final"


def agg(counter):
    rows = list(csv.DictReader(open(os.path.join(F, f"pmc_{counter}", "splg_counter_collection.csv"))))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
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
a = out["attn_split_kernel"]
json.dump({"kernel": "attn_split_kernel", "source": f"profiles/{tag}_pmc_traffic.json", "batch_pairs": B,
           "fetch_bytes_per_launch": a["fetch_bytes_per_launch"], "write_bytes_per_launch": a["write_bytes_per_launch"],
           "traffic_bytes_per_launch": a["fetch_bytes_per_launch"] + a["write_bytes_per_launch"],
           "algorithmic_bytes_per_launch": 16777216 * B,
           "note": "Q, K, V^T planes read once + O written once per launch; the XCD-aware grid keeps the K/V of a (sequence, head) in one L2 "
                   "(traffic == compulsory bytes; it was 4.5x that before the remap)"},
          open(os.path.join(P, "r01_attention_traffic.json"), "w"), indent=1)
for opt in ("adaptive", "b1"):  # operating points beside the headline line (collected when present)
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
                 ("lab_clock.txt", "r01_lab_mfma_clock.txt"), ("lab_overlap.txt", "r01_lab_mfma_valu_overlap.txt"),
                 ("lab_launch.txt", "r01_lab_workgroup_launch.txt"), ("lab_gridsync.txt", "r01_lab_gridsync_barrier.txt")]:
    if not os.path.exists(os.path.join(F, src)):
        continue
    shutil.copy(os.path.join(F, src), os.path.join(P, dst))
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["fetch_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(f"{k[:50]:50s} n={v['launches']:4d} fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB write {v['write_bytes_per_launch'] / 1e6:9.1f} MB")
print("pairs/s", bench["value"], "batch", B)
