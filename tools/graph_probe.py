#!/usr/bin/env python
"""Probe: HIP-graph replay of the SuperPoint+LightGlue step vs eager launches (equality + latency)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-matching-webui_amd"))
from imcui_hip.pipeline import GraphedPipeline, SuperPointLightGluePipeline  # noqa: E402
from imcui_hip.synth import make_pair_batch  # noqa: E402
from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict  # noqa: E402  (seeded weights only)

dev = torch.device("cuda:0")
pipe = SuperPointLightGluePipeline(
    {"nms_radius": 3, "max_keypoints": 2048, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)},
    {"depth_confidence": 0.95, "width_confidence": 0.99, "match_threshold": 0.1, "state_dict": lightglue_state_dict(0)},
).eval().to(dev)
for B in (1, 4):
    img0, img1, _ = make_pair_batch(7, B, 480, 640, distinct=B)
    img0, img1 = img0.to(dev), img1.to(dev)
    eager = {k: v.clone() for k, v in pipe(img0, img1).items()}
    g = GraphedPipeline(pipe, img0, img1)
    other0, other1, _ = make_pair_batch(8, B, 480, 640, distinct=B)
    g(other0.to(dev), other1.to(dev))  # different inputs in between
    rep = g(img0, img1)
    torch.cuda.synchronize()
    same = all(torch.equal(eager[k], rep[k]) for k in ("matches0", "matching_scores0", "stop", "num_keypoints0", "keypoints0"))
    n = 50
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        pipe(img0, img1)
    torch.cuda.synchronize()
    t_e = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n):
        g(img0, img1)
    torch.cuda.synchronize()
    t_g = (time.perf_counter() - t0) / n * 1e3
    print(f"B={B} identical={same} eager {t_e:.3f} ms/step  graph {t_g:.3f} ms/step  matches {int((rep['matches0'] > -1).sum())}", flush=True)
