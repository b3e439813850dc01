"""Experiment: does running two half-batches of the headline step on two HIP streams (tails of one stream's kernels filled by the other's
work-groups) beat one full batch on one stream?    python tools/two_stream_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-matching-webui_amd")]
import torch  # noqa: E402

from imcui_hip.pipeline import SuperPointLightGluePipeline  # noqa: E402
from imcui_hip.synth import make_pair_batch  # noqa: E402
from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict  # noqa: E402

dev = torch.device("cuda:0")
spc = {"nms_radius": 3, "max_keypoints": 2048, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)}
lgc = {"depth_confidence": -1.0, "width_confidence": -1.0, "match_threshold": 0.1, "state_dict": lightglue_state_dict(0)}


def run(B, nstream, steps=15, warm=4):
    pipes = [SuperPointLightGluePipeline(dict(spc), dict(lgc)).eval().to(dev) for _ in range(nstream)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstream)]
    img0, img1, _ = make_pair_batch(1234, B, 480, 640, distinct=8)
    img0, img1 = img0.to(dev), img1.to(dev)
    per = B // nstream
    parts = [(img0[i * per : (i + 1) * per].contiguous(), img1[i * per : (i + 1) * per].contiguous()) for i in range(nstream)]

    def step():
        for p, s, (a, b) in zip(pipes, streams, parts):
            with torch.cuda.stream(s):
                p(a, b)

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"B = {B:3d} pairs per step on {nstream} stream(s) ({per} per stream): {dt * 1e3:7.2f} ms / step = {B / dt:7.1f} pairs/s", flush=True)
    del pipes


for B, ns in ((64, 1), (64, 2), (128, 1), (128, 2), (128, 4), (96, 3), (64, 1)):
    run(B, ns)
