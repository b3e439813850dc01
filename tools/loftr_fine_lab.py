"""Lab (round 6): where the window evaluation of LoFTR's last FPN stage stops paying.  1024 x 1024 pairs, the match count driven by the coarse
threshold (synthetic weights: the dual soft-max is diffuse, so a lower threshold admits more mutual matches); per threshold the step time with
option loftr_fine_sparse = 0 (dense maps, no read-back of the match count: capacity-sized fine-level grids), 2 (always on the windows) and 1
(the cost model of csrc/loftr.hip picks; with the count read back the fine level launches for the matches that exist, whichever way the stage runs).
    python tools/loftr_fine_lab.py > profiles/r06_lab_loftr_fine.txt"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-matching-webui_amd")]
from imcui_hip import backend  # noqa: E402
from imcui_hip.hloc.matchers.loftr import LoFTR  # noqa: E402
from imcui_hip.synth import make_pair  # noqa: E402
from imcui_hip.synth_weights import loftr_state_dict  # noqa: E402

dev = torch.device("cuda:0")
B, H, W = int(os.environ.get("LAB_B", "8")), 1024, 1024
base, _, _ = make_pair(77, H + 16, W + 16, n_blobs=H * W // 150)
img0 = base[..., 0:H, 0:W].contiguous().repeat(B, 1, 1, 1).to(dev)
img1 = base[..., 8 : H + 8, 16 : W + 16].contiguous().repeat(B, 1, 1, 1).to(dev)
sd = loftr_state_dict(0)
print(f"# LoFTR {W}x{H}, {B} pairs per step, 3 x f16 split arithmetic; ms per step (HIP events over 5 steps after 2 warm-ups)")
print(f"# {'threshold':>10s} {'matches/pair':>12s} {'dense':>9s} {'windows':>9s} {'auto':>9s}  auto took")
for thr in (0.2, 0.05, 0.01, 0.003, 0.001, 0.0003, 0.0001, 0.00003, 0.00001, 0.000005, 0.000002):
    model = LoFTR({"match_threshold": thr, "max_keypoints": None, "state_dict": sd}).eval().to(dev)
    row, nm, took = [], 0, ""
    for opt in (0, 2, 1):
        with backend.option(dev, loftr_fine_sparse=opt):
            for _ in range(2):
                out = model.forward_batched(img0, img1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = model.forward_batched(img0, img1)
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 5)
            nm = int(out["num_matches"][0])
            if opt == 1:
                took = "windows" if model._impl.last_fine_mode(dev)[0] == 1 else "dense"
    print(f"  {thr:10.5f} {nm / B:12.1f} {row[0]:9.2f} {row[1]:9.2f} {row[2]:9.2f}  {took}", flush=True)
    del model
