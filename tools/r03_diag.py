"""Round-3 diagnosis: where do replicas of one LightGlue problem differ, per layer, under each GEMM routing mode?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "image-matching-webui_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from imcui_hip.hloc.matchers.lightglue import LightGlue  # noqa: E402
from imcui_hip.synth_weights import lightglue_state_dict  # noqa: E402
from parity_utils import synthetic_matching_problem  # noqa: E402

LSD = lightglue_state_dict(0)


def batch(problems):
    B = len(problems)
    ncap = max(max(p[0].shape[0], p[1].shape[0]) for p in problems)
    k0, k1 = torch.zeros(B, ncap, 2), torch.zeros(B, ncap, 2)
    d0, d1 = torch.zeros(B, ncap, 256), torch.zeros(B, ncap, 256)
    n0, n1 = torch.zeros(B, dtype=torch.int32), torch.zeros(B, dtype=torch.int32)
    for b, (a, c, e, f) in enumerate(problems):
        k0[b, : len(a)], k1[b, : len(c)], d0[b, : len(a)], d1[b, : len(c)] = a, c, e, f
        n0[b], n1[b] = len(a), len(c)
    return [t.cuda() for t in (k0, k1, d0, d1, n0, n1)]


def run(problems):
    model = LightGlue({"depth_confidence": -1, "width_confidence": -1, "match_threshold": 0.1, "state_dict": LSD}).eval().to("cuda:0")
    out = model.forward_batched(*batch(problems), (640, 480), (640, 480), layer_dump=True)
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in out.items()}


DBGS = os.environ.get("DIAG_DBGS", "0").split(",")
for sizes in ((2048, 2048), (1500, 1300)):
    prob = [synthetic_matching_problem(5, sizes[0], sizes[1], 300)] * 8
    for mode in os.environ.get("DIAG_MODES", "0,1,2").split(","):
      os.environ["IMCUI_GEMM_WREG"] = mode
      for dbg in DBGS:
        os.environ["IMCUI_WREG_DBG"] = dbg
        for stats in ("pass",):
            if stats:
                os.environ["IMCUI_LG_ASSIGN_STATS"] = stats
            else:
                os.environ.pop("IMCUI_LG_ASSIGN_STATS", None)
            outs = [run(prob) for _ in range(2)]
            o = outs[0]
            L = o["_layers"]  # [layers, 2B, R, 256]
            msg = []
            for li in range(L.shape[0]):
                for s, cnt in enumerate(sizes):
                    ref = L[li, s, :cnt]
                    for b in range(1, 8):
                        d = (L[li, 2 * b + s, :cnt] != ref)
                        if d.any():
                            rows = d.any(1).nonzero()[:, 0]
                            cols = d.any(0).nonzero()[:, 0]
                            mx = (L[li, 2 * b + s, :cnt] - ref).abs().max().item()
                            msg.append(f"layer {li} img {s} replica {b}: {int(d.sum())} elems, {len(rows)} rows (first {rows[:6].tolist()}), {len(cols)} cols (first {cols[:6].tolist()}), max {mx:.2e}")
                if len(msg) > 6:
                    break
            rr = (outs[0]["_layers"] != outs[1]["_layers"])
            same_out = all(torch.equal(outs[0][k], outs[1][k]) for k in ("matches0", "matching_scores0"))
            rep_out = all(torch.equal(o["matching_scores0"][b], o["matching_scores0"][0]) and torch.equal(o["matches0"][b], o["matches0"][0]) for b in range(1, 8))
            print(f"sizes {sizes} WREG={mode} DBG={dbg} stats={stats}: layer-replica diffs {len(msg)}; run-to-run layer diffs {int(rr[:, :, :min(sizes)].sum())}; outputs run-to-run equal {same_out}; replicas equal {rep_out}")
            for m in msg[:6]:
                print("   ", m)
            if not rep_out:
                sc = o["matching_scores0"]
                for b in range(1, 8):
                    d = (sc[b] != sc[0])
                    if d.any():
                        print(f"    scores replica {b}: {int(d.sum())} differ, max {(sc[b] - sc[0]).abs().max().item():.2e}, idx {d.nonzero()[:5, 0].tolist()}")
                        break
