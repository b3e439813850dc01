"""CPU probe for VERDICT round 4, item 4 (second half): can LightGlue's cross block share ONE exp(s - c) per similarity between its two
soft-max directions, with c constant per (128-query block, 64-key tile)?

The attention kernel keeps P in f16 hi / lo planes for the second product (attention.hip): a row's probabilities are exp(s - m_row), its
largest is 1 and everything below 2^-24 of it is dropped, harmlessly.  With a tile constant c >= max(tile) instead of the row's own
maximum, row i's largest stored value is exp(m_i - c): the row keeps 24 + log2(exp(m_i - c)) bits of range, and none once m_i - c < -16.6
(f16 flushes) -- or -87 in f32 arithmetic (exp under-flows).  This script runs the oracle's nine layers on a synthetic pair for the three
weight sets of the attention audits and reports, per cross block and direction, how far the row maxima of a block lie below (a) the exact
maximum of their (block, tile) and (b) the cheap norm bound max|q| max|k| the proposal names -- the quantity m_i - c above.

    python tools/shared_s_probe.py > profiles/r05_lab_shared_s_probe.txt        (CPU only, about a minute)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-matching-webui_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from imcui_hip.synth_weights import lightglue_state_dict  # noqa: E402
from oracle.lightglue import LightGlueOracle, normalize_keypoints  # noqa: E402
from parity_utils import synthetic_matching_problem  # noqa: E402

N = 1024
WEIGHTS = {"damped": lightglue_state_dict(0), "strong": lightglue_state_dict(0, damp=0.1, ln_noise=0.1, final_gain=10.0),
           "random": lightglue_state_dict(1, structured=False)}  # fmt: skip


def probe(sim, qn, kn, rows=128, cols=64):
    """sim [h, n, m]; qn [h, n], kn [h, m] vector norms.  Returns the worst (most negative) m_i - c over all rows and tiles whose row
    actually has its maximum in that tile or not -- the row is rescaled by ONE constant per tile, so every tile counts -- for c = the
    exact tile maximum and for c = the norm bound, and the fraction of (row, tile) pairs beyond the f16 range (-16.6)."""
    h, n, m = sim.shape
    rowmax = sim.max(2).values  # [h, n]
    worst_exact, worst_bound, lost_exact, lost_bound, cnt = 0.0, 0.0, 0, 0, 0
    for r0 in range(0, n, rows):
        for c0 in range(0, m, cols):
            t = sim[:, r0 : r0 + rows, c0 : c0 + cols]
            tmax = t.amax((1, 2))  # [h]
            bound = qn[:, r0 : r0 + rows].amax(1) * kn[:, c0 : c0 + cols].amax(1)
            # the row's own best inside this tile against the tile's constant: what its stored probabilities of this tile top out at
            rbest = t.amax(2)  # [h, rows]
            # only rows for which this tile matters (its best here is within 16.6 of the row's overall maximum: it carries visible mass)
            live = rbest >= rowmax[:, r0 : r0 + rows] - 16.6
            de, db = rbest - tmax[:, None], rbest - bound[:, None]
            worst_exact = min(worst_exact, de[live].min().item())
            worst_bound = min(worst_bound, db[live].min().item())
            lost_exact += int((de[live] < -16.6).sum())
            lost_bound += int((db[live] < -16.6).sum())
            cnt += int(live.sum())
    return worst_exact, worst_bound, lost_exact / max(cnt, 1), lost_bound / max(cnt, 1)


def main():
    torch.set_num_threads(16)
    k0, k1, d0, d1 = synthetic_matching_problem(7, N, N, N // 8)
    print(f"# shared-S probe: {N} x {N} key-points, 4 heads x 64, tiles of 128 queries x 64 keys; m_i - c = the row's best in a tile minus the tile's constant")
    print("# (f16 P planes flush a row's tile to zero below -16.6; 'lost' = fraction of (row, tile) pairs that carry visible mass and would be flushed)")
    for name, sd in WEIGHTS.items():
        ora = LightGlueOracle(sd, {"depth_confidence": -1, "width_confidence": -1})
        x0, x1 = d0[None].clone(), d1[None].clone()
        e0 = ora.posenc(normalize_keypoints(k0[None], (640, 480)))
        e1 = ora.posenc(normalize_keypoints(k1[None], (640, 480)))
        for i in range(9):
            x0, x1 = ora.self_block(i, x0, e0), ora.self_block(i, x1, e1)
            p = f"transformers.{i}.cross_attn"
            q0 = ora._lin(x0, p + ".to_qk").unflatten(-1, (4, -1)).transpose(1, 2)[0] * 64 ** -0.25
            q1 = ora._lin(x1, p + ".to_qk").unflatten(-1, (4, -1)).transpose(1, 2)[0] * 64 ** -0.25
            sim = torch.einsum("hid,hjd->hij", q0, q1)
            n0, n1 = q0.norm(dim=-1), q1.norm(dim=-1)
            a = probe(sim, n0, n1)
            b = probe(sim.transpose(1, 2).contiguous(), n1, n0)
            print(f"{name:7s} layer {i}: |sim| max {sim.abs().max().item():7.2f}   0->1: exact-tile-max {a[0]:8.2f} (lost {a[2]:.4f})  norm bound {a[1]:8.2f} (lost {a[3]:.4f})"
                  f"   1->0: exact {b[0]:8.2f} (lost {b[2]:.4f})  bound {b[1]:8.2f} (lost {b[3]:.4f})")
            x0, x1 = ora.cross_block(i, x0, x1)


if __name__ == "__main__":
    main()
