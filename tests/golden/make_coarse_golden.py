"""Golden vectors for the dual-softmax COARSE MATCHING RULE of the LoFTR family, produced by the REFERENCE's own
`dual_softmax_matcher` (imcui/hloc/matchers/dual_softmax.py:8-41; build container only, like make_golden.py).

kornia's `CoarseMatching` (behind imcui/hloc/matchers/loftr.py:54) computes conf = softmax(sim, 1) * softmax(sim, 2) with sim =
(f0 / sqrt(C)) . (f1 / sqrt(C)) / temperature and keeps the mutual maxima above the threshold: the reference's function computes
exactly that from un-normalised descriptors when it is called with `normalize=False, inv_temperature = 1 / (C * temperature)`.
The border removal is not part of the reference function (the oracle is compared with `border_rm = 0`).

    cd /tmp && python /root/repo/tests/golden/make_coarse_golden.py
"""
import os
import sys

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")

from imcui.hloc.matchers.dual_softmax import dual_softmax_matcher  # noqa: E402

CASES = {
    # name: (grid0 (h, w), grid1 (h, w), C, feature gain, threshold, temperature)
    "coarse_square": ((6, 8), (6, 8), 256, 0.6, 0.2, 0.1),
    "coarse_unequal": ((5, 9), (7, 6), 256, 0.5, 0.05, 0.1),
    "coarse_low_threshold": ((8, 8), (8, 8), 256, 0.48, 0.01, 0.1),
}

if __name__ == "__main__":
    torch.set_num_threads(1)
    for i, (name, (g0, g1, C, gain, thr, temp)) in enumerate(CASES.items()):
        g = torch.Generator().manual_seed(900 + i)
        L, S = g0[0] * g0[1], g1[0] * g1[1]
        f0 = torch.randn(1, L, C, generator=g) * gain
        perm = torch.randperm(L, generator=g)
        f1 = (f0[:, perm[torch.arange(S) % L]] + 0.9 * gain * torch.randn(1, S, C, generator=g)).contiguous()
        with torch.no_grad():
            m0, s0 = dual_softmax_matcher(f0.transpose(1, 2), f1.transpose(1, 2), threshold=thr, inv_temperature=1.0 / (C * temp), normalize=False)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), feat0=f0.numpy(), feat1=f1.numpy(), grid0=np.array(g0), grid1=np.array(g1),
                            threshold=np.float64(thr), temperature=np.float64(temp), matches0=m0.numpy(), scores0=s0.numpy())
        print(name, "matches", int((m0 > -1).sum()), "of", L)
