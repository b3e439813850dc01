"""LoFTR oracle self-consistency on CPU (parity unpinned: no reference implementation is available).

Property checked: two crops of one image offset by a multiple of the coarse stride must match
cell-to-cell with the known displacement (a wrong unfold / transposition / image swap breaks it).
"""
import torch

from imcui_hip.synth import make_pair
from oracle.loftr import LoFTROracle
from imcui_hip.synth_weights import loftr_state_dict


def test_loftr_oracle_recovers_known_translation():
    torch.set_num_threads(4)
    h, w = 160, 224
    base, _, _ = make_pair(5, h + 16, w + 16, n_blobs=500)
    img0, img1 = base[..., 0:h, 0:w].contiguous(), base[..., 8 : h + 8, 16 : w + 16].contiguous()
    sd = loftr_state_dict(0)
    assert abs(sum(v.numel() for k, v in sd.items() if "num_batches" not in k) - 11.57e6) < 0.05e6  # LoFTR-outdoor size
    out = LoFTROracle(sd, {"match_threshold": 0.01, "max_keypoints": 2000})({"image0": img0, "image1": img1})
    assert len(out["scores"]) > 30
    # img1(y, x) = img0(y + 8, x + 16): a point at p0 in image0 is at p0 - (16, 8) in image1
    err = (out["keypoints0"] - torch.tensor([16.0, 8.0]) - out["keypoints1"]).norm(dim=1)
    assert err.median().item() < 3.0 and (err < 8).float().mean().item() > 0.95
    # the wrapper refines image0: image1's key-points stay on the coarse 8-px grid
    assert (out["keypoints1"] % 8 == 0).all()
    top = LoFTROracle(sd, {"match_threshold": 0.01, "max_keypoints": 20})({"image0": img0, "image1": img1})
    assert len(top["scores"]) == 20 and torch.all(top["scores"][:-1] >= top["scores"][1:])  # top-k by confidence
