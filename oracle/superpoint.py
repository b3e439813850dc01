"""SuperPoint oracle (torch CPU fp32)  --  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what `imcui/hloc/extractors/superpoint.py:56-57` executes:
``self.net(data, self.conf)`` where ``self.net`` is the (absent) submodule
Vincentqyw/SuperGluePretrainedNetwork ``models/superpoint.py``.  Semantics per
SURVEY.md section 8(a) rows a2-a6 and Appendix A.1; independent cross-check:
transformers/models/superpoint/modeling_superpoint.py:37-71,225-320.

parity unpinned: no reference golden vectors exist for this path.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def simple_nms(scores: torch.Tensor, nms_radius: int) -> torch.Tensor:
    """a4: iterative NMS with exact-equality semantics (upstream `simple_nms`)."""
    assert nms_radius >= 0

    def max_pool(x):
        return F.max_pool2d(x, kernel_size=nms_radius * 2 + 1, stride=1, padding=nms_radius)

    zeros = torch.zeros_like(scores)
    max_mask = scores == max_pool(scores)
    for _ in range(2):
        supp_mask = max_pool(max_mask.float()) > 0
        supp_scores = torch.where(supp_mask, zeros, scores)
        new_max_mask = supp_scores == max_pool(supp_scores)
        max_mask = max_mask | (new_max_mask & (~supp_mask))
    return torch.where(max_mask, scores, zeros)


def remove_borders(keypoints, scores, border: int, height: int, width: int):
    """a5: keep b <= y < H-b, b <= x < W-b (keypoints are (y, x) here)."""
    mask_h = (keypoints[:, 0] >= border) & (keypoints[:, 0] < (height - border))
    mask_w = (keypoints[:, 1] >= border) & (keypoints[:, 1] < (width - border))
    mask = mask_h & mask_w
    return keypoints[mask], scores[mask]


def top_k_keypoints(keypoints, scores, k: int):
    """a5: `torch.topk` only when more than k candidates; result sorted descending."""
    if k >= len(keypoints):
        return keypoints, scores
    scores, indices = torch.topk(scores, k, dim=0)
    return keypoints[indices], scores


def sample_descriptors(keypoints, descriptors, s: int = 8):
    """a6: upstream (un-fixed) sampling, `grid_sample(align_corners=True)`."""
    b, c, h, w = descriptors.shape
    keypoints = keypoints - s / 2 + 0.5
    keypoints = keypoints / torch.tensor(
        [(w * s - s / 2 - 0.5), (h * s - s / 2 - 0.5)],
    ).to(keypoints)[None]
    keypoints = keypoints * 2 - 1
    descriptors = F.grid_sample(descriptors, keypoints.view(b, 1, -1, 2), mode="bilinear", align_corners=True)
    descriptors = F.normalize(descriptors.reshape(b, c, -1), p=2, dim=1)
    return descriptors


def sample_descriptors_fix_sampling(keypoints, descriptors, s: int = 8):
    """imcui/hloc/extractors/superpoint.py:16-30 (used only if conf["fix_sampling"])."""
    b, c, h, w = descriptors.shape
    keypoints = (keypoints + 0.5) / (keypoints.new_tensor([w, h]) * s)
    keypoints = keypoints * 2 - 1
    descriptors = F.grid_sample(descriptors, keypoints.view(b, 1, -1, 2), mode="bilinear", align_corners=False)
    descriptors = F.normalize(descriptors.reshape(b, c, -1), p=2, dim=1)
    return descriptors


DEFAULT_CONF = {  # imcui/hloc/extractors/superpoint.py:34-41
    "nms_radius": 4,
    "keypoint_threshold": 0.005,
    "max_keypoints": -1,
    "remove_borders": 4,
    "fix_sampling": False,
}


class SuperPointOracle:
    def __init__(self, state_dict: dict):
        self.sd = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items()}

    def _conv(self, x, name, relu=True, pad=1):
        x = F.conv2d(x, self.sd[name + ".weight"], self.sd[name + ".bias"], padding=pad)
        return F.relu(x) if relu else x

    def encoder(self, image):
        """a2: VGG encoder, [B,1,H,W] -> [B,128,H/8,W/8]."""
        x = self._conv(image, "conv1a")
        x = self._conv(x, "conv1b")
        x = F.max_pool2d(x, 2, 2)
        x = self._conv(x, "conv2a")
        x = self._conv(x, "conv2b")
        x = F.max_pool2d(x, 2, 2)
        x = self._conv(x, "conv3a")
        x = self._conv(x, "conv3b")
        x = F.max_pool2d(x, 2, 2)
        x = self._conv(x, "conv4a")
        x = self._conv(x, "conv4b")
        return x

    def score_map(self, feat):
        """a3: detector head -> dense score map [B, 8h, 8w] (before NMS)."""
        cPa = self._conv(feat, "convPa")
        scores = self._conv(cPa, "convPb", relu=False, pad=0)
        scores = F.softmax(scores, 1)[:, :-1]
        b, _, h, w = scores.shape
        scores = scores.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8)
        scores = scores.permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)
        return scores

    def dense_descriptors(self, feat):
        """a6 (first half): convDa/convDb + L2 norm over channels."""
        cDa = self._conv(feat, "convDa")
        desc = self._conv(cDa, "convDb", relu=False, pad=0)
        return F.normalize(desc, p=2, dim=1)

    @torch.no_grad()
    def __call__(self, data: dict, conf: dict | None = None, return_intermediates: bool = False):
        conf = {**DEFAULT_CONF, **(conf or {})}
        image = data["image"].to(torch.float32).cpu()
        if image.shape[1] == 3:  # upstream converts RGB to gray with fixed weights
            scale = image.new_tensor([0.299, 0.587, 0.114]).view(1, 3, 1, 1)
            image = (image * scale).sum(1, keepdim=True)
        feat = self.encoder(image)
        dense = self.score_map(feat)
        scores = simple_nms(dense, conf["nms_radius"])
        b, H, W = scores.shape
        keypoints = [torch.nonzero(s > conf["keypoint_threshold"]) for s in scores]
        kscores = [s[tuple(k.t())] for s, k in zip(scores, keypoints)]
        keypoints, kscores = list(
            zip(*[remove_borders(k, s, conf["remove_borders"], H, W) for k, s in zip(keypoints, kscores)])
        )
        if conf["max_keypoints"] >= 0:
            keypoints, kscores = list(
                zip(*[top_k_keypoints(k, s, conf["max_keypoints"]) for k, s in zip(keypoints, kscores)])
            )
        keypoints = [torch.flip(k, [1]).float() for k in keypoints]
        ddesc = self.dense_descriptors(feat)
        sampler = sample_descriptors_fix_sampling if conf["fix_sampling"] else sample_descriptors
        descriptors = [sampler(k[None], d[None], 8)[0] for k, d in zip(keypoints, ddesc)]
        out = {"keypoints": list(keypoints), "scores": list(kscores), "descriptors": descriptors}
        if return_intermediates:
            out["_feat"] = feat
            out["_dense_scores"] = dense
            out["_nms_scores"] = scores
            out["_dense_desc"] = ddesc
        return out
