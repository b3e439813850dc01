"""SuperGlue oracle (torch CPU fp32)  --  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what `imcui/hloc/matchers/superglue.py:42-43` executes: the wrapper hands the flat
hloc dict straight to the (absent) submodule Vincentqyw/SuperGluePretrainedNetwork
``models/superglue.py`` (fork of magicleap's; the fork only adds ``conf["weights_path"]``,
superglue.py:32-37).  Semantics per SURVEY.md section 8(f) rank 1 and the published upstream
algorithm:

  * keypoints normalised by ``(k - [W,H]/2) / (0.7 max(W,H))``; keypoint encoder
    ``MLP[3,32,64,128,256,256]`` (Conv1d k=1 + BatchNorm1d + ReLU, last layer bare) on
    ``[x, y, score]``, added to the descriptors;
  * 18 attentional propagation layers, ``['self','cross'] * 9``: three 256->256 projections, 4 heads
    whose channels are INTERLEAVED (``view(b, 64, 4, n)``: channel = d * 4 + head), soft-max(QK^T / 8) V,
    ``merge`` 256->256, ``MLP[512,512,256]`` on ``cat[x, message]`` (BatchNorm after the first layer),
    residual add; both images are updated from the same (old) state;
  * ``final_proj`` 256->256, ``scores = md0 . md1^T / 16``, log-domain Sinkhorn with a dust-bin row and
    column at ``bin_score``, ``sinkhorn_iterations`` rounds, then mutual arg-max + ``match_threshold``.

Independent cross-check: transformers/models/superglue/modeling_superglue.py (heads contiguous there,
tests/test_oracle_crosscheck.py permutes the weights).

parity unpinned: no reference golden vectors exist for this path.
"""
from __future__ import annotations

import torch

DEFAULT_CONF = {  # imcui/hloc/matchers/superglue.py:14-19 merged over the upstream defaults
    "descriptor_dim": 256,
    "keypoint_encoder": [32, 64, 128, 256],
    "GNN_layers": ["self", "cross"] * 9,
    "sinkhorn_iterations": 100,
    "match_threshold": 0.2,
}
BN_EPS = 1e-5  # nn.BatchNorm1d default


def normalize_keypoints(kpts: torch.Tensor, image_shape) -> torch.Tensor:
    """(k - size/2) / (0.7 * max(size)); `image_shape` = image.shape = (B, C, H, W)."""
    _, _, height, width = image_shape
    one = kpts.new_tensor(1)
    size = torch.stack([one * width, one * height])[None]
    center = size / 2
    scaling = size.max(1, keepdim=True).values * 0.7
    return (kpts - center[:, None, :]) / scaling[:, None, :]


def log_sinkhorn_iterations(Z, log_mu, log_nu, iters: int):
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(Z + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(Z + u.unsqueeze(2), dim=1)
    return Z + u.unsqueeze(2) + v.unsqueeze(1)


def log_optimal_transport(scores, alpha, iters: int):
    """Differentiable optimal transport in log space with one dust-bin per side."""
    b, m, n = scores.shape
    one = scores.new_tensor(1)
    ms, ns = (m * one).to(scores), (n * one).to(scores)
    bins0 = alpha.expand(b, m, 1)
    bins1 = alpha.expand(b, 1, n)
    corner = alpha.expand(b, 1, 1)
    couplings = torch.cat([torch.cat([scores, bins0], -1), torch.cat([bins1, corner], -1)], 1)
    norm = -(ms + ns).log()
    log_mu = torch.cat([norm.expand(m), ns.log()[None] + norm])
    log_nu = torch.cat([norm.expand(n), ms.log()[None] + norm])
    log_mu, log_nu = log_mu[None].expand(b, -1), log_nu[None].expand(b, -1)
    Z = log_sinkhorn_iterations(couplings, log_mu, log_nu, iters)
    return Z - norm  # multiply probabilities by M + N


class SuperGlueOracle:
    """state dict in the upstream layout (imcui_hip/synth_weights.py superglue_state_dict) -> callable(hloc dict)."""

    def __init__(self, sd: dict, conf: dict | None = None):
        self.sd = {k: v.detach().float() if v.is_floating_point() else v for k, v in sd.items()}
        self.conf = {**DEFAULT_CONF, **(conf or {})}

    # Conv1d(k=1) on [B, C, N]
    def _conv(self, x, name):
        w = self.sd[name + ".weight"]
        return torch.einsum("oc,bcn->bon", w.reshape(w.shape[0], -1), x) + self.sd[name + ".bias"][None, :, None]

    def _bn(self, x, name):
        s = self.sd
        inv = torch.rsqrt(s[name + ".running_var"] + BN_EPS)
        return (x - s[name + ".running_mean"][None, :, None]) * inv[None, :, None] * s[name + ".weight"][None, :, None] + s[
            name + ".bias"
        ][None, :, None]

    def _mlp(self, x, prefix, n_layers):
        """nn.Sequential(Conv1d, BN, ReLU, ..., Conv1d): module indices 0,1,2 | 3,4,5 | ..."""
        for i in range(n_layers):
            x = self._conv(x, f"{prefix}.{3 * i}")
            if i < n_layers - 1:
                x = torch.relu(self._bn(x, f"{prefix}.{3 * i + 1}"))
        return x

    def _kenc(self, kpts, scores):
        inputs = torch.cat([kpts.transpose(1, 2), scores.unsqueeze(1)], dim=1)
        return self._mlp(inputs, "kenc.encoder", len(self.conf["keypoint_encoder"]) + 1)

    def _attn(self, x, source, p):
        b = x.shape[0]
        q, k, v = (self._conv(t, f"{p}.proj.{i}").view(b, 64, 4, -1) for i, t in enumerate((x, source, source)))
        scores = torch.einsum("bdhn,bdhm->bhnm", q, k) / 64**0.5
        prob = torch.softmax(scores, dim=-1)
        msg = torch.einsum("bhnm,bdhm->bdhn", prob, v)
        return self._conv(msg.contiguous().view(b, 256, -1), f"{p}.merge")

    def _layer(self, x, source, i):
        message = self._attn(x, source, f"gnn.layers.{i}.attn")
        return self._mlp(torch.cat([x, message], dim=1), f"gnn.layers.{i}.mlp", 2)

    @torch.no_grad()
    def __call__(self, data: dict) -> dict:
        desc0, desc1 = data["descriptors0"].float(), data["descriptors1"].float()
        kpts0, kpts1 = data["keypoints0"].float(), data["keypoints1"].float()
        if kpts0.shape[1] == 0 or kpts1.shape[1] == 0:  # no keypoints
            shape0, shape1 = kpts0.shape[:-1], kpts1.shape[:-1]
            return {
                "matches0": kpts0.new_full(shape0, -1, dtype=torch.int),
                "matches1": kpts1.new_full(shape1, -1, dtype=torch.int),
                "matching_scores0": kpts0.new_zeros(shape0),
                "matching_scores1": kpts1.new_zeros(shape1),
            }
        kpts0 = normalize_keypoints(kpts0, data["image0"].shape)
        kpts1 = normalize_keypoints(kpts1, data["image1"].shape)
        desc0 = desc0 + self._kenc(kpts0, data["scores0"].float())
        desc1 = desc1 + self._kenc(kpts1, data["scores1"].float())
        for i, name in enumerate(self.conf["GNN_layers"]):
            src0, src1 = (desc1, desc0) if name == "cross" else (desc0, desc1)
            delta0, delta1 = self._layer(desc0, src0, i), self._layer(desc1, src1, i)
            desc0, desc1 = desc0 + delta0, desc1 + delta1
        md0, md1 = self._conv(desc0, "final_proj"), self._conv(desc1, "final_proj")
        scores = torch.einsum("bdn,bdm->bnm", md0, md1)
        scores = scores / self.conf["descriptor_dim"] ** 0.5
        scores = log_optimal_transport(scores, self.sd["bin_score"], iters=self.conf["sinkhorn_iterations"])
        max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
        indices0, indices1 = max0.indices, max1.indices
        ar0 = torch.arange(indices0.shape[1])[None]
        ar1 = torch.arange(indices1.shape[1])[None]
        mutual0 = ar0 == indices1.gather(1, indices0)
        mutual1 = ar1 == indices0.gather(1, indices1)
        zero = scores.new_tensor(0)
        mscores0 = torch.where(mutual0, max0.values.exp(), zero)
        mscores1 = torch.where(mutual1, mscores0.gather(1, indices1), zero)
        valid0 = mutual0 & (mscores0 > self.conf["match_threshold"])
        valid1 = mutual1 & valid0.gather(1, indices1)
        indices0 = torch.where(valid0, indices0, indices0.new_tensor(-1))
        indices1 = torch.where(valid1, indices1, indices1.new_tensor(-1))
        return {
            "matches0": indices0,  # use -1 for invalid match
            "matches1": indices1,
            "matching_scores0": mscores0,
            "matching_scores1": mscores1,
        }
