"""LightGlue oracle (torch CPU fp32)  --  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what `imcui/hloc/matchers/lightglue.py:54-75` executes: it re-packs the
flat hloc dict into ``{"image0": {image, keypoints, descriptors[B,N,D]}, ...}``
and calls the (absent) submodule cvg/LightGlue ``lightglue/lightglue.py``.
Semantics per SURVEY.md section 8(a) rows a8-a11 and Appendix A.2; the *CPU path* of
upstream is restated (SDPA/einsum attention in fp32, shared cross `sim`, point
pruning threshold -1 => always on when width_confidence > 0).  Independent
cross-check: transformers/models/lightglue/modeling_lightglue.py.

parity unpinned: no reference golden vectors exist for this path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_CONF = {  # imcui/hloc/matchers/lightglue.py:15-25 merged over upstream defaults
    "n_layers": 9,
    "num_heads": 4,
    "descriptor_dim": 256,
    "match_threshold": 0.2,
    "filter_threshold": 0.2,
    "width_confidence": 0.99,
    "depth_confidence": 0.95,
    # upstream pruning_keypoint_thresholds[device]: {"cpu": -1, "mps": -1, "cuda": 1024, "flash": 1536}
    "pruning_threshold": -1,
}


def normalize_keypoints(kpts: torch.Tensor, size) -> torch.Tensor:
    """a8: (k - [W,H]/2) / (max(W,H)/2); `size` = (W, H) from image.shape[-2:][::-1]."""
    size = torch.tensor(size, device=kpts.device, dtype=kpts.dtype)
    shift = size / 2
    scale = size.max(-1).values / 2
    return (kpts - shift[..., None, :]) / scale[..., None, None]


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)


def apply_cached_rotary_emb(freqs: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    return (t * freqs[0]) + (rotate_half(t) * freqs[1])


def sigmoid_log_double_softmax(sim, z0, z1):
    """a11: create the log assignment matrix from logits and similarity."""
    b, m, n = sim.shape
    certainties = F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    scores0 = F.log_softmax(sim, 2)
    scores1 = F.log_softmax(sim.transpose(-1, -2).contiguous(), 2).transpose(-1, -2)
    scores = sim.new_full((b, m + 1, n + 1), 0)
    scores[:, :m, :n] = scores0 + scores1 + certainties
    scores[:, :-1, -1] = F.logsigmoid(-z0.squeeze(-1))
    scores[:, -1, :-1] = F.logsigmoid(-z1.squeeze(-1))
    return scores


def filter_matches(scores: torch.Tensor, th: float):
    """a11: obtain matches from a log assignment matrix [B x M+1 x N+1]."""
    max0, max1 = scores[:, :-1, :-1].max(2), scores[:, :-1, :-1].max(1)
    m0, m1 = max0.indices, max1.indices
    indices0 = torch.arange(m0.shape[1], device=m0.device)[None]
    indices1 = torch.arange(m1.shape[1], device=m1.device)[None]
    mutual0 = indices0 == m1.gather(1, m0)
    mutual1 = indices1 == m0.gather(1, m1)
    max0_exp = max0.values.exp()
    zero = max0_exp.new_tensor(0)
    mscores0 = torch.where(mutual0, max0_exp, zero)
    mscores1 = torch.where(mutual1, mscores0.gather(1, m1), zero)
    valid0 = mutual0 & (mscores0 > th)
    valid1 = mutual1 & valid0.gather(1, m1)
    m0 = torch.where(valid0, m0, -1)
    m1 = torch.where(valid1, m1, -1)
    return m0, m1, mscores0, mscores1


class LightGlueOracle:
    def __init__(self, state_dict: dict, conf: dict | None = None):
        self.conf = {**DEFAULT_CONF, **(conf or {})}
        self.sd = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items()}
        n = self.conf["n_layers"]
        self.confidence_thresholds = [self.confidence_threshold(i) for i in range(n)]

    # -- building blocks -------------------------------------------------
    def _lin(self, x, prefix):
        return F.linear(x, self.sd[prefix + ".weight"], self.sd.get(prefix + ".bias"))

    def _ffn(self, x, prefix):
        x = self._lin(x, prefix + ".0")
        x = F.layer_norm(x, (x.shape[-1],), self.sd[prefix + ".1.weight"], self.sd[prefix + ".1.bias"], 1e-5)
        x = F.gelu(x)
        return self._lin(x, prefix + ".3")

    def posenc(self, kpts):
        """a8: LearnableFourierPositionalEncoding -> [2, B, 1, N, 64]."""
        projected = F.linear(kpts, self.sd["posenc.Wr.weight"])
        emb = torch.stack([torch.cos(projected), torch.sin(projected)], 0).unsqueeze(-3)
        return emb.repeat_interleave(2, dim=-1)

    def self_block(self, i, x, encoding):
        """a9 SelfBlock: interleaved Wqkv (heads, head_dim, 3), RoPE on q/k, softmax(QK^T/8)V."""
        p = f"transformers.{i}.self_attn"
        h = self.conf["num_heads"]
        qkv = self._lin(x, p + ".Wqkv")
        qkv = qkv.unflatten(-1, (h, -1, 3)).transpose(1, 2)
        q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
        q = apply_cached_rotary_emb(encoding, q)
        k = apply_cached_rotary_emb(encoding, k)
        if q.shape[-2] == 0 or k.shape[-2] == 0:
            context = q.new_zeros((*q.shape[:-1], v.shape[-1]))
        else:
            context = F.scaled_dot_product_attention(q.contiguous(), k.contiguous(), v.contiguous())
        message = self._lin(context.transpose(1, 2).flatten(start_dim=-2), p + ".out_proj")
        return x + self._ffn(torch.cat([x, message], -1), p + ".ffn")

    def cross_block(self, i, x0, x1):
        """a9 CrossBlock (CPU path): shared `sim`, softmax along both axes."""
        p = f"transformers.{i}.cross_attn"
        h = self.conf["num_heads"]
        qk0, qk1 = self._lin(x0, p + ".to_qk"), self._lin(x1, p + ".to_qk")
        v0, v1 = self._lin(x0, p + ".to_v"), self._lin(x1, p + ".to_v")
        qk0, qk1, v0, v1 = (t.unflatten(-1, (h, -1)).transpose(1, 2) for t in (qk0, qk1, v0, v1))
        scale = qk0.shape[-1] ** -0.5
        qk0, qk1 = qk0 * scale**0.5, qk1 * scale**0.5
        sim = torch.einsum("bhid, bhjd -> bhij", qk0, qk1)
        attn01 = F.softmax(sim, dim=-1)
        attn10 = F.softmax(sim.transpose(-2, -1).contiguous(), dim=-1)
        m0 = torch.einsum("bhij, bhjd -> bhid", attn01, v1)
        m1 = torch.einsum("bhji, bhjd -> bhid", attn10.transpose(-2, -1), v0)
        m0, m1 = (t.transpose(1, 2).flatten(start_dim=-2) for t in (m0, m1))
        m0, m1 = self._lin(m0, p + ".to_out"), self._lin(m1, p + ".to_out")
        x0 = x0 + self._ffn(torch.cat([x0, m0], -1), p + ".ffn")
        x1 = x1 + self._ffn(torch.cat([x1, m1], -1), p + ".ffn")
        return x0, x1

    def token_confidence(self, i, desc0, desc1):
        p = f"token_confidence.{i}.token.0"
        return (
            torch.sigmoid(self._lin(desc0, p)).squeeze(-1),
            torch.sigmoid(self._lin(desc1, p)).squeeze(-1),
        )

    def get_matchability(self, i, desc):
        return torch.sigmoid(self._lin(desc, f"log_assignment.{i}.matchability")).squeeze(-1)

    def log_assignment(self, i, desc0, desc1):
        p = f"log_assignment.{i}"
        mdesc0, mdesc1 = self._lin(desc0, p + ".final_proj"), self._lin(desc1, p + ".final_proj")
        d = mdesc0.shape[-1]
        mdesc0, mdesc1 = mdesc0 / d**0.25, mdesc1 / d**0.25
        sim = torch.einsum("bmd,bnd->bmn", mdesc0, mdesc1)
        z0 = self._lin(desc0, p + ".matchability")
        z1 = self._lin(desc1, p + ".matchability")
        return sigmoid_log_double_softmax(sim, z0, z1), sim

    def confidence_threshold(self, layer_index: int) -> float:
        """a10: scaled confidence threshold."""
        threshold = 0.8 + 0.1 * np.exp(-4.0 * layer_index / self.conf["n_layers"])
        return np.clip(threshold, 0, 1)

    def get_pruning_mask(self, confidences, scores, layer_index):
        keep = scores > (1 - self.conf["width_confidence"])
        if confidences is not None:  # low-confidence points are never pruned
            keep |= confidences <= self.confidence_thresholds[layer_index]
        return keep

    def check_if_stop(self, confidences0, confidences1, layer_index, num_points):
        confidences = torch.cat([confidences0, confidences1], -1)
        threshold = self.confidence_thresholds[layer_index]
        ratio_confident = 1.0 - (confidences < threshold).float().sum() / num_points
        return ratio_confident > self.conf["depth_confidence"]

    # -- forward -----------------------------------------------------------
    @torch.no_grad()
    def __call__(self, data: dict, return_intermediates: bool = False) -> dict:
        """`data` is the flat hloc matcher dict (imcui/hloc/match_features.py:217-226)."""
        conf = self.conf
        # imcui/hloc/matchers/lightglue.py:56-70 (permute descriptors to [B,N,D])
        kpts0, kpts1 = data["keypoints0"].float().cpu(), data["keypoints1"].float().cpu()
        desc0 = data["descriptors0"].float().cpu().permute(0, 2, 1).contiguous()
        desc1 = data["descriptors1"].float().cpu().permute(0, 2, 1).contiguous()
        b, m, _ = kpts0.shape
        b, n, _ = kpts1.shape
        size0 = data["image0"].shape[-2:][::-1]
        size1 = data["image1"].shape[-2:][::-1]
        kpts0 = normalize_keypoints(kpts0, size0).clone()
        kpts1 = normalize_keypoints(kpts1, size1).clone()
        if self.sd["posenc.Wr.weight"].shape[1] == 4:  # add_scale_ori (upstream: sift, doghardnet)
            kpts0 = torch.cat([kpts0, data["scales0"].float().cpu().unsqueeze(-1), data["oris0"].float().cpu().unsqueeze(-1)], -1)
            kpts1 = torch.cat([kpts1, data["scales1"].float().cpu().unsqueeze(-1), data["oris1"].float().cpu().unsqueeze(-1)], -1)
        # input_proj is Identity for 256-d SuperPoint descriptors, Linear(input_dim, 256) otherwise (disk, aliked, sift: 128)
        if "input_proj.weight" in self.sd:
            desc0, desc1 = self._lin(desc0, "input_proj"), self._lin(desc1, "input_proj")
        encoding0 = self.posenc(kpts0)
        encoding1 = self.posenc(kpts1)

        do_early_stop = conf["depth_confidence"] > 0
        do_point_pruning = conf["width_confidence"] > 0
        pruning_th = conf["pruning_threshold"]  # upstream pruning_keypoint_thresholds[device]; "cpu" = -1
        if do_point_pruning:
            ind0 = torch.arange(0, m)[None]
            ind1 = torch.arange(0, n)[None]
            prune0 = torch.ones_like(ind0)
            prune1 = torch.ones_like(ind1)
        token0, token1 = None, None
        inter = []
        i = 0
        for i in range(conf["n_layers"]):
            if desc0.shape[1] == 0 or desc1.shape[1] == 0:  # no keypoints
                break
            desc0 = self.self_block(i, desc0, encoding0)
            desc1 = self.self_block(i, desc1, encoding1)
            desc0, desc1 = self.cross_block(i, desc0, desc1)
            if return_intermediates:
                inter.append((desc0.clone(), desc1.clone()))
            if i == conf["n_layers"] - 1:
                continue  # no early stopping or adaptive width at last layer
            if do_early_stop:
                token0, token1 = self.token_confidence(i, desc0, desc1)
                if self.check_if_stop(token0[..., :m], token1[..., :n], i, m + n):
                    break
            if do_point_pruning and desc0.shape[-2] > pruning_th:
                scores0 = self.get_matchability(i, desc0)
                prunemask0 = self.get_pruning_mask(token0, scores0, i)
                keep0 = torch.where(prunemask0)[1]
                ind0 = ind0.index_select(1, keep0)
                desc0 = desc0.index_select(1, keep0)
                encoding0 = encoding0.index_select(-2, keep0)
                prune0[:, ind0] += 1
            if do_point_pruning and desc1.shape[-2] > pruning_th:
                scores1 = self.get_matchability(i, desc1)
                prunemask1 = self.get_pruning_mask(token1, scores1, i)
                keep1 = torch.where(prunemask1)[1]
                ind1 = ind1.index_select(1, keep1)
                desc1 = desc1.index_select(1, keep1)
                encoding1 = encoding1.index_select(-2, keep1)
                prune1[:, ind1] += 1

        if desc0.shape[1] == 0 or desc1.shape[1] == 0:  # no keypoints
            m0 = desc0.new_full((b, m), -1, dtype=torch.long)
            m1 = desc1.new_full((b, n), -1, dtype=torch.long)
            mscores0 = desc0.new_zeros((b, m))
            mscores1 = desc1.new_zeros((b, n))
            if not do_point_pruning:
                prune0 = torch.ones_like(mscores0) * conf["n_layers"]
                prune1 = torch.ones_like(mscores1) * conf["n_layers"]
            return {
                "matches0": m0,
                "matches1": m1,
                "matching_scores0": mscores0,
                "matching_scores1": mscores1,
                "stop": i + 1,
                "matches": [desc0.new_empty((0, 2), dtype=torch.long) for _ in range(b)],
                "scores": [desc0.new_empty((0,)) for _ in range(b)],
                "prune0": prune0,
                "prune1": prune1,
            }

        scores, sim = self.log_assignment(i, desc0, desc1)
        m0, m1, mscores0, mscores1 = filter_matches(scores, conf["filter_threshold"])
        matches, mscores = [], []
        for k in range(b):
            valid = m0[k] > -1
            m_indices_0 = torch.where(valid)[0]
            m_indices_1 = m0[k][valid]
            if do_point_pruning:
                m_indices_0 = ind0[k, m_indices_0]
                m_indices_1 = ind1[k, m_indices_1]
            matches.append(torch.stack([m_indices_0, m_indices_1], -1))
            mscores.append(mscores0[k][valid])

        if do_point_pruning:
            m0_ = torch.full((b, m), -1, dtype=m0.dtype)
            m1_ = torch.full((b, n), -1, dtype=m1.dtype)
            m0_[:, ind0] = torch.where(m0 == -1, -1, ind1.gather(1, m0.clamp(min=0)))
            m1_[:, ind1] = torch.where(m1 == -1, -1, ind0.gather(1, m1.clamp(min=0)))
            mscores0_ = torch.zeros((b, m))
            mscores1_ = torch.zeros((b, n))
            mscores0_[:, ind0] = mscores0
            mscores1_[:, ind1] = mscores1
            m0, m1, mscores0, mscores1 = m0_, m1_, mscores0_, mscores1_
        else:
            prune0 = torch.ones_like(mscores0) * conf["n_layers"]
            prune1 = torch.ones_like(mscores1) * conf["n_layers"]

        out = {
            "matches0": m0,
            "matches1": m1,
            "matching_scores0": mscores0,
            "matching_scores1": mscores1,
            "stop": i + 1,
            "matches": matches,
            "scores": mscores,
            "prune0": prune0,
            "prune1": prune1,
        }
        if return_intermediates:
            out["_layers"] = inter
            out["_log_assignment"] = scores
            out["_sim"] = sim
            out["_final_desc"] = (desc0, desc1)
            if do_point_pruning:
                out["_ind0"], out["_ind1"] = ind0, ind1
        return out
