"""LoFTR oracle (torch CPU fp32)  --  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what `imcui/hloc/matchers/loftr.py:41-71` executes: it swaps image0 <-> image1, calls the
(absent, un-pinned pip dependency) `kornia.feature.LoFTR` with `default_cfg` patched at :21-24, keeps
the top-k matches by confidence (:58-65) and swaps the key names back (:68-70).  Model semantics per
SURVEY.md section 8(a) rows a13-a17 and Appendix A.3 (kornia/feature/loftr: ResNetFPN_8_2 backbone,
PositionEncodingSine with the original precedence bug, linear-attention LocalFeatureTransformer,
dual-softmax CoarseMatching, FinePreprocess + FineMatching).

parity unpinned: kornia is not installed, the reference ships no golden vectors for this path and no
independent implementation of this architecture exists in the build container.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

DEFAULT_CONF = {  # imcui/hloc/matchers/loftr.py:13-18
    "weights": "outdoor",
    "match_threshold": 0.2,
    "sinkhorn_iterations": 20,
    "max_keypoints": -1,
}
BLOCK_DIMS = [128, 196, 256]
BN_EPS = 1e-5


class LoFTROracle:
    def __init__(self, state_dict: dict, conf: dict | None = None, temp_bug_fix: bool = False):
        self.conf = {**DEFAULT_CONF, **(conf or {})}
        self.sd = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items()}
        self.temp_bug_fix = temp_bug_fix  # True only for the MINIMA weights (loftr.py:28)
        self.border_rm = 2
        self.temperature = 0.1
        self.W = 5

    # -- backbone: ResNetFPN_8_2 -----------------------------------------------------------
    def _bn(self, x, p):
        sd = self.sd
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, BN_EPS)

    def _block(self, x, p, stride):
        sd = self.sd
        y = F.relu(self._bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), p + ".bn1"))
        y = self._bn(F.conv2d(y, sd[p + ".conv2.weight"], None, 1, 1), p + ".bn2")
        if p + ".downsample.0.weight" in sd:
            x = self._bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0), p + ".downsample.1")
        return F.relu(x + y)

    def encoder_stages(self, x):
        sd = self.sd
        b = "backbone."
        x0 = F.relu(self._bn(F.conv2d(x, sd[b + "conv1.weight"], None, 2, 3), b + "bn1"))
        x1 = self._block(self._block(x0, b + "layer1.0", 1), b + "layer1.1", 1)  # 1/2
        x2 = self._block(self._block(x1, b + "layer2.0", 2), b + "layer2.1", 1)  # 1/4
        x3 = self._block(self._block(x2, b + "layer3.0", 2), b + "layer3.1", 1)  # 1/8
        return x1, x2, x3

    def backbone(self, x):
        sd = self.sd
        b = "backbone."
        x1, x2, x3 = self.encoder_stages(x)
        x3_out = F.conv2d(x3, sd[b + "layer3_outconv.weight"])
        x3_out_2x = F.interpolate(x3_out, scale_factor=2.0, mode="bilinear", align_corners=True)
        x2_out = F.conv2d(x2, sd[b + "layer2_outconv.weight"])
        x2_out = self._outconv2(x2_out + x3_out_2x, b + "layer2_outconv2")
        x2_out_2x = F.interpolate(x2_out, scale_factor=2.0, mode="bilinear", align_corners=True)
        x1_out = F.conv2d(x1, sd[b + "layer1_outconv.weight"])
        x1_out = self._outconv2(x1_out + x2_out_2x, b + "layer1_outconv2")
        return x3_out, x1_out

    def _outconv2(self, x, p):
        sd = self.sd
        x = F.conv2d(x, sd[p + ".0.weight"], None, 1, 1)
        x = F.leaky_relu(self._bn(x, p + ".1"), 0.01)
        return F.conv2d(x, sd[p + ".3.weight"], None, 1, 1)

    # -- positional encoding (coarse) ---------------------------------------------------------
    def pos_encoding(self, x):
        d_model = x.shape[1]
        h, w = x.shape[2:]
        y_position = torch.ones((h, w)).cumsum(0).float().unsqueeze(0)
        x_position = torch.ones((h, w)).cumsum(1).float().unsqueeze(0)
        if self.temp_bug_fix:
            div_term = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))
        else:  # the original implementation's operator-precedence bug, kept by the released weights
            div_term = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / d_model // 2))
        div_term = div_term[:, None, None]
        pe = torch.zeros((d_model, h, w))
        pe[0::4] = torch.sin(x_position * div_term)
        pe[1::4] = torch.cos(x_position * div_term)
        pe[2::4] = torch.sin(y_position * div_term)
        pe[3::4] = torch.cos(y_position * div_term)
        return x + pe[None]

    # -- LocalFeatureTransformer with linear attention ------------------------------------------
    def _encoder_layer(self, p, x, source, nhead):
        sd = self.sd
        bs, _, d = x.shape
        dim = d // nhead
        q = F.linear(x, sd[p + ".q_proj.weight"]).view(bs, -1, nhead, dim)
        k = F.linear(source, sd[p + ".k_proj.weight"]).view(bs, -1, nhead, dim)
        v = F.linear(source, sd[p + ".v_proj.weight"]).view(bs, -1, nhead, dim)
        # LinearAttention
        Q = F.elu(q) + 1
        K = F.elu(k) + 1
        v_length = v.size(1)
        v = v / v_length
        KV = torch.einsum("nshd,nshv->nhdv", K, v)
        Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(dim=1)) + 1e-6)
        message = torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * v_length
        message = F.linear(message.contiguous().view(bs, -1, nhead * dim), sd[p + ".merge.weight"])
        message = F.layer_norm(message, (d,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
        message = F.linear(F.relu(F.linear(torch.cat([x, message], dim=2), sd[p + ".mlp.0.weight"])), sd[p + ".mlp.2.weight"])
        message = F.layer_norm(message, (d,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
        return x + message

    def transformer(self, prefix, layer_names, feat0, feat1, nhead=8):
        for i, name in enumerate(layer_names):
            p = f"{prefix}.layers.{i}"
            if name == "self":
                feat0 = self._encoder_layer(p, feat0, feat0, nhead)
                feat1 = self._encoder_layer(p, feat1, feat1, nhead)
            else:  # cross: feat1 attends to the UPDATED feat0
                feat0 = self._encoder_layer(p, feat0, feat1, nhead)
                feat1 = self._encoder_layer(p, feat1, feat0, nhead)
        return feat0, feat1

    # -- coarse matching (dual soft-max) ---------------------------------------------------------
    def coarse_matching(self, feat_c0, feat_c1, hw0_c, hw1_c, hw0_i, thr):
        c = feat_c0.shape[-1]
        f0, f1 = feat_c0 / c**0.5, feat_c1 / c**0.5
        sim = torch.einsum("nlc,nsc->nls", f0, f1) / self.temperature
        conf = F.softmax(sim, 1) * F.softmax(sim, 2)
        n = conf.shape[0]
        mask = conf > thr
        mask = mask.view(n, hw0_c[0], hw0_c[1], hw1_c[0], hw1_c[1]).clone()
        b = self.border_rm
        if b > 0:
            mask[:, :b] = False
            mask[:, :, :b] = False
            mask[:, :, :, :b] = False
            mask[:, :, :, :, :b] = False
            mask[:, -b:] = False
            mask[:, :, -b:] = False
            mask[:, :, :, -b:] = False
            mask[:, :, :, :, -b:] = False
        mask = mask.view(n, hw0_c[0] * hw0_c[1], hw1_c[0] * hw1_c[1])
        mask = mask * (conf == conf.max(dim=2, keepdim=True)[0]) * (conf == conf.max(dim=1, keepdim=True)[0])
        mask_v, all_j_ids = mask.max(dim=2)
        b_ids, i_ids = torch.where(mask_v)
        j_ids = all_j_ids[b_ids, i_ids]
        mconf = conf[b_ids, i_ids, j_ids]
        scale = hw0_i[0] / hw0_c[0]
        mkpts0_c = torch.stack([i_ids % hw0_c[1], torch.div(i_ids, hw0_c[1], rounding_mode="trunc")], dim=1) * scale
        mkpts1_c = torch.stack([j_ids % hw1_c[1], torch.div(j_ids, hw1_c[1], rounding_mode="trunc")], dim=1) * scale
        return dict(b_ids=b_ids, i_ids=i_ids, j_ids=j_ids, mconf=mconf, mkpts0_c=mkpts0_c, mkpts1_c=mkpts1_c, conf_matrix=conf)

    # -- fine level ---------------------------------------------------------------------------
    def fine_preprocess(self, feat_f0, feat_f1, feat_c0, feat_c1, cm, stride):
        sd = self.sd
        W = self.W
        b_ids, i_ids, j_ids = cm["b_ids"], cm["i_ids"], cm["j_ids"]
        if b_ids.shape[0] == 0:
            e = torch.empty(0, W * W, feat_f0.shape[1])
            return e, e.clone()
        u0 = F.unfold(feat_f0, kernel_size=(W, W), stride=stride, padding=W // 2)
        u1 = F.unfold(feat_f1, kernel_size=(W, W), stride=stride, padding=W // 2)
        n, cww, l = u0.shape
        u0 = u0.view(n, cww // (W * W), W * W, l).permute(0, 3, 2, 1)  # n l ww c
        u1 = u1.view(n, cww // (W * W), W * W, u1.shape[2]).permute(0, 3, 2, 1)
        u0 = u0[b_ids, i_ids]
        u1 = u1[b_ids, j_ids]
        feat_c_win = F.linear(
            torch.cat([feat_c0[b_ids, i_ids], feat_c1[b_ids, j_ids]], 0),
            sd["fine_preprocess.down_proj.weight"],
            sd["fine_preprocess.down_proj.bias"],
        )
        feat_cf_win = F.linear(
            torch.cat([torch.cat([u0, u1], 0), feat_c_win[:, None, :].expand(-1, W * W, -1)], -1),
            sd["fine_preprocess.merge_feat.weight"],
            sd["fine_preprocess.merge_feat.bias"],
        )
        u0, u1 = torch.chunk(feat_cf_win, 2, dim=0)
        return u0, u1

    def fine_matching(self, feat_f0, feat_f1, cm, scale):
        M, WW, C = feat_f0.shape
        W = int(math.sqrt(WW)) if WW else self.W
        if M == 0:
            return cm["mkpts0_c"], cm["mkpts1_c"]
        picked = feat_f0[:, WW // 2, :]
        sim = torch.einsum("mc,mrc->mr", picked, feat_f1)
        heat = torch.softmax((1.0 / C**0.5) * sim, dim=1).view(-1, W, W)
        xs = torch.linspace(-1, 1, W)
        gx = xs[None, :].expand(W, W).reshape(-1)
        gy = xs[:, None].expand(W, W).reshape(-1)
        flat = heat.view(M, -1)
        coords = torch.stack([(gx * flat).sum(-1), (gy * flat).sum(-1)], -1)  # spatial_expectation2d, (x, y)
        mkpts0_f = cm["mkpts0_c"]
        mkpts1_f = cm["mkpts1_c"] + (coords * (W // 2) * scale)[: len(cm["mconf"])]
        return mkpts0_f, mkpts1_f

    # -- kornia LoFTR.forward ---------------------------------------------------------------------
    @torch.no_grad()
    def net(self, image0, image1, return_intermediates=False):
        bs = image0.shape[0]
        hw0_i, hw1_i = image0.shape[2:], image1.shape[2:]
        if hw0_i == hw1_i:
            fc, ff = self.backbone(torch.cat([image0, image1], 0))
            (feat_c0, feat_c1), (feat_f0, feat_f1) = fc.split(bs), ff.split(bs)
        else:
            (feat_c0, feat_f0), (feat_c1, feat_f1) = self.backbone(image0), self.backbone(image1)
        hw0_c, hw1_c, hw0_f = feat_c0.shape[2:], feat_c1.shape[2:], feat_f0.shape[2:]
        raw_c0 = feat_c0
        feat_c0 = self.pos_encoding(feat_c0).permute(0, 2, 3, 1).reshape(bs, -1, feat_c0.shape[1])
        feat_c1 = self.pos_encoding(feat_c1).permute(0, 2, 3, 1).reshape(bs, -1, feat_c1.shape[1])
        feat_c0, feat_c1 = self.transformer("loftr_coarse", ["self", "cross"] * 4, feat_c0, feat_c1)
        cm = self.coarse_matching(feat_c0, feat_c1, hw0_c, hw1_c, hw0_i, self.conf["match_threshold"])
        u0, u1 = self.fine_preprocess(feat_f0, feat_f1, feat_c0, feat_c1, cm, hw0_f[0] // hw0_c[0])
        if u0.shape[0] != 0:
            u0, u1 = self.transformer("loftr_fine", ["self", "cross"], u0, u1)
        mk0, mk1 = self.fine_matching(u0, u1, cm, hw0_i[0] / hw0_f[0])
        out = {"keypoints0": mk0, "keypoints1": mk1, "confidence": cm["mconf"], "batch_indexes": cm["b_ids"]}
        if return_intermediates:
            out.update(_feat_c_raw=raw_c0, _feat_f0=feat_f0, _feat_c0=feat_c0, _feat_c1=feat_c1, _conf=cm["conf_matrix"],
                       _i_ids=cm["i_ids"], _j_ids=cm["j_ids"], _fine0=u0, _fine1=u1)  # fmt: skip
        return out

    # -- the reference wrapper (imcui/hloc/matchers/loftr.py:41-71) ------------------------------------
    @torch.no_grad()
    def __call__(self, data: dict, return_intermediates=False) -> dict:
        # "For consistency with hloc pairs, we refine kpts in image0!": swap the images
        pred = self.net(data["image1"].float().cpu(), data["image0"].float().cpu(), return_intermediates)
        scores = pred["confidence"]
        top_k = self.conf["max_keypoints"]
        if top_k is not None and len(scores) > top_k:
            keep = torch.argsort(scores, descending=True)[:top_k]
            pred["keypoints0"], pred["keypoints1"] = pred["keypoints0"][keep], pred["keypoints1"][keep]
            scores = scores[keep]
        # switch the indices back
        pred["keypoints0"], pred["keypoints1"] = pred["keypoints1"], pred["keypoints0"]
        pred["scores"] = scores
        del pred["confidence"]
        return pred
