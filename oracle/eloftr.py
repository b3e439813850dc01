"""EfficientLoFTR oracle (torch CPU fp32)  --  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates what `imcui/hloc/matchers/eloftr.py:68-104` executes: it swaps image0 <-> image1 (:70-78), calls the
(absent, un-vendored submodule) `third_party/EfficientLoFTR/src/loftr.LoFTR` built from `full_default_cfg` with the
match threshold patched (:53-54) and re-parameterised (:61), keeps the top-k matches by confidence (:90-97) and swaps the
key names back (:100).  Model semantics: Wang et al., "Efficient LoFTR" (CVPR 2024) as restated by the maintained port
`transformers/models/efficientloftr/modeling_efficientloftr.py` (RepVGG backbone at 1/2, 1/4, 1/8; aggregated
self / cross attention with 2-D rotary embedding; dual-softmax coarse matching; fine feature fusion to full
resolution; two-stage fine matching).  The state-dict names are the port's.

Pinning: every stage up to and including the confidence matrix and the fine feature maps is checked against that port
(tests/test_oracle_crosscheck.py); the match LIST (mutual nearest neighbours above the threshold, row-major, upstream
`get_coarse_match` order as in oracle/loftr.py) and the pairing of fine windows follow the upstream semantics, which
the port's per-cell output only reproduces for matches on the diagonal -- so the end-to-end output is
"parity unpinned" in the sense of the task statement (no upstream sources, no golden vectors in the reference).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

DEFAULT_CONF = {  # imcui/hloc/matchers/eloftr.py:25-34
    "model_name": "eloftr_outdoor.ckpt",
    "match_threshold": 0.2,
    "max_keypoints": -1,
    "model_type": "full",
    "precision": "fp32",
}
STAGE_BLOCKS = [1, 2, 4, 14]
STAGE_STRIDE = [2, 1, 2, 2]
STAGE_DIMS = [64, 64, 128, 256]
BN_EPS = 1e-5
HEADS = 8
AGG = 4  # aggregation kernel = stride (queries: depth-wise conv, keys / values: max-pool)
FINE_W = 8
SLICE_DIM = 8
REGRESS_TEMP = 10.0
P = "efficientloftr."


class ELoFTROracle:
    def __init__(self, state_dict: dict, conf: dict | None = None):
        self.conf = {**DEFAULT_CONF, **(conf or {})}
        self.sd = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items() if torch.is_tensor(v)}
        self.border_rm = 2
        self.temperature = 0.1

    # -- RepVGG backbone (training-time form: 3x3 + 1x1 + identity branches, each with its BatchNorm) ---------------
    def _bn(self, x, p, eps=BN_EPS):
        sd = self.sd
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)

    def _repvgg(self, x, p, stride):
        sd = self.sd
        y = self._bn(F.conv2d(x, sd[p + ".conv1.conv.weight"], None, stride, 1), p + ".conv1.norm")
        y = y + self._bn(F.conv2d(x, sd[p + ".conv2.conv.weight"], None, stride, 0), p + ".conv2.norm")
        if p + ".identity.weight" in sd:
            y = y + self._bn(x, p + ".identity")
        return F.relu(y)

    def backbone(self, x):
        """[N,1,H,W] -> features at 1/2 (64), 1/4 (128), 1/8 (256)."""
        outs = []
        for s in range(4):
            for b in range(STAGE_BLOCKS[s]):
                x = self._repvgg(x, f"{P}backbone.stages.{s}.blocks.{b}", STAGE_STRIDE[s] if b == 0 else 1)
            outs.append(x)
        return outs[1], outs[2], outs[3]

    # -- rotary embedding of the aggregated grid -----------------------------------------------------------------
    @staticmethod
    def rope(eh, ew, hidden=256):
        dim = int((hidden // HEADS) * 4.0)  # partial_rotary_factor 4 on head_dim 32
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))  # [64]
        i = torch.ones(eh, ew).cumsum(0).unsqueeze(-1)
        j = torch.ones(eh, ew).cumsum(1).unsqueeze(-1)
        emb = torch.zeros(eh, ew, hidden // 2)
        emb[:, :, 0::2] = i * inv_freq
        emb[:, :, 1::2] = j * inv_freq
        cos = emb.cos().repeat_interleave(2, dim=-1).reshape(1, eh * ew, hidden)
        sin = emb.sin().repeat_interleave(2, dim=-1).reshape(1, eh * ew, hidden)
        return cos, sin

    @staticmethod
    def _rotate_half(x):
        x1, x2 = x[..., ::2], x[..., 1::2]
        return torch.stack([-x2, x1], dim=-1).flatten(-2)

    # -- one aggregated-attention block ---------------------------------------------------------------------------
    def _agg_attention(self, p, x, src, rope):
        """x, src: [N,256,h,w] (src = x for self attention).  Returns x + LN(MLP(cat[x, up(attn)]))."""
        sd = self.sd
        n, c, h, w = x.shape
        q = F.conv2d(x, sd[p + ".aggregation.q_aggregation.weight"], None, AGG, 0, 1, c)
        kv = F.max_pool2d(src, AGG, AGG)
        ah, aw = q.shape[2:]
        q = q.permute(0, 2, 3, 1).reshape(n, ah * aw, c)
        kv = kv.permute(0, 2, 3, 1).reshape(n, -1, c)
        nw, nb = sd[p + ".aggregation.norm.weight"], sd[p + ".aggregation.norm.bias"]
        q, kv = F.layer_norm(q, (c,), nw, nb, 1e-5), F.layer_norm(kv, (c,), nw, nb, 1e-5)
        qs = F.linear(q, sd[p + ".attention.q_proj.weight"])
        ks = F.linear(kv, sd[p + ".attention.k_proj.weight"])
        vs = F.linear(kv, sd[p + ".attention.v_proj.weight"])
        if rope is not None:
            cos, sin = rope
            qs = qs * cos + self._rotate_half(qs) * sin
            ks = ks * cos + self._rotate_half(ks) * sin
        d = c // HEADS
        qh = qs.view(n, -1, HEADS, d).transpose(1, 2)
        kh = ks.view(n, -1, HEADS, d).transpose(1, 2)
        vh = vs.view(n, -1, HEADS, d).transpose(1, 2)
        att = torch.softmax(qh @ kh.transpose(2, 3) * d**-0.5, dim=-1)
        o = (att @ vh).transpose(1, 2).reshape(n, ah * aw, c)
        o = F.linear(o, sd[p + ".attention.o_proj.weight"])
        o = o.permute(0, 2, 1).reshape(n, c, ah, aw)
        o = F.interpolate(o, scale_factor=AGG, mode="bilinear", align_corners=False)
        t = torch.cat([x, o], 1).permute(0, 2, 3, 1)
        t = F.linear(F.leaky_relu(F.linear(t, sd[p + ".mlp.fc1.weight"]), 0.01), sd[p + ".mlp.fc2.weight"])
        t = F.layer_norm(t, (c,), sd[p + ".mlp.layer_norm.weight"], sd[p + ".mlp.layer_norm.bias"], 1e-5)
        return x + t.permute(0, 3, 1, 2)

    def transformer(self, f0, f1):
        """f0 [B,256,h0,w0], f1 [B,256,h1,w1] -> transformed.  Self attention with the rotary embedding of the map's own
        aggregated grid; cross attention without; image 1 attends to the UPDATED image 0 (upstream behaviour the port
        keeps, see its layer forward).  The two maps may differ in size (the attention is between token sets)."""
        ropes = [self.rope((f.shape[2] - AGG) // AGG + 1, (f.shape[3] - AGG) // AGG + 1) for f in (f0, f1)]
        for i in range(4):
            p = f"{P}local_feature_transformer.layers.{i}"
            f0 = self._agg_attention(p + ".self_attention", f0, f0, ropes[0])
            f1 = self._agg_attention(p + ".self_attention", f1, f1, ropes[1])
            f0 = self._agg_attention(p + ".cross_attention", f0, f1, None)
            f1 = self._agg_attention(p + ".cross_attention", f1, f0, None)
        return f0, f1

    # -- coarse matching (dual soft-max, upstream match-list order) ---------------------------------------------------
    def coarse_matching(self, fc0, fc1, hw_i, thr):
        n, c, h, w = fc0.shape
        h1, w1 = fc1.shape[2:]
        f0 = fc0.permute(0, 2, 3, 1).reshape(n, h * w, c) / c**0.5
        f1 = fc1.permute(0, 2, 3, 1).reshape(n, h1 * w1, c) / c**0.5
        sim = f0 @ f1.transpose(-1, -2) / self.temperature
        conf = F.softmax(sim, 1) * F.softmax(sim, 2)
        mask = (conf > thr).view(n, h, w, h1, w1).clone()
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
        mask = mask.view(n, h * w, h1 * w1)
        mask = mask * (conf == conf.max(dim=2, keepdim=True)[0]) * (conf == conf.max(dim=1, keepdim=True)[0])
        mask_v, all_j = mask.max(dim=2)
        b_ids, i_ids = torch.where(mask_v)
        j_ids = all_j[b_ids, i_ids]
        scale = hw_i[0] / h  # as in LoFTR: the scale of image 0 for both key-point sets (8 either way)
        mk0 = torch.stack([i_ids % w, torch.div(i_ids, w, rounding_mode="trunc")], 1) * scale
        mk1 = torch.stack([j_ids % w1, torch.div(j_ids, w1, rounding_mode="trunc")], 1) * scale
        return dict(b_ids=b_ids, i_ids=i_ids, j_ids=j_ids, mconf=conf[b_ids, i_ids, j_ids], mkpts0_c=mk0, mkpts1_c=mk1, conf_matrix=conf)

    # -- fine feature fusion: 1/8 -> 1/4 -> 1/2 -> 1/1, 64 channels ------------------------------------------------------
    def _outconv_block(self, p, hidden, residual):
        sd = self.sd
        r = F.conv2d(residual, sd[p + ".out_conv1.weight"]) + hidden
        r = F.conv2d(r, sd[p + ".out_conv2.weight"], None, 1, 1)
        r = F.leaky_relu(self._bn(r, p + ".batch_norm"), 0.01)
        self._last_pre_upsample = F.conv2d(r, sd[p + ".out_conv3.weight"], None, 1, 1)
        return F.interpolate(self._last_pre_upsample, scale_factor=2.0, mode="bilinear", align_corners=False)

    def fine_features(self, fc, x2, x1):
        """fc [N,256,h/8,w/8] (transformed coarse, already divided by sqrt(256)), x2 (1/4, 128), x1 (1/2, 64) -> [N,64,H,W]."""
        h = F.conv2d(fc, self.sd["refinement_layer.out_conv.weight"])
        h = F.interpolate(h, scale_factor=2.0, mode="bilinear", align_corners=False)
        h = self._outconv_block("refinement_layer.out_conv_layers.0", h, x2)
        return self._outconv_block("refinement_layer.out_conv_layers.1", h, x1)

    # -- two-stage fine matching ------------------------------------------------------------------------------------
    @staticmethod
    def fine_windows(ff0, ff1, cm, stride):
        b, i, j = cm["b_ids"], cm["i_ids"], cm["j_ids"]
        c = ff0.shape[1]
        u0 = F.unfold(ff0, FINE_W, stride=stride, padding=0)
        u1 = F.unfold(ff1, FINE_W + 2, stride=stride, padding=1)
        n, _, l = u0.shape
        u0 = u0.view(n, c, FINE_W**2, l).permute(0, 3, 2, 1)[b, i]           # [M, 64, C]
        u1 = u1.view(n, c, (FINE_W + 2) ** 2, u1.shape[2]).permute(0, 3, 2, 1)[b, j]   # [M, 100, C]
        return u0, u1

    @staticmethod
    def fine_matching(u0, u1, cm, fine_scale):
        m, ww, c = u0.shape
        if m == 0:
            return cm["mkpts0_c"], cm["mkpts1_c"]
        w = FINE_W
        a0, a1 = u0[..., : c - SLICE_DIM], u1[..., : c - SLICE_DIM]
        a0, a1 = a0 / a0.shape[-1] ** 0.5, a1 / a1.shape[-1] ** 0.5
        s = a0 @ a1.transpose(-1, -2)                                        # [M, 64, 100]
        cf = F.softmax(s, 1) * F.softmax(s, 2)
        cf = cf.reshape(m, ww, w + 2, w + 2)[..., 1:-1, 1:-1].reshape(m, ww * ww)
        idx = cf.argmax(-1)
        il, ir = torch.div(idx, ww, rounding_mode="trunc"), idx % ww
        g = torch.stack(torch.meshgrid(torch.arange(w, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij"), -1)
        g = g.flip(-1).reshape(ww, 2) - (w // 2) + 0.5                       # (x, y) of a window position
        mk0 = cm["mkpts0_c"] + g[il] * fine_scale
        mk1 = cm["mkpts1_c"] + g[ir] * fine_scale
        # second stage: the last SLICE_DIM channels, 3 x 3 neighbourhood of the picked position in the 10 x 10 window
        b0 = u0[..., c - SLICE_DIM :][torch.arange(m), il]                    # [M, 8]
        b1 = u1[..., c - SLICE_DIM :] / SLICE_DIM**0.5                        # [M, 100, 8]
        s2 = (b1 @ b0[:, :, None])[..., 0].view(m, w + 2, w + 2)            # [M, 10, 10]
        ri, rj = torch.div(ir, w, rounding_mode="trunc"), ir % w             # position in the 8 x 8 interior
        # interior position (ri, rj) sits at (ri + 1, rj + 1) of the padded window; the port indexes the 10 x 10 grid with
        # (ri + d, rj + d), d in {-1, 0, 1}: i.e. the 3 x 3 block whose CENTRE is padded position (ri, rj)
        d = torch.tensor([-1, 0, 1])
        rows = (ri[:, None, None] + d[None, :, None]).expand(m, 3, 3)
        cols = (rj[:, None, None] + d[None, None, :]).expand(m, 3, 3)
        patch = s2[torch.arange(m)[:, None, None], rows, cols].reshape(m, 9)
        heat = F.softmax(patch / REGRESS_TEMP, -1).view(m, 3, 3)
        xs = torch.tensor([-1.0, 0.0, 1.0])
        ex = (heat * xs[None, None, :]).sum((1, 2))
        ey = (heat * xs[None, :, None]).sum((1, 2))
        mk1 = mk1 + torch.stack([ex, ey], -1) * (3 // 2) * fine_scale
        return mk0, mk1

    # -- upstream LoFTR.forward -----------------------------------------------------------------------------------------
    @torch.no_grad()
    def net(self, image0, image1, return_intermediates=False):
        bs = image0.shape[0]
        hw_i = image0.shape[2:]
        if image1.shape == image0.shape:
            x1, x2, x3 = self.backbone(torch.cat([image0, image1], 0))
            x1, x2, x3 = (x1[:bs], x1[bs:]), (x2[:bs], x2[bs:]), (x3[:bs], x3[bs:])
        else:  # images of different sizes go through the backbone one after the other (upstream LoFTR.forward does the same)
            a, b = self.backbone(image0), self.backbone(image1)
            x1, x2, x3 = (a[0], b[0]), (a[1], b[1]), (a[2], b[2])
        f0, f1 = self.transformer(x3[0], x3[1])
        cm = self.coarse_matching(f0, f1, hw_i, self.conf["match_threshold"])
        ff0 = self.fine_features(f0 / 256**0.5, x2[0], x1[0])
        half0 = self._last_pre_upsample
        ff1 = self.fine_features(f1 / 256**0.5, x2[1], x1[1])
        half1 = self._last_pre_upsample
        u0, u1 = self.fine_windows(ff0, ff1, cm, ff0.shape[2] // f0.shape[2])
        mk0, mk1 = self.fine_matching(u0, u1, cm, hw_i[0] / ff0.shape[2])
        out = {"keypoints0": mk0, "keypoints1": mk1, "confidence": cm["mconf"], "batch_indexes": cm["b_ids"]}
        if return_intermediates:
            out.update(_x1=x1, _x2=x2, _x3=x3, _feat_c0=f0, _feat_c1=f1, _conf=cm["conf_matrix"], _i_ids=cm["i_ids"], _j_ids=cm["j_ids"],
                       _fine=(ff0, ff1), _fine_half=(half0, half1), _win0=u0, _win1=u1, _mkpts0_c=cm["mkpts0_c"], _mkpts1_c=cm["mkpts1_c"])  # fmt: skip
        return out

    # -- the reference wrapper (imcui/hloc/matchers/eloftr.py:68-104) --------------------------------------------------
    @torch.no_grad()
    def __call__(self, data: dict, return_intermediates=False) -> dict:
        pred = self.net(data["image1"].float().cpu(), data["image0"].float().cpu(), return_intermediates)
        scores = pred["confidence"]
        top_k = self.conf["max_keypoints"]
        if top_k is not None and len(scores) > top_k:  # as written upstream: -1 drops the weakest match
            keep = torch.argsort(scores, descending=True)[:top_k]
            pred["keypoints0"], pred["keypoints1"] = pred["keypoints0"][keep], pred["keypoints1"][keep]
            scores = scores[keep]
        pred["keypoints0"], pred["keypoints1"] = pred["keypoints1"], pred["keypoints0"]
        pred["scores"] = scores
        del pred["confidence"]
        return pred
