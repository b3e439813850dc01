"""DUSt3R pair network oracle (torch CPU fp32)  --  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the network that `imcui/hloc/matchers/duster.py:58-74` drives: the wrapper normalises both images with mean = std
= 0.5 (:60-64), builds the two directed pairs (img0, img1), (img1, img0) (`make_pairs(..., symmetrize=True)`, :70-72) and
runs `dust3r.inference.inference(pairs, self.net, device, batch_size=1)` (:73), where `self.net` is
`AsymmetricCroCo3DStereo.from_pretrained("duster_vit_large.pth")` (:37).  `third_party/dust3r` is an un-vendored git
submodule (`.gitmodules`, commit unknown, absent from /root/reference), so the arithmetic below follows the PUBLISHED
model: Wang et al., "DUSt3R: Geometric 3D Vision Made Easy" (CVPR 2024) and Weinzaepfel et al., "CroCo v2" (ICCV 2023)
for the backbone, Ranftl et al., "Vision Transformers for Dense Prediction" (ICCV 2021) for the head:

  * `PatchEmbedDust3R`: 16x16 stride-16 convolution 3 -> E, tokens row-major over the (H/16, W/16) grid, integer (y, x)
    positions; no class token, no learnt position embedding.
  * encoder: `enc_depth` pre-norm ViT blocks (LayerNorm eps 1e-6, fused qkv with bias laid out [3][heads][64], 2-D rotary
    embedding "RoPE100" on q and k, soft-max(q k^T / 8) v, projection, MLP E -> 4E -> E with exact GELU), `enc_norm`.
  * RoPE2D(freq = 100): each 64-wide head is split into a y half and an x half of 32; inside a half,
    inv_freq_i = 100^(-2i/32) (16 frequencies, repeated twice), out = t * cos(p * f) + rotate_half(t) * sin(p * f) with
    rotate_half(t) = [-t[16:], t[:16]] and p the integer row (y half) / column (x half) of the token.
  * decoder: `decoder_embed` E -> D, then `dec_depth` blocks run SYMMETRICALLY: view 1 goes through `dec_blocks`, view 2
    through `dec_blocks2`, and block i of either side reads the OTHER side's tokens as they were BEFORE block i.  A block is
    x += self_attn(norm1(x)); x += cross_attn(norm2(x), norm_y(y), norm_y(y)) (separate q / k / v projections, RoPE on q
    with x's positions and on k with y's); x += mlp(norm3(x)).  `dec_norm` on the last output.
  * head (`head_type='dpt'`, `output_mode='pts3d'`): DPT over the token maps hooked at [encoder output, decoder block
    dec_depth/2, 3 dec_depth/4, dec_depth]: 1x1 projection to (96, 192, 384, 768) channels followed by a x4 / x2
    transposed convolution, nothing, or a 3x3 stride-2 convolution; 3x3 `layer_rn` to 256 channels (no bias); four
    fusion blocks (pre-activation residual units, bilinear x2 with align_corners=True, 1x1 `out_conv`); head = 3x3
    256 -> 128, bilinear x2 (align_corners=True), 3x3 128 -> 128, ReLU, 1x1 128 -> 4.  Post-processing
    (`depth_mode=('exp', -inf, inf)`, `conf_mode=('exp', 1, inf)`): pts3d = xyz / |xyz| * expm1(|xyz|), conf = 1 + exp(c).
  * `forward(view1, view2)` returns `(res1, res2)` with res1 = {pts3d, conf} of view 1 in its own frame and res2 =
    {pts3d_in_other_view, conf} of view 2 in view 1's frame; head 1 reads view 1's tokens, head 2 view 2's.

State-dict names are upstream's (`patch_embed.proj`, `enc_blocks.{i}.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}`,
`enc_norm`, `decoder_embed`, `dec_blocks{,2}.{i}.{norm1,attn.*,norm2,norm_y,cross_attn.{projq,projk,projv,proj},norm3,
mlp.*}`, `dec_norm`, `downstream_head{1,2}.dpt.{act_postprocess.{k}.{0,1},scratch.layer_rn.{k},
scratch.refinenet{1..4}.{resConfUnit{1,2}.conv{1,2},out_conv},head.{0,2,4}}`).

Pinning: PARITY UNPINNED -- the reference holds neither the sources nor golden vectors for this path.  Independent checks
that ARE possible in the container (tests/test_oracle_dust3r.py): the fusion blocks and the reassemble stage against
`transformers.models.dpt.modeling_dpt` (an independent restatement of the same DPT blocks), the encoder block without its
rotation against `transformers`' ViTLayer, the attention block against
`torch.nn.functional.scaled_dot_product_attention`, the rotary embedding against a complex-number formulation.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

DEFAULT_CFG = {  # DUSt3R_ViTLarge_BaseDecoder_512_dpt
    "enc_dim": 1024,
    "enc_depth": 24,
    "dec_dim": 768,
    "dec_depth": 12,
    "patch": 16,
    "rope_freq": 100.0,
}
LAYER_DIMS = (96, 192, 384, 768)
FEATURE_DIM = 256
LAST_DIM = 128
HEAD_DIM = 64
LN_EPS = 1e-6


def rope2d(tokens: torch.Tensor, pos: torch.Tensor, base: float) -> torch.Tensor:
    """tokens [B, heads, N, 64], pos [B, N, 2] integer (y, x)."""
    D = tokens.shape[-1] // 2
    inv_freq = 1.0 / (base ** (torch.arange(0, D, 2).float() / D))

    def rot1d(t, p):
        fr = p[:, None, :, None].float() * inv_freq  # [B, 1, N, D/2]
        fr = torch.cat((fr, fr), -1)
        t1, t2 = t[..., : D // 2], t[..., D // 2 :]
        return t * fr.cos() + torch.cat((-t2, t1), -1) * fr.sin()

    y, x = tokens.chunk(2, -1)
    return torch.cat((rot1d(y, pos[..., 0]), rot1d(x, pos[..., 1])), -1)


class DUSt3ROracle:
    def __init__(self, state_dict: dict, cfg: dict | None = None):
        self.cfg = {**DEFAULT_CFG, **(cfg or {})}
        self.sd = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items() if torch.is_tensor(v)}
        self.hooks = [0, self.cfg["dec_depth"] * 2 // 4, self.cfg["dec_depth"] * 3 // 4, self.cfg["dec_depth"]]

    # -- building blocks -------------------------------------------------------------------------------------------
    def _ln(self, x, p):
        return F.layer_norm(x, (x.shape[-1],), self.sd[p + ".weight"], self.sd[p + ".bias"], LN_EPS)

    def _lin(self, x, p):
        return F.linear(x, self.sd[p + ".weight"], self.sd[p + ".bias"])

    def _heads(self, t):  # [B, N, C] -> [B, heads, N, 64]
        B, N, Cc = t.shape
        return t.view(B, N, Cc // HEAD_DIM, HEAD_DIM).transpose(1, 2)

    def _attend(self, q, k, v):
        a = (q @ k.transpose(-1, -2)) * (HEAD_DIM**-0.5)
        o = a.softmax(-1) @ v
        return o.transpose(1, 2).reshape(o.shape[0], o.shape[2], -1)

    def _self_attn(self, x, pos, p):
        B, N, Cc = x.shape
        qkv = self._lin(x, p + ".qkv").view(B, N, 3, Cc // HEAD_DIM, HEAD_DIM).permute(2, 0, 3, 1, 4)
        q = rope2d(qkv[0], pos, self.cfg["rope_freq"])
        k = rope2d(qkv[1], pos, self.cfg["rope_freq"])
        return self._lin(self._attend(q, k, qkv[2]), p + ".proj")

    def _cross_attn(self, x, y, xpos, ypos, p):
        q = rope2d(self._heads(self._lin(x, p + ".projq")), xpos, self.cfg["rope_freq"])
        k = rope2d(self._heads(self._lin(y, p + ".projk")), ypos, self.cfg["rope_freq"])
        v = self._heads(self._lin(y, p + ".projv"))
        return self._lin(self._attend(q, k, v), p + ".proj")

    def _mlp(self, x, p):
        return self._lin(F.gelu(self._lin(x, p + ".fc1")), p + ".fc2")

    def _enc_block(self, x, pos, p):
        x = x + self._self_attn(self._ln(x, p + ".norm1"), pos, p + ".attn")
        return x + self._mlp(self._ln(x, p + ".norm2"), p + ".mlp")

    def _dec_block(self, x, y, xpos, ypos, p):
        x = x + self._self_attn(self._ln(x, p + ".norm1"), xpos, p + ".attn")
        y_ = self._ln(y, p + ".norm_y")
        x = x + self._cross_attn(self._ln(x, p + ".norm2"), y_, xpos, ypos, p + ".cross_attn")
        return x + self._mlp(self._ln(x, p + ".norm3"), p + ".mlp")

    # -- encoder ---------------------------------------------------------------------------------------------------
    def positions(self, B, h, w):
        yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        return torch.stack((yy, xx), -1).view(1, h * w, 2).expand(B, -1, -1)

    def patch_embed(self, img):
        """img [B, 3, H, W], already normalised -> tokens [B, T, E]."""
        x = F.conv2d(img, self.sd["patch_embed.proj.weight"], self.sd["patch_embed.proj.bias"], stride=self.cfg["patch"])
        return x.flatten(2).transpose(1, 2)

    def encode(self, img, return_layers=False):
        B, _, H, W = img.shape
        pos = self.positions(B, H // self.cfg["patch"], W // self.cfg["patch"])
        x = self.patch_embed(img)
        layers = [x]
        for i in range(self.cfg["enc_depth"]):
            x = self._enc_block(x, pos, f"enc_blocks.{i}")
            layers.append(x)
        x = self._ln(x, "enc_norm")
        return (x, pos, layers) if return_layers else (x, pos)

    # -- decoder ---------------------------------------------------------------------------------------------------
    def decode(self, f1, pos1, f2, pos2):
        """-> two lists (view 1, view 2) of dec_depth + 1 token maps: [encoder output, block 1, ..., block dec_depth (normed)]."""
        out = [(f1, f2)]
        g1, g2 = self._lin(f1, "decoder_embed"), self._lin(f2, "decoder_embed")
        cur = (g1, g2)
        for i in range(self.cfg["dec_depth"]):
            n1 = self._dec_block(cur[0], cur[1], pos1, pos2, f"dec_blocks.{i}")
            n2 = self._dec_block(cur[1], cur[0], pos2, pos1, f"dec_blocks2.{i}")
            cur = (n1, n2)
            out.append(cur)
        out[-1] = (self._ln(out[-1][0], "dec_norm"), self._ln(out[-1][1], "dec_norm"))
        return [o[0] for o in out], [o[1] for o in out]

    # -- DPT head --------------------------------------------------------------------------------------------------
    def _conv(self, x, p, stride=1, padding=0):
        return F.conv2d(x, self.sd[p + ".weight"], self.sd.get(p + ".bias"), stride, padding)

    def _rcu(self, x, p):
        out = self._conv(F.relu(x), p + ".conv1", 1, 1)
        out = self._conv(F.relu(out), p + ".conv2", 1, 1)
        return out + x

    def _fusion(self, p, x, skip=None):
        if skip is not None:
            x = x + self._rcu(skip, p + ".resConfUnit1")
        x = self._rcu(x, p + ".resConfUnit2")
        x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        return self._conv(x, p + ".out_conv")

    def reassemble(self, tokens, head, h, w):
        """hooked token maps -> the four `layer_rn` outputs [B, 256, .., ..] at 1/4, 1/8, 1/16, 1/32."""
        p = f"downstream_head{head}.dpt."
        sd = self.sd
        maps = [tokens[k].transpose(1, 2).reshape(tokens[k].shape[0], -1, h, w) for k in self.hooks]
        a = []
        for k, m in enumerate(maps):
            m = self._conv(m, f"{p}act_postprocess.{k}.0")
            if k == 0:
                m = F.conv_transpose2d(m, sd[f"{p}act_postprocess.0.1.weight"], sd[f"{p}act_postprocess.0.1.bias"], stride=4)
            elif k == 1:
                m = F.conv_transpose2d(m, sd[f"{p}act_postprocess.1.1.weight"], sd[f"{p}act_postprocess.1.1.bias"], stride=2)
            elif k == 3:
                m = self._conv(m, f"{p}act_postprocess.3.1", 2, 1)
            a.append(m)
        return [self._conv(m, f"{p}scratch.layer_rn.{k}", 1, 1) for k, m in enumerate(a)]

    def head(self, tokens, head, H, W, return_intermediates=False):
        p = f"downstream_head{head}.dpt."
        h, w = H // self.cfg["patch"], W // self.cfg["patch"]
        layers = self.reassemble(tokens, head, h, w)
        path4 = self._fusion(p + "scratch.refinenet4", layers[3])[:, :, : layers[2].shape[2], : layers[2].shape[3]]
        path3 = self._fusion(p + "scratch.refinenet3", path4, layers[2])
        path2 = self._fusion(p + "scratch.refinenet2", path3, layers[1])
        path1 = self._fusion(p + "scratch.refinenet1", path2, layers[0])
        o = self._conv(path1, p + "head.0", 1, 1)
        o = F.interpolate(o, scale_factor=2, mode="bilinear", align_corners=True)
        feat = F.relu(self._conv(o, p + "head.2", 1, 1))
        o = self._conv(feat, p + "head.4")
        fmap = o.permute(0, 2, 3, 1)
        xyz = fmap[..., :3]
        d = xyz.norm(dim=-1, keepdim=True)
        pts = xyz / d.clip(min=1e-8) * torch.expm1(d)
        conf = 1.0 + fmap[..., 3].exp()
        res = {"pts3d": pts, "conf": conf}
        if return_intermediates:
            res.update(_layers=layers, _path4=path4, _path3=path3, _path2=path2, _path1=path1, _feat=feat, _raw=fmap)
        return res

    # -- the network and the wrapper's driver ------------------------------------------------------------------------
    def forward(self, img1, img2, return_intermediates=False):
        """`AsymmetricCroCo3DStereo.forward` on normalised images [B, 3, H1, W1], [B, 3, H2, W2] -> (res1, res2).  Views of one size
        are encoded as one batch, views of different sizes one after the other (upstream's `_encode_image_pairs`; per-sample the
        same arithmetic); the decoder cross-attends token sequences of different lengths as they are, each head works at its
        own view's size."""
        B, _, H1, W1 = img1.shape
        H2, W2 = img2.shape[-2:]
        if (H1, W1) == (H2, W2):
            both, pos, enc_layers = self.encode(torch.cat((img1, img2), 0), return_layers=True)
            f1, f2 = both[:B], both[B:]
            pos1, pos2 = pos[:B], pos[B:]
            enc1, enc2 = [t[:B] for t in enc_layers], [t[B:] for t in enc_layers]
        else:
            f1, pos1, enc1 = self.encode(img1, return_layers=True)
            f2, pos2, enc2 = self.encode(img2, return_layers=True)
        dec1, dec2 = self.decode(f1, pos1, f2, pos2)
        res1 = self.head(dec1, 1, H1, W1, return_intermediates)
        res2 = self.head(dec2, 2, H2, W2, return_intermediates)
        res2["pts3d_in_other_view"] = res2.pop("pts3d")
        if return_intermediates:
            res1["_enc_layers"], res2["_enc_layers"] = enc1, enc2
            res1["_dec"], res2["_dec"] = dec1, dec2
        return res1, res2

    def inference_symmetrized(self, image0, image1, return_intermediates=False):
        """What duster.py:60-73 hands to the global aligner: images in [0, 1], the two directed pairs collated
        -> {'pred1': {pts3d, conf}, 'pred2': {pts3d_in_other_view, conf}} with the batch entries in upstream's order: `make_pairs(images,
        scene_graph="complete", symmetrize=True)` lists (image1, image0) first (`for i in range(n): for j in range(i)`), then the
        swapped pairs, so entry 0 = (image1 as view 1, image0 as view 2) and entry 1 = (image0, image1) -- the entry mast3r.py:61-64 reads
        its descriptors from (`pred1["desc"][1]` = image0's, `pred2["desc"][1]` = image1's).  Images of one size: the entries are
        concatenated ([2, H, W, ...] tensors); of two sizes: upstream's `inference` collates with `lists=True`, every value is a
        LIST of the two per-entry maps ([H, W, ...], the batch axis dropped) -- indexing `[1]` reads the same entry either way."""
        n0, n1 = (image0 - 0.5) / 0.5, (image1 - 0.5) / 0.5
        out = []
        for a, b in ((n1, n0), (n0, n1)):  # batch_size = 1: one forward per directed pair, in make_pairs' order
            out.append(self.forward(a, b, return_intermediates))
        same = image0.shape[-2:] == image1.shape[-2:]
        collate = (lambda maps: torch.cat(maps, 0)) if same else (lambda maps: [m[0] for m in maps])
        keys1 = ("pts3d", "conf")
        keys2 = ("pts3d_in_other_view", "conf")
        pred1 = {k: collate([o[0][k] for o in out]) for k in keys1}
        pred2 = {k: collate([o[1][k] for o in out]) for k in keys2}
        res = {"pred1": pred1, "pred2": pred2}
        if return_intermediates:
            res["_passes"] = out
        return res


class MASt3ROracle(DUSt3ROracle):
    """`AsymmetricMASt3R` (imcui/hloc/matchers/mast3r.py:41; `third_party/mast3r`, un-vendored): the DUSt3R network with the
    'catmlp+dpt' head (Leroy et al., "Grounding Image Matching in 3D with MASt3R", ECCV 2024): next to the DPT point-map head an
    MLP (`head_local_features`, hidden width 4 (E + D), GELU) reads the concatenation [encoder tokens | last decoder tokens] and
    emits (desc_dim + 1) x 16 x 16 values per token; `F.pixel_shuffle(., 16)` spreads them over the token's pixels; the first
    desc_dim channels, L2-normalised (`desc_mode='norm'`), are the local descriptors, the last is the logit of their confidence
    (`desc_conf_mode=('exp', 0, inf)`: exp(x)).  The wrapper matches `pred1["desc"][1]` with `pred2["desc"][1]` (mast3r.py:61-75).
    PARITY UNPINNED like the base class."""

    def __init__(self, state_dict: dict, cfg: dict | None = None):
        super().__init__(state_dict, cfg)
        self.desc_dim = self.sd["downstream_head1.head_local_features.fc2.weight"].shape[0] // 256 - 1

    def head(self, tokens, head, H, W, return_intermediates=False):
        res = super().head(tokens, head, H, W, return_intermediates)
        p = f"downstream_head{head}.head_local_features"
        cat = torch.cat((tokens[0], tokens[-1]), -1)
        lf = self._lin(F.gelu(self._lin(cat, p + ".fc1")), p + ".fc2")  # [B, T, (dd + 1) * 256]
        B = lf.shape[0]
        lf = lf.transpose(-1, -2).reshape(B, -1, H // self.cfg["patch"], W // self.cfg["patch"])
        fmap = F.pixel_shuffle(lf, self.cfg["patch"]).permute(0, 2, 3, 1)  # [B, H, W, dd + 1]
        d = fmap[..., : self.desc_dim]
        res["desc"] = d / d.norm(dim=-1, keepdim=True)
        res["desc_conf"] = fmap[..., self.desc_dim].exp()
        return res

    def inference_symmetrized(self, image0, image1, return_intermediates=False):
        res = super().inference_symmetrized(image0, image1, return_intermediates=True)
        out = res["_passes"]
        same = image0.shape[-2:] == image1.shape[-2:]
        collate = (lambda maps: torch.cat(maps, 0)) if same else (lambda maps: [m[0] for m in maps])
        for k, pred in ((0, "pred1"), (1, "pred2")):
            res[pred]["desc"] = collate([o[k]["desc"] for o in out])
            res[pred]["desc_conf"] = collate([o[k]["desc_conf"] for o in out])
        if not return_intermediates:
            res.pop("_passes")
        return res


def nn_dot_first_argmax(queries: torch.Tensor, db: torch.Tensor, block: int = 2**13) -> torch.Tensor:
    """`cdistMatcher(db, dist='dot').query(queries)` of upstream's mast3r/fast_nn.py: per query the index of the largest dot product,
    the first one on ties (blocks of `block` x `block` similarities, a later block only wins with a strictly larger value)."""
    best = torch.full((len(queries),), -float("inf"))
    arg = torch.full((len(queries),), -1, dtype=torch.int64)
    for i in range(0, len(queries), block):
        for j in range(0, len(db), block):
            v, k = (queries[i : i + block] @ db[j : j + block].T).max(dim=1)
            upd = v > best[i : i + block]
            best[i : i + block][upd] = v[upd]
            arg[i : i + block][upd] = k[upd] + j
    return arg


def fast_reciprocal_nns(desc1: torch.Tensor, desc2: torch.Tensor, subsample: int = 2, max_iter: int = 10):
    """`fast_reciprocal_NNs(desc1, desc2, subsample_or_initxy1=S, ret_xy=True, pixel_tol=0, dist='dot')` as called at
    imcui/hloc/matchers/mast3r.py:68-75 (upstream mast3r/fast_nn.py, restated with numpy bookkeeping as upstream has it): chains
    start on the grid S//2::S of image 1 and alternate nearest-neighbour hops image 1 -> 2 -> 1; a chain is done when a hop returns
    to where the previous round left it; after at most `max_iter` rounds the closed chains give the matches, de-duplicated and
    sorted by (y1 * W1 + x1, y2 * W2 + x2).  Returns (xy1 [K,2], xy2 [K,2]) int64 (x, y)."""
    import numpy as np

    H1, W1, D = desc1.shape
    H2, W2, _ = desc2.shape
    pts1, pts2 = desc1.reshape(-1, D), desc2.reshape(-1, D)
    y1, x1 = np.mgrid[subsample // 2 : H1 : subsample, subsample // 2 : W1 : subsample].reshape(2, -1)
    xy1 = np.int32(np.unique(x1 + W1 * y1))
    xy2 = np.full_like(xy1, -1)
    old_xy1, old_xy2 = xy1.copy(), xy2.copy()
    notyet = np.ones(len(xy1), dtype=bool)
    niter = 0
    while notyet.any():
        xy2[notyet] = nn_dot_first_argmax(pts1[torch.from_numpy(xy1[notyet]).long()], pts2).numpy()
        notyet &= old_xy2 != xy2
        if notyet.any():
            xy1[notyet] = nn_dot_first_argmax(pts2[torch.from_numpy(xy2[notyet]).long()], pts1).numpy()
        notyet &= old_xy1 != xy1
        niter += 1
        if niter >= max_iter:
            break
        old_xy2[:] = xy2
        old_xy1[:] = xy1
    conv = ~notyet
    corres = np.unique(np.c_[xy2[conv], xy1[conv]].view(np.int64))  # low word xy2, high word xy1: sorted by xy1, then xy2
    b, a = corres[:, None].view(np.int32).T
    a, b = a.astype(np.int64), b.astype(np.int64)
    return torch.from_numpy(np.c_[a % W1, a // W1]), torch.from_numpy(np.c_[b % W2, b // W2])


def duster_matches_from_scene(imgs, masks, pts3d, max_keypoints=3000):
    """The wrapper's steps after `global_aligner` (imcui/hloc/matchers/duster.py:76-108), restated with brute-force numpy:
    confidence masks select points (:80-86: `xy_grid(W, H)[conf_i]`, `pts3d[i][conf_i]`), `find_reciprocal_matches(P1, P2)`
    (upstream dust3r/utils/geometry.py: `nn1_in_P2 = tree2.query(P1)`, `nn2_in_P1 = tree1.query(P2)`, a point j of P2 is kept
    when `nn1_in_P2[nn2_in_P1[j]] == j`), `mkpts1 = pts2d[1][reciprocal_in_P2]`, `mkpts0 = pts2d[0][nn2_in_P1][reciprocal_in_P2]`
    (:97-98), then `np.round(np.linspace(0, n - 1, top_k))` (:100-103).  Nearest neighbours are exhaustive float64 distance
    arg-mins (first index on ties), so the KD-trees of the product are checked by an independent search.  Test infrastructure."""
    import numpy as np

    px, pts = [], []
    for im, m, p in zip(imgs, masks, pts3d):
        m = np.asarray(m, dtype=bool)
        H, W = im.shape[:2]
        grid = np.zeros((H, W, 2), dtype=np.int32)
        for y in range(H):
            for x in range(W):
                grid[y, x] = (x, y)
        px.append(grid[m])
        pts.append(np.asarray(p, dtype=np.float64)[m])
    if len(pts[1]) == 0:
        return np.zeros((0, 2), dtype=np.int32), np.zeros((0, 2), dtype=np.int32)

    def nearest(q, db):
        out = np.zeros(len(q), dtype=np.int64)
        for i in range(len(q)):
            out[i] = np.argmin(((db - q[i]) ** 2).sum(1))
        return out

    nn1_in_p2 = nearest(pts[0], pts[1]) if len(pts[0]) else np.zeros(0, dtype=np.int64)
    nn2_in_p1 = nearest(pts[1], pts[0]) if len(pts[0]) else np.zeros(len(pts[1]), dtype=np.int64)
    keep2 = np.array([len(pts[0]) > 0 and nn1_in_p2[nn2_in_p1[j]] == j for j in range(len(pts[1]))], dtype=bool)
    k1 = px[1][keep2]
    k0 = px[0][nn2_in_p1][keep2] if len(pts[0]) else np.zeros((0, 2), dtype=np.int32)
    if max_keypoints is not None and len(k0) > max_keypoints:
        sel = np.round(np.linspace(0, len(k0) - 1, max_keypoints)).astype(int)
        k0, k1 = k0[sel], k1[sel]
    return k0, k1


def num_params(cfg: dict | None = None) -> int:
    c = {**DEFAULT_CFG, **(cfg or {})}
    E, D = c["enc_dim"], c["dec_dim"]
    enc = c["enc_depth"] * (4 * E * E + 8 * E * E)
    dec = 2 * c["dec_depth"] * (8 * D * D + 8 * D * D)
    return enc + dec + E * D + 3 * 256 * E
