"""Seeded synthetic weights in the upstream state-dict layouts.

No checkpoint exists offline (weights come from the HF hub at `_init` time,
imcui/hloc/extractors/superpoint.py:48-53, imcui/hloc/matchers/lightglue.py:39-51), so benches, the smoke test and
the parity tests load the SAME seeded tensors into the HIP backend and -- in the tests -- into the oracle.  Key names /
shapes follow SURVEY.md Appendix A.1 / A.2, so a real `superpoint_v1.pth` / `superpoint_lightglue.pth` /
`superglue_outdoor.pth` state dict drops in unchanged.  This module is data generation only (like synth.py): it holds
no model arithmetic and imports nothing from `oracle/`.
"""
from __future__ import annotations

import math

import torch

SP_LAYERS = [
    # name, cout, cin, k
    ("conv1a", 64, 1, 3),
    ("conv1b", 64, 64, 3),
    ("conv2a", 64, 64, 3),
    ("conv2b", 64, 64, 3),
    ("conv3a", 128, 64, 3),
    ("conv3b", 128, 128, 3),
    ("conv4a", 128, 128, 3),
    ("conv4b", 128, 128, 3),
    ("convPa", 256, 128, 3),
    ("convPb", 65, 256, 1),
    ("convDa", 256, 128, 3),
    ("convDb", 256, 256, 1),
]


def superpoint_state_dict(seed: int = 0, peaky: bool = True) -> dict:
    """Kaiming-scaled random SuperPoint weights (1 300 865 params).

    `peaky=True` scales the detector logits (convPb) up and biases the dustbin
    channel so the soft-max heat-map has many well separated peaks above the
    0.005 threshold -- random heads otherwise give a near-uniform 1/65 map and
    NMS / top-k are exercised degenerately (golden data (i)).
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, cout, cin, k in SP_LAYERS:
        fan_in = cin * k * k
        w = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / fan_in)
        b = torch.randn(cout, generator=g) * 0.05
        sd[f"{name}.weight"] = w
        sd[f"{name}.bias"] = b
    if peaky:
        sd["convPb.weight"] = sd["convPb.weight"] * 6.0
        sd["convPb.bias"][-1] += 2.0
        # ReLU features have a large positive mean; centre the descriptor projection over
        # its inputs so descriptors are not dominated by one common direction.
        w = sd["convDb.weight"]
        sd["convDb.weight"] = (w - w.mean(dim=1, keepdim=True)) * 3.0
        sd["convDb.bias"] = -_mean_descriptor_logits(sd, g)
    return sd


def _mean_descriptor_logits(sd: dict, g: torch.Generator) -> torch.Tensor:
    """Mean convDb pre-activation over a small noise image (used to centre descriptors)."""
    import torch.nn.functional as F

    x = torch.rand(1, 1, 96, 128, generator=g)
    with torch.no_grad():
        for name in ("conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convDa"):
            x = F.relu(F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=1))
            if name in ("conv1b", "conv2b", "conv3b"):
                x = F.max_pool2d(x, 2, 2)
        y = F.conv2d(x, sd["convDb.weight"], None)
    return y.mean(dim=(0, 2, 3))


def lightglue_state_dict(seed: int = 0, n_layers: int = 9, dim: int = 256, heads: int = 4, structured: bool = True,
                         damp: float | None = None, ln_noise: float | None = None, final_gain: float | None = None,
                         input_dim: int = 256, add_scale_ori: bool = False) -> dict:
    """Random LightGlue weights in the upstream (new-style) key layout.

    transformers.{i}.self_attn.{Wqkv,out_proj,ffn.0,ffn.1,ffn.3}
    transformers.{i}.cross_attn.{to_qk,to_v,to_out,ffn.0,ffn.1,ffn.3}
    log_assignment.{i}.{matchability,final_proj}, token_confidence.{i}.token.0,
    posenc.Wr.weight

    `structured=True` shapes the heads so the data-dependent control flow is
    exercised non-degenerately with random transformer weights: residual updates are
    damped, `final_proj` is a scaled identity + noise (true correspondences of
    SuperPoint descriptors then win the dual soft-max), matchability logits spread
    around +1 (a few points fall below the 1 - width_confidence prune threshold) and
    the token-confidence bias rises with depth (pairs early-stop at varying layers).

    `damp` / `ln_noise` override the residual damping (default 0.03 structured, 1.0 otherwise) and the spread of
    the LayerNorm gamma / beta (default 0 structured, 0.1 otherwise): `structured=True, damp=1.0, ln_noise=0.1`
    keeps the shaped heads but makes every layer a full-strength update, so an error in the attention / FFN
    arithmetic reaches the outputs undiminished (the "undamped" parity tests); `final_gain` rescales the identity part
    of `final_proj` (the residual stream grows with strong updates, and similarities far above ~100 turn fp32 round-off
    into score differences no two fp32 implementations can agree on to 1e-4).
    """
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f, scale=1.0, bias=True, prefix=""):
        bound = scale / math.sqrt(in_f)
        d = {prefix + ".weight": (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound * math.sqrt(3.0)}
        if bias:
            d[prefix + ".bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * 0.1
        return d

    sd = {}
    head_dim = dim // heads
    # gamma = 1 -> std 1 upstream (nn.init.normal_(std=gamma**-2))
    sd["posenc.Wr.weight"] = torch.randn(head_dim // 2, 2, generator=g)
    damp = (0.03 if structured else 1.0) if damp is None else damp
    ln_noise = (0.0 if structured else 0.1) if ln_noise is None else ln_noise
    for i in range(n_layers):
        p = f"transformers.{i}."
        sd.update(lin(3 * dim, dim, scale=10.0 if structured else 1.0, prefix=p + "self_attn.Wqkv"))
        sd.update(lin(dim, dim, prefix=p + "self_attn.out_proj"))
        sd.update(lin(2 * dim, 2 * dim, prefix=p + "self_attn.ffn.0"))
        sd[p + "self_attn.ffn.1.weight"] = 1.0 + ln_noise * torch.randn(2 * dim, generator=g)
        sd[p + "self_attn.ffn.1.bias"] = ln_noise * torch.randn(2 * dim, generator=g)
        sd.update(lin(dim, 2 * dim, scale=damp, prefix=p + "self_attn.ffn.3"))
        if structured:  # zero-mean rows: no common-mode drift of the residual stream
            w = sd[p + "self_attn.ffn.3.weight"]
            sd[p + "self_attn.ffn.3.weight"] = w - w.mean(dim=1, keepdim=True)
            sd[p + "self_attn.ffn.3.bias"] = sd[p + "self_attn.ffn.3.bias"] * 0.05
        sd.update(lin(dim, dim, scale=10.0 if structured else 1.0, prefix=p + "cross_attn.to_qk"))
        sd.update(lin(dim, dim, prefix=p + "cross_attn.to_v"))
        sd.update(lin(dim, dim, prefix=p + "cross_attn.to_out"))
        sd.update(lin(2 * dim, 2 * dim, prefix=p + "cross_attn.ffn.0"))
        sd[p + "cross_attn.ffn.1.weight"] = 1.0 + ln_noise * torch.randn(2 * dim, generator=g)
        sd[p + "cross_attn.ffn.1.bias"] = ln_noise * torch.randn(2 * dim, generator=g)
        sd.update(lin(dim, 2 * dim, scale=damp, prefix=p + "cross_attn.ffn.3"))
        if structured:
            w = sd[p + "cross_attn.ffn.3.weight"]
            sd[p + "cross_attn.ffn.3.weight"] = w - w.mean(dim=1, keepdim=True)
            sd[p + "cross_attn.ffn.3.bias"] = sd[p + "cross_attn.ffn.3.bias"] * 0.05
        q = f"log_assignment.{i}."
        if structured:
            sd.update(lin(1, dim, scale=48.0, prefix=q + "matchability"))  # logit std ~2.5 per unit |x|
            sd[q + "matchability.bias"] = sd[q + "matchability.bias"] + 2.0
            sd.update(lin(dim, dim, scale=2.0, prefix=q + "final_proj"))
            sd[q + "final_proj.weight"] = sd[q + "final_proj.weight"] + (4.0 * math.sqrt(60.0) if final_gain is None else final_gain) * torch.eye(dim)
        else:
            sd.update(lin(1, dim, prefix=q + "matchability"))
            sd.update(lin(dim, dim, scale=4.0, prefix=q + "final_proj"))
        if i < n_layers - 1:
            t = f"token_confidence.{i}.token.0"
            if structured:
                sd.update(lin(1, dim, scale=12.0, prefix=t))  # logit std ~1 per unit |x|
                sd[t + ".bias"] = sd[t + ".bias"] + 1.2 + 0.4 * i
            else:
                sd.update(lin(1, dim, scale=2.0, prefix=t))
    # variants of upstream's `features` table, drawn last so that the 256-d superpoint tensors above do not change
    if add_scale_ori:
        sd["posenc.Wr.weight"] = torch.cat([sd["posenc.Wr.weight"], 0.3 * torch.randn(head_dim // 2, 2, generator=g)], 1)
    if input_dim != dim:
        # disk / aliked / sift: Linear(input_dim, 256); a scaled partial isometry + noise keeps distinctive descriptors distinctive
        q, _ = torch.linalg.qr(torch.randn(dim, input_dim, generator=g))
        sd["input_proj.weight"] = q + 0.02 * torch.randn(dim, input_dim, generator=g)
        sd["input_proj.bias"] = 0.01 * torch.randn(dim, generator=g)
    return sd


def loftr_state_dict(seed: int = 0, structured: bool = True) -> dict:
    """Random LoFTR weights in kornia's state-dict layout (ResNetFPN_8_2 + coarse/fine transformers).

    backbone.{conv1,bn1,layer{1,2,3}.{0,1}.{conv1,bn1,conv2,bn2[,downsample.{0,1}]},layer3_outconv,
    layer2_outconv,layer2_outconv2.{0,1,3},layer1_outconv,layer1_outconv2.{0,1,3}},
    loftr_coarse.layers.{0..7}.{q_proj,k_proj,v_proj,merge,mlp.0,mlp.2,norm1,norm2},
    fine_preprocess.{down_proj,merge_feat}, loftr_fine.layers.{0,1}.*
    `structured` damps the transformer updates so coarse features stay image-dependent and the
    dual soft-max produces confident mutual matches with random weights.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, k, gain=1.0):
        sd[name + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * math.sqrt(2.0 / (cin * k * k)) * gain

    def bn(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g)
        sd[name + ".num_batches_tracked"] = torch.tensor(0)

    b = "backbone."
    conv(b + "conv1", 128, 1, 7)
    bn(b + "bn1", 128)
    dims = [128, 128, 196, 256]
    for li in range(1, 4):
        cin, cout = dims[li - 1], dims[li]
        for bi in range(2):
            p = f"{b}layer{li}.{bi}"
            c_in = cin if bi == 0 else cout
            conv(p + ".conv1", cout, c_in, 3)
            bn(p + ".bn1", cout)
            conv(p + ".conv2", cout, cout, 3, gain=0.5)
            bn(p + ".bn2", cout)
            if bi == 0 and li > 1:
                conv(p + ".downsample.0", cout, c_in, 1)
                bn(p + ".downsample.1", cout)
    conv(b + "layer3_outconv", 256, 256, 1)
    if structured:
        # x3 is post-ReLU and dominated by one common direction: project the mean feature of a small
        # calibration image out of every row, then apply a gain, so that the coarse dual soft-max is
        # driven by image content and produces confident mutual matches with random weights
        m = _loftr_mean_stage3_feature(sd, torch.rand(1, 1, 96, 128, generator=g))
        m = m / m.norm()
        wc = sd[b + "layer3_outconv.weight"][:, :, 0, 0]
        wc = wc - (wc @ m)[:, None] * m[None, :]
        sd[b + "layer3_outconv.weight"] = (wc * 4.0)[:, :, None, None].contiguous()
    conv(b + "layer2_outconv", 256, 196, 1)
    conv(b + "layer2_outconv2.0", 256, 256, 3)
    bn(b + "layer2_outconv2.1", 256)
    conv(b + "layer2_outconv2.3", 196, 256, 3)
    conv(b + "layer1_outconv", 196, 128, 1)
    conv(b + "layer1_outconv2.0", 196, 196, 3)
    bn(b + "layer1_outconv2.1", 196)
    conv(b + "layer1_outconv2.3", 128, 196, 3)

    def lin(name, out_f, in_f, gain=1.0, bias=False):
        sd[name + ".weight"] = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * math.sqrt(3.0 / in_f) * gain
        if bias:
            sd[name + ".bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * 0.1

    def encoder(prefix, n_layers, d):
        for i in range(n_layers):
            p = f"{prefix}.layers.{i}"
            for nm in ("q_proj", "k_proj", "v_proj", "merge"):
                lin(f"{p}.{nm}", d, d)
            lin(f"{p}.mlp.0", 2 * d, 2 * d)
            lin(f"{p}.mlp.2", d, 2 * d)
            for nm in ("norm1", "norm2"):
                sd[f"{p}.{nm}.weight"] = (0.25 if (structured and nm == "norm2") else 1.0) + 0.05 * torch.randn(d, generator=g)
                sd[f"{p}.{nm}.bias"] = 0.02 * torch.randn(d, generator=g)

    encoder("loftr_coarse", 8, 256)
    lin("fine_preprocess.down_proj", 128, 256, bias=True)
    lin("fine_preprocess.merge_feat", 128, 256, bias=True)
    encoder("loftr_fine", 2, 128)
    return sd


def _loftr_mean_stage3_feature(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """Mean 1/8-resolution ResNet feature of a small calibration image (plain torch; weight shaping only, like
    `_mean_descriptor_logits` for SuperPoint).  Stages = conv7x7/2 + BN + ReLU, then three pairs of BasicBlocks."""
    import torch.nn.functional as F

    def bn(t, p):
        return F.batch_norm(t, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)

    def block(t, p, stride):
        y = F.relu(bn(F.conv2d(t, sd[p + ".conv1.weight"], None, stride, 1), p + ".bn1"))
        y = bn(F.conv2d(y, sd[p + ".conv2.weight"], None, 1, 1), p + ".bn2")
        if p + ".downsample.0.weight" in sd:
            t = bn(F.conv2d(t, sd[p + ".downsample.0.weight"], None, stride, 0), p + ".downsample.1")
        return F.relu(t + y)

    b = "backbone."
    with torch.no_grad():
        t = F.relu(bn(F.conv2d(x, sd[b + "conv1.weight"], None, 2, 3), b + "bn1"))
        for li, stride in ((1, 1), (2, 2), (3, 2)):
            t = block(block(t, f"{b}layer{li}.0", stride), f"{b}layer{li}.1", 1)
    return t.mean(dim=(0, 2, 3))


def superglue_state_dict(seed: int = 0, structured: bool = True) -> dict:
    """Random SuperGlue weights in the upstream (magicleap / Vincentqyw fork) state-dict layout:

    kenc.encoder.{0,3,6,9,12}.{weight[out,in,1],bias}, kenc.encoder.{1,4,7,10}.{weight,bias,running_mean,
    running_var,num_batches_tracked}, gnn.layers.{i}.attn.{merge,proj.0,proj.1,proj.2}.{weight,bias},
    gnn.layers.{i}.mlp.{0,3}.{weight,bias}, gnn.layers.{i}.mlp.1.<BatchNorm>, final_proj.{weight,bias}, bin_score.

    `structured` damps the residual updates and makes `final_proj` a scaled identity + noise, so the optimal
    transport of random-weight features still resolves the true correspondences of distinctive descriptors
    (mutual matches above `match_threshold`, the rest in the dust-bins).
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, scale=1.0, bias_scale=0.1):
        bound = scale * math.sqrt(3.0 / cin)
        sd[name + ".weight"] = (torch.rand(cout, cin, 1, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bias_scale

    def bn(name, c):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g)
        sd[name + ".num_batches_tracked"] = torch.tensor(0)

    chans = [3, 32, 64, 128, 256, 256]
    for i in range(1, len(chans)):
        last = i == len(chans) - 1
        conv(f"kenc.encoder.{3 * (i - 1)}", chans[i], chans[i - 1], scale=(0.05 if structured else 1.0) if last else 1.4)
        if last:
            sd[f"kenc.encoder.{3 * (i - 1)}.bias"].zero_()  # nn.init.constant_(encoder[-1].bias, 0)
            if structured:  # ReLU features have a positive mean: zero-mean rows avoid a common-mode offset
                w = sd[f"kenc.encoder.{3 * (i - 1)}.weight"]
                sd[f"kenc.encoder.{3 * (i - 1)}.weight"] = w - w.mean(dim=1, keepdim=True)
        else:
            bn(f"kenc.encoder.{3 * (i - 1) + 1}", chans[i])
    for i in range(18):
        p = f"gnn.layers.{i}."
        conv(p + "attn.merge", 256, 256)
        for j in range(3):
            conv(p + f"attn.proj.{j}", 256, 256, scale=6.0 if (structured and j < 2) else 1.0)
        conv(p + "mlp.0", 512, 512, scale=1.4)
        bn(p + "mlp.1", 512)
        conv(p + "mlp.3", 256, 512, scale=0.08 if structured else 1.0)
        if structured:
            w = sd[p + "mlp.3.weight"]
            sd[p + "mlp.3.weight"] = w - w.mean(dim=1, keepdim=True)
        sd[p + "mlp.3.bias"].zero_()  # nn.init.constant_(mlp[-1].bias, 0)
    conv("final_proj", 256, 256, scale=2.0 if structured else 4.0)
    if structured:
        sd["final_proj.weight"] = sd["final_proj.weight"] + 4.0 * math.sqrt(60.0) * torch.eye(256)[:, :, None]
    sd["bin_score"] = torch.tensor(35.0 if structured else 1.0)  # above the best random-column score of an outlier
    return sd


def eloftr_state_dict(seed: int = 0, gain: float = 1.0, shaped: bool = True) -> dict:
    """Random EfficientLoFTR weights in the layout of `transformers.EfficientLoFTRForKeypointMatching.state_dict()`
    (the maintained port of the upstream network the reference wrapper imports, imcui/hloc/matchers/eloftr.py:13-18):

    efficientloftr.backbone.stages.{s}.blocks.{b}.{conv1,conv2}.{conv.weight,norm.*}[, identity.*]   RepVGG 1-64-64-128-256
    efficientloftr.local_feature_transformer.layers.{0..3}.{self,cross}_attention.
        {aggregation.{q_aggregation.weight,norm.*}, attention.{q,k,v,o}_proj.weight, mlp.{fc1,fc2}.weight, mlp.layer_norm.*}
    refinement_layer.{out_conv.weight, out_conv_layers.{0,1}.{out_conv1,out_conv2,out_conv3}.weight, batch_norm.*}

    `gain` scales the transformer's residual updates (the LayerNorm that closes each block); `shaped` re-weights the last
    block so that with random weights the coarse features are content codes and the dual soft-max at temperature 0.1
    produces confident mutual matches on overlapping images.
    """
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, k, s=1.0, groups=1):
        sd[name] = torch.randn(cout, cin // groups, k, k, generator=g) * math.sqrt(2.0 / (cin // groups * k * k)) * s

    def bn(name, c, wmean=1.0):
        sd[name + ".weight"] = wmean * (1.0 + 0.1 * torch.randn(c, generator=g))
        sd[name + ".bias"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_mean"] = 0.1 * torch.randn(c, generator=g)
        sd[name + ".running_var"] = 1.0 + 0.2 * torch.rand(c, generator=g)
        sd[name + ".num_batches_tracked"] = torch.tensor(0)

    def lin(name, out_f, in_f, s=1.0):
        sd[name] = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * math.sqrt(3.0 / in_f) * s

    def ln(name, c, wmean=1.0):
        sd[name + ".weight"] = wmean * (1.0 + 0.05 * torch.randn(c, generator=g))
        sd[name + ".bias"] = 0.02 * torch.randn(c, generator=g)

    blocks, strides, dims = [1, 2, 4, 14], [2, 1, 2, 2], [64, 64, 128, 256]
    cin = 1
    for s in range(4):
        for b in range(blocks[s]):
            p = f"efficientloftr.backbone.stages.{s}.blocks.{b}"
            cout = dims[s]
            stride = strides[s] if b == 0 else 1
            has_id = cin == cout and stride == 1
            # three branches are summed before the ReLU: keep the sum at unit variance
            local = shaped and s == 3 and b > 0  # keep the receptive field of the 14-block stage small
            conv(p + ".conv1.conv.weight", cout, cin, 3, 0.3 if local else 0.8)
            bn(p + ".conv1.norm", cout)
            conv(p + ".conv2.conv.weight", cout, cin, 1, 0.7 if local else 0.5)
            bn(p + ".conv2.norm", cout)
            if has_id:
                bn(p + ".identity", cin, 0.6 if local else 0.5)
            cin = cout
    for i in range(4):
        for kind in ("self_attention", "cross_attention"):
            p = f"efficientloftr.local_feature_transformer.layers.{i}.{kind}"
            sd[p + ".aggregation.q_aggregation.weight"] = torch.randn(256, 1, 4, 4, generator=g) * 0.25 + 1.0 / 16
            ln(p + ".aggregation.norm", 256)
            for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
                lin(f"{p}.attention.{nm}.weight", 256, 256, 2.0 if nm in ("q_proj", "k_proj") else 1.0)
            lin(p + ".mlp.fc1.weight", 512, 512)
            lin(p + ".mlp.fc2.weight", 256, 512)
            ln(p + ".mlp.layer_norm", 256, 0.05 if shaped else gain * 0.25)  # shaped: the last block carries the code (below)
    if shaped:
        # The 1/8 features are post-ReLU, share one large common direction and vary smoothly, which makes a few cells
        # the nearest neighbour of everything.  Let the block every token passes through last (layer 3, cross
        # attention) add a large content code: its first linear layer reads the WHITENED deviation of the token from
        # the mean feature (principal components of a calibration image, each scaled to unit variance, then a random
        # mix) and mostly ignores the attention half; the LayerNorm that closes the block applies a big gamma.
        from .synth import _add_blobs, _band_limited_noise

        p = "efficientloftr.local_feature_transformer.layers.3.cross_attention"
        cal = _eloftr_stage3_cells(sd, _add_blobs(g, _band_limited_noise(g, 160, 192), 600))  # [480, 256]
        mean = cal.mean(0)
        _, sv, vt = torch.linalg.svd(cal - mean, full_matrices=False)
        k = 64
        white = vt[:k] / (sv[:k, None] / math.sqrt(cal.shape[0] - 1))  # [k, 256]: z = white @ (x - mean), unit variance
        mix = torch.randn(512, k, generator=g) / math.sqrt(k)
        wx = mix @ white
        w1 = sd[p + ".mlp.fc1.weight"]
        sd[p + ".mlp.fc1.weight"] = torch.cat([wx, w1[:, 256:] * 0.1], 1).contiguous()
        # fc1 has no bias: the mean feature is removed by making every row orthogonal to it
        wx = sd[p + ".mlp.fc1.weight"][:, :256]
        mh = mean / mean.norm()
        sd[p + ".mlp.fc1.weight"][:, :256] = wx - (wx @ mh)[:, None] * mh[None, :]
        sd[p + ".mlp.layer_norm.weight"] = sd[p + ".mlp.layer_norm.weight"] / 0.05 * 3.0
    conv("refinement_layer.out_conv.weight", 256, 256, 1)
    for i, (hid, mid) in enumerate([(128, 256), (64, 128)]):
        p = f"refinement_layer.out_conv_layers.{i}"
        conv(p + ".out_conv1.weight", mid, hid, 1)
        conv(p + ".out_conv2.weight", mid, mid, 3)
        bn(p + ".batch_norm", mid)
        conv(p + ".out_conv3.weight", hid, mid, 3)
    return sd


def _eloftr_stage3_cells(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """1/8-resolution RepVGG features of a small calibration image as [cells, 256] (plain torch; weight shaping only)."""
    import torch.nn.functional as F

    def bn(t, p):
        return F.batch_norm(t, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)

    for s, (nb, stride) in enumerate(zip([1, 2, 4, 14], [2, 1, 2, 2])):
        for b in range(nb):
            p = f"efficientloftr.backbone.stages.{s}.blocks.{b}"
            st = stride if b == 0 else 1
            y = bn(F.conv2d(x, sd[p + ".conv1.conv.weight"], None, st, 1), p + ".conv1.norm")
            y = y + bn(F.conv2d(x, sd[p + ".conv2.conv.weight"], None, st, 0), p + ".conv2.norm")
            if p + ".identity.weight" in sd:
                y = y + bn(x, p + ".identity")
            x = F.relu(y)
    return x[0].permute(1, 2, 0).reshape(-1, x.shape[1])


# ---------------------------------------------------------------------------------------------------------------- DUSt3R
DUST3R_CFG = {"enc_dim": 1024, "enc_depth": 24, "dec_dim": 768, "dec_depth": 12}  # DUSt3R_ViTLarge_BaseDecoder_512_dpt
DUST3R_LAYER_DIMS = (96, 192, 384, 768)


def dust3r_state_dict(seed: int = 0, cfg: dict | None = None, gain: float = 1.0) -> dict:
    """Random weights in the state-dict layout of `AsymmetricCroCo3DStereo` (head_type 'dpt', output_mode 'pts3d';
    imcui/hloc/matchers/duster.py:37 loads `duster_vit_large.pth` into it).  Linear layers are drawn at
    gain / sqrt(fan_in), so every sub-layer moves the residual stream by O(1) (attention logits have unit spread, the
    MLPs are not negligible next to the stream): the parity tests then see each block's arithmetic, which the usual
    std = 0.02 initialisation would hide.  The last 1x1 convolution is small enough that expm1(|xyz|) stays O(1).
    `cfg` overrides enc_dim / enc_depth / dec_dim / dec_depth (dims multiples of 64, dec_depth a multiple of 4); cfg["desc_dim"] > 0
    adds MASt3R's `head_local_features` (`AsymmetricMASt3R`, imcui/hloc/matchers/mast3r.py:41: 24 for the shipped checkpoint)."""
    c = {**DUST3R_CFG, **(cfg or {})}
    E, D = c["enc_dim"], c["dec_dim"]
    g = torch.Generator().manual_seed(seed)
    sd: dict = {}

    def lin(name, n, k, s=1.0, bias=True):
        sd[name + ".weight"] = torch.randn(n, k, generator=g) * (gain * s / math.sqrt(k))
        if bias:
            sd[name + ".bias"] = torch.randn(n, generator=g) * 0.1

    def norm(name, n):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=g)
        sd[name + ".bias"] = 0.05 * torch.randn(n, generator=g)

    def conv(name, co, ci, k, s=1.0, bias=True, transposed=False):
        shape = (ci, co, k, k) if transposed else (co, ci, k, k)
        fan = ci if transposed else ci * k * k  # a transposed convolution with kernel = stride sums over Cin only
        sd[name + ".weight"] = torch.randn(*shape, generator=g) * (gain * s / math.sqrt(fan))
        if bias:
            sd[name + ".bias"] = torch.randn(co, generator=g) * 0.1

    conv("patch_embed.proj", E, 3, 16, s=2.0)
    for i in range(c["enc_depth"]):
        p = f"enc_blocks.{i}."
        norm(p + "norm1", E)
        lin(p + "attn.qkv", 3 * E, E)
        lin(p + "attn.proj", E, E, 0.5)
        norm(p + "norm2", E)
        lin(p + "mlp.fc1", 4 * E, E)
        lin(p + "mlp.fc2", E, 4 * E, 0.7)
    norm("enc_norm", E)
    lin("decoder_embed", D, E)
    for blocks in ("dec_blocks", "dec_blocks2"):
        for i in range(c["dec_depth"]):
            p = f"{blocks}.{i}."
            norm(p + "norm1", D)
            lin(p + "attn.qkv", 3 * D, D)
            lin(p + "attn.proj", D, D, 0.5)
            norm(p + "norm2", D)
            norm(p + "norm_y", D)
            for n in ("projq", "projk", "projv"):
                lin(p + "cross_attn." + n, D, D)
            lin(p + "cross_attn.proj", D, D, 0.5)
            norm(p + "norm3", D)
            lin(p + "mlp.fc1", 4 * D, D)
            lin(p + "mlp.fc2", D, 4 * D, 0.7)
    norm("dec_norm", D)
    for hd in (1, 2):
        p = f"downstream_head{hd}.dpt."
        dims_in = (E, D, D, D)
        for k, ld in enumerate(DUST3R_LAYER_DIMS):
            conv(f"{p}act_postprocess.{k}.0", ld, dims_in[k], 1)
            if k == 0:
                conv(f"{p}act_postprocess.0.1", ld, ld, 4, transposed=True)
            elif k == 1:
                conv(f"{p}act_postprocess.1.1", ld, ld, 2, transposed=True)
            elif k == 3:
                conv(f"{p}act_postprocess.3.1", ld, ld, 3)
            conv(f"{p}scratch.layer_rn.{k}", 256, ld, 3, bias=False)
            sd[f"{p}scratch.layer{k + 1}_rn.weight"] = sd[f"{p}scratch.layer_rn.{k}.weight"]  # upstream registers the module twice
        for r in (1, 2, 3, 4):
            for u in (1, 2):
                conv(f"{p}scratch.refinenet{r}.resConfUnit{u}.conv1", 256, 256, 3, 1.2)
                conv(f"{p}scratch.refinenet{r}.resConfUnit{u}.conv2", 256, 256, 3, 0.6)
            conv(f"{p}scratch.refinenet{r}.out_conv", 256, 256, 1, 0.8)
        conv(p + "head.0", 128, 256, 3)
        conv(p + "head.2", 128, 128, 3, 1.4)
        conv(p + "head.4", 4, 128, 1, 0.5)
        dd = c.get("desc_dim", 0)
        if dd:
            q = f"downstream_head{hd}.head_local_features."
            lin(q + "fc1", 4 * (E + D), E + D)
            lin(q + "fc2", (dd + 1) * 256, 4 * (E + D), 1.5)
    return sd
