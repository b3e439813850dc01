"""Seeded synthetic weights in the upstream state-dict layouts (oracle side).

No checkpoint exists offline (weights come from the HF hub
at `_init` time, imcui/hloc/extractors/superpoint.py:48-53,
imcui/hloc/matchers/lightglue.py:39-51), so parity runs load the SAME seeded
tensors into the oracle and into the HIP backend.  Key names/shapes follow
SURVEY.md Appendix A.1 / A.2, so a real `superpoint_v1.pth` /
`superpoint_lightglue.pth` state dict drops in unchanged.
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
        sd["convDb.weight"] = (w - w.mean(dim=1, keepdim=True)) * 2.0
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


def lightglue_state_dict(seed: int = 0, n_layers: int = 9, dim: int = 256, heads: int = 4, structured: bool = True) -> dict:
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
    damp = 0.03 if structured else 1.0
    for i in range(n_layers):
        p = f"transformers.{i}."
        sd.update(lin(3 * dim, dim, scale=10.0 if structured else 1.0, prefix=p + "self_attn.Wqkv"))
        sd.update(lin(dim, dim, prefix=p + "self_attn.out_proj"))
        sd.update(lin(2 * dim, 2 * dim, prefix=p + "self_attn.ffn.0"))
        sd[p + "self_attn.ffn.1.weight"] = 1.0 + (0.0 if structured else 0.1) * torch.randn(2 * dim, generator=g)
        sd[p + "self_attn.ffn.1.bias"] = (0.0 if structured else 0.1) * torch.randn(2 * dim, generator=g)
        sd.update(lin(dim, 2 * dim, scale=damp, prefix=p + "self_attn.ffn.3"))
        if structured:  # zero-mean rows: no common-mode drift of the residual stream
            w = sd[p + "self_attn.ffn.3.weight"]
            sd[p + "self_attn.ffn.3.weight"] = w - w.mean(dim=1, keepdim=True)
            sd[p + "self_attn.ffn.3.bias"] = sd[p + "self_attn.ffn.3.bias"] * 0.05
        sd.update(lin(dim, dim, scale=10.0 if structured else 1.0, prefix=p + "cross_attn.to_qk"))
        sd.update(lin(dim, dim, prefix=p + "cross_attn.to_v"))
        sd.update(lin(dim, dim, prefix=p + "cross_attn.to_out"))
        sd.update(lin(2 * dim, 2 * dim, prefix=p + "cross_attn.ffn.0"))
        sd[p + "cross_attn.ffn.1.weight"] = 1.0 + (0.0 if structured else 0.1) * torch.randn(2 * dim, generator=g)
        sd[p + "cross_attn.ffn.1.bias"] = (0.0 if structured else 0.1) * torch.randn(2 * dim, generator=g)
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
            sd[q + "final_proj.weight"] = sd[q + "final_proj.weight"] + 4.0 * math.sqrt(60.0) * torch.eye(dim)
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
    return sd
