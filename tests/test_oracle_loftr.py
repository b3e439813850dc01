"""LoFTR oracle on CPU (parity unpinned: no reference implementation of the whole network is available).

End to end: two crops of one image offset by a multiple of the coarse stride must match cell-to-cell with the known displacement
(a wrong unfold / transposition / image swap breaks it).  Components: the ResNet block against transformers' ResNetBasicLayer, linear
attention against its quadratic formulation, fine matching against the copy of kornia's `spatial_expectation2d` in transformers,
coarse matching's border / mutual-maximum logic against transformers' copy of LoFTR's `mask_border`.
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


# ---------------------------------------------------------------------------------------------------------------------
# Component checks against independent formulations available in the container (the network as a whole stays unpinned).
import math

import pytest
import torch.nn.functional as F

SD = None


def _sd():
    global SD
    if SD is None:
        SD = loftr_state_dict(3, structured=False)
    return SD


def test_basic_block_against_the_transformers_resnet_layer():
    """ResNetFPN's BasicBlock (conv3x3-BN-ReLU, conv3x3-BN, 1x1-conv-BN shortcut when the shape changes, ReLU after the sum) is
    transformers' ResNetBasicLayer: same tensors, eval-mode BatchNorm, same output -- for a stride-2 and a stride-1 block."""
    resnet = pytest.importorskip("transformers.models.resnet.modeling_resnet")
    sd = _sd()
    o = LoFTROracle(sd)
    g = torch.Generator().manual_seed(1)
    for p, cin, cout, stride in (("backbone.layer2.0", 128, 196, 2), ("backbone.layer2.1", 196, 196, 1)):
        layer = resnet.ResNetBasicLayer(cin, cout, stride=stride).eval()
        tens = {}
        for k, (conv, bn) in enumerate(((".conv1", ".bn1"), (".conv2", ".bn2"))):
            tens[f"layer.{k}.convolution.weight"] = sd[p + conv + ".weight"]
            for t in ("weight", "bias", "running_mean", "running_var"):
                tens[f"layer.{k}.normalization.{t}"] = sd[p + bn + "." + t]
        if stride != 1 or cin != cout:
            tens["shortcut.convolution.weight"] = sd[p + ".downsample.0.weight"]
            for t in ("weight", "bias", "running_mean", "running_var"):
                tens[f"shortcut.normalization.{t}"] = sd[p + ".downsample.1." + t]
        layer.load_state_dict(tens, strict=False)  # num_batches_tracked stays at its default
        x = torch.randn(2, cin, 12, 16, generator=g)
        with torch.no_grad():
            want = layer(x.clone())
        got = o._block(x, p, stride)
        assert got.shape == want.shape and (got - want).abs().max().item() < 1e-4 * want.abs().max().item()


def test_linear_attention_is_normalised_kernel_attention():
    """`LinearAttention` (Q' (K'^T V) / (Q' sum K')) equals the quadratic formulation out_l = sum_s k(q_l, k_s) v_s / sum_s k(q_l, k_s) with
    k(q, k) = (elu(q) + 1) . (elu(k) + 1), per head -- evaluated here the slow way through an encoder layer's projections."""
    sd = _sd()
    o = LoFTROracle(sd)
    g = torch.Generator().manual_seed(2)
    x, src = torch.randn(1, 40, 256, generator=g), torch.randn(1, 56, 256, generator=g)
    p = "loftr_coarse.layers.1"
    got = o._encoder_layer(p, x, src, 8)
    q = F.linear(x, sd[p + ".q_proj.weight"]).view(1, 40, 8, 32)
    k = F.linear(src, sd[p + ".k_proj.weight"]).view(1, 56, 8, 32)
    v = F.linear(src, sd[p + ".v_proj.weight"]).view(1, 56, 8, 32)
    kern = torch.einsum("nlhd,nshd->nhls", F.elu(q) + 1, F.elu(k) + 1).double()
    msg = (torch.einsum("nhls,nshd->nlhd", kern, v.double()) / (kern.sum(-1).permute(0, 2, 1)[..., None] + 1e-6)).float()
    msg = F.linear(msg.reshape(1, 40, 256), sd[p + ".merge.weight"])
    msg = F.layer_norm(msg, (256,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    msg = F.linear(F.relu(F.linear(torch.cat([x, msg], 2), sd[p + ".mlp.0.weight"])), sd[p + ".mlp.2.weight"])
    want = x + F.layer_norm(msg, (256,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    assert (got - want).abs().max().item() < 2e-4 * want.abs().max().item()


def test_fine_matching_expectation_against_the_kornia_copy_in_transformers():
    """FineMatching: soft-max of the centre-vs-window similarities / sqrt(C), then kornia's `spatial_expectation2d` -- the copy of that
    kornia function shipped in transformers' EfficientLoFTR port gives the same sub-pixel offsets."""
    elo = pytest.importorskip("transformers.models.efficientloftr.modeling_efficientloftr")
    o = LoFTROracle(_sd())
    g = torch.Generator().manual_seed(3)
    M, C = 37, 128
    f0, f1 = torch.randn(M, 25, C, generator=g), torch.randn(M, 25, C, generator=g)
    cm = {"mkpts0_c": torch.rand(M, 2, generator=g) * 100, "mkpts1_c": torch.rand(M, 2, generator=g) * 100, "mconf": torch.rand(M, generator=g)}
    k0, k1 = o.fine_matching(f0, f1, cm, scale=2.0)
    heat = torch.softmax(torch.einsum("mc,mrc->mr", f0[:, 12], f1) / C**0.5, 1).view(1, M, 5, 5)
    coords = elo.spatial_expectation2d(heat, True)[0]  # [M, 2] (x, y) in [-1, 1]
    assert torch.equal(k0, cm["mkpts0_c"])
    assert (k1 - (cm["mkpts1_c"] + coords * 2 * 2.0)).abs().max().item() < 1e-4


def test_coarse_matching_border_and_mutual_maximum():
    """CoarseMatching on features built so that cell i of image 0 matches cell i of image 1 with confidence ~1: the match list is
    exactly the cells inside the 2-cell border (transformers' copy of LoFTR's `mask_border` on the 4-D grid), in row-major order,
    with pixel coordinates = grid x 8."""
    elo = pytest.importorskip("transformers.models.efficientloftr.modeling_efficientloftr")
    o = LoFTROracle(_sd())
    h, w = 7, 9
    L = h * w
    f = F.pad(torch.eye(L), (0, 256 - L)) * 16.0 * math.sqrt(12.0)  # sim = 12 on the diagonal / temperature 0.1 -> soft-max ~ 1
    cm = o.coarse_matching(f[None], f[None], (h, w), (h, w), (h * 8, w * 8), 0.2)
    mask = torch.ones(1, h, w, h, w, dtype=torch.bool)
    mask = elo.mask_border(mask, 2, False)
    inner = mask[0].reshape(L, L).diagonal().nonzero()[:, 0]
    assert torch.equal(cm["i_ids"], inner) and torch.equal(cm["j_ids"], inner) and (cm["mconf"] > 0.99).all()
    assert torch.equal(cm["mkpts0_c"], torch.stack([inner % w, inner // w], 1) * 8.0)


def test_backbone_against_a_module_tree_written_from_the_published_layer_list():
    """the FPN of the LoFTR oracle had no independent cross-check.  `ResNetFPN_8_2` is re-expressed here as a plain
    `torch.nn` module tree from kornia's published layer list (feature/loftr/backbone/resnet_fpn.py: conv1 7x7/2 -> 128, three stages of two
    BasicBlocks 128 / 196 / 256 with 1x1 down-sample branches, 1x1 lateral convolutions, x2 bilinear up-sampling with align_corners=True, and the
    conv3x3 - BN - LeakyReLU - conv3x3 merge blocks), with kornia's attribute names -- so loading the seeded state dict with strict=True checks the
    key set and shapes, and the forward pass checks the wiring the functional oracle restates (strides, paddings, which sum feeds which block)."""
    import torch.nn as nn
    import torch.nn.functional as F

    def conv3x3(i, o, s=1):
        return nn.Conv2d(i, o, 3, s, 1, bias=False)

    def conv1x1(i, o, s=1):
        return nn.Conv2d(i, o, 1, s, 0, bias=False)

    class BasicBlock(nn.Module):
        def __init__(self, i, o, stride):
            super().__init__()
            self.conv1, self.conv2 = conv3x3(i, o, stride), conv3x3(o, o)
            self.bn1, self.bn2 = nn.BatchNorm2d(o), nn.BatchNorm2d(o)
            self.relu = nn.ReLU(inplace=True)
            self.downsample = None if stride == 1 else nn.Sequential(conv1x1(i, o, stride), nn.BatchNorm2d(o))

        def forward(self, x):
            y = self.relu(self.bn1(self.conv1(x)))
            y = self.bn2(self.conv2(y))
            if self.downsample is not None:
                x = self.downsample(x)
            return self.relu(x + y)

    class ResNetFPN82(nn.Module):
        def __init__(self, initial=128, dims=(128, 196, 256)):
            super().__init__()
            self.conv1 = nn.Conv2d(1, initial, 7, 2, 3, bias=False)
            self.bn1 = nn.BatchNorm2d(initial)
            self.relu = nn.ReLU(inplace=True)
            self.layer1 = nn.Sequential(BasicBlock(initial, dims[0], 1), BasicBlock(dims[0], dims[0], 1))
            self.layer2 = nn.Sequential(BasicBlock(dims[0], dims[1], 2), BasicBlock(dims[1], dims[1], 1))
            self.layer3 = nn.Sequential(BasicBlock(dims[1], dims[2], 2), BasicBlock(dims[2], dims[2], 1))
            self.layer3_outconv = conv1x1(dims[2], dims[2])
            self.layer2_outconv = conv1x1(dims[1], dims[2])
            self.layer2_outconv2 = nn.Sequential(conv3x3(dims[2], dims[2]), nn.BatchNorm2d(dims[2]), nn.LeakyReLU(), conv3x3(dims[2], dims[1]))
            self.layer1_outconv = conv1x1(dims[0], dims[1])
            self.layer1_outconv2 = nn.Sequential(conv3x3(dims[1], dims[1]), nn.BatchNorm2d(dims[1]), nn.LeakyReLU(), conv3x3(dims[1], dims[0]))

        def forward(self, x):
            x0 = self.relu(self.bn1(self.conv1(x)))
            x1 = self.layer1(x0)
            x2 = self.layer2(x1)
            x3 = self.layer3(x2)
            x3_out = self.layer3_outconv(x3)
            x3_out_2x = F.interpolate(x3_out, scale_factor=2.0, mode="bilinear", align_corners=True)
            x2_out = self.layer2_outconv2(self.layer2_outconv(x2) + x3_out_2x)
            x2_out_2x = F.interpolate(x2_out, scale_factor=2.0, mode="bilinear", align_corners=True)
            x1_out = self.layer1_outconv2(self.layer1_outconv(x1) + x2_out_2x)
            return x3_out, x1_out

    sd = loftr_state_dict(0)
    net = ResNetFPN82().eval()
    missing = net.load_state_dict({k[len("backbone."):]: v for k, v in sd.items() if k.startswith("backbone.")}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert sum(p.numel() for p in net.parameters()) == sum(v.numel() for k, v in sd.items() if k.startswith("backbone.") and "running" not in k and "num_batches" not in k)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 1, 96, 128, generator=g)
    ora = LoFTROracle(sd)
    with torch.no_grad():
        c_mod, f_mod = net(x)
        c_ora, f_ora = ora.backbone(x)
    assert c_mod.shape == (2, 256, 12, 16) and f_mod.shape == (2, 128, 48, 64)
    assert (c_mod - c_ora).abs().max() <= 1e-5 * c_ora.abs().max()
    assert (f_mod - f_ora).abs().max() <= 1e-5 * f_ora.abs().max()
