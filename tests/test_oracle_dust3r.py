"""CPU checks of the DUSt3R oracle's building blocks against independent formulations available in the container
(oracle/dust3r.py: the network itself is "parity unpinned" -- upstream's sources are an un-vendored submodule)."""
import math

import pytest
import torch
import torch.nn.functional as F

from imcui_hip.synth_weights import dust3r_state_dict
from oracle.dust3r import DUSt3ROracle, rope2d

CFG = {"enc_dim": 128, "enc_depth": 2, "dec_dim": 128, "dec_depth": 4}


def test_rope2d_is_a_rotation_by_position_times_frequency():
    """Complex-number formulation: inside each 32-wide half, feature i < 16 and feature i + 16 form one complex number that is
    multiplied by exp(j * pos * 100^(-i/16)); the y half uses the row, the x half the column."""
    g = torch.Generator().manual_seed(0)
    t = torch.randn(2, 3, 35, 64, generator=g)
    pos = torch.stack((torch.randint(0, 32, (2, 35), generator=g), torch.randint(0, 32, (2, 35), generator=g)), -1)
    got = rope2d(t, pos, 100.0)
    want = torch.empty_like(t)
    for half in (0, 1):
        z = torch.complex(t[..., 32 * half : 32 * half + 16].double(), t[..., 32 * half + 16 : 32 * half + 32].double())
        f = 100.0 ** (-torch.arange(16, dtype=torch.float64) / 16)
        ang = pos[..., half][:, None, :, None].double() * f
        z = z * torch.polar(torch.ones_like(ang), ang)
        want[..., 32 * half : 32 * half + 16] = z.real.float()
        want[..., 32 * half + 16 : 32 * half + 32] = z.imag.float()
    assert (got - want).abs().max().item() < 1e-5
    # relative-position property: <rope(q, p), rope(k, p')> depends on p - p' only
    q, k = torch.randn(1, 1, 1, 64, generator=g), torch.randn(1, 1, 1, 64, generator=g)
    def dot(pq, pk):
        return (rope2d(q, torch.tensor([[pq]]), 100.0) * rope2d(k, torch.tensor([[pk]]), 100.0)).sum().item()
    assert abs(dot([3, 7], [1, 2]) - dot([13, 17], [11, 12])) < 1e-4


def test_attention_block_against_sdpa():
    sd = dust3r_state_dict(1, CFG)
    o = DUSt3ROracle(sd, CFG)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 24, 128, generator=g)
    pos = o.positions(2, 4, 6)
    got = o._self_attn(x, pos, "enc_blocks.0.attn")
    qkv = F.linear(x, sd["enc_blocks.0.attn.qkv.weight"], sd["enc_blocks.0.attn.qkv.bias"]).view(2, 24, 3, 2, 64)
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    att = F.scaled_dot_product_attention(rope2d(q, pos, 100.0), rope2d(k, pos, 100.0), v)
    want = F.linear(att.transpose(1, 2).reshape(2, 24, 128), sd["enc_blocks.0.attn.proj.weight"], sd["enc_blocks.0.attn.proj.bias"])
    assert (got - want).abs().max().item() < 1e-4 * want.abs().max().item()


def test_fusion_blocks_against_the_transformers_dpt_modules():
    """`FeatureFusionBlock_custom` (pre-activation residual units, x2 bilinear align_corners=True, 1x1 out_conv) restated by
    transformers' DPTFeatureFusionLayer: same tensors loaded, same outputs."""
    dpt = pytest.importorskip("transformers.models.dpt.modeling_dpt")
    from transformers import DPTConfig

    sd = dust3r_state_dict(3, CFG)
    o = DUSt3ROracle(sd, CFG)
    conf = DPTConfig(fusion_hidden_size=256, use_batch_norm_in_fusion_residual=False)
    layer = dpt.DPTFeatureFusionLayer(conf, align_corners=True).eval()
    p = "downstream_head1.dpt.scratch.refinenet2."
    m = {"projection": "out_conv", "residual_layer1.convolution1": "resConfUnit1.conv1", "residual_layer1.convolution2": "resConfUnit1.conv2",
         "residual_layer2.convolution1": "resConfUnit2.conv1", "residual_layer2.convolution2": "resConfUnit2.conv2"}
    layer.load_state_dict({f"{a}.{t}": sd[f"{p}{b}.{t}"] for a, b in m.items() for t in ("weight", "bias")})
    g = torch.Generator().manual_seed(4)
    x, skip = torch.randn(1, 256, 6, 9, generator=g), torch.randn(1, 256, 6, 9, generator=g)
    with torch.no_grad():
        want2 = layer(x, skip)
        want1 = layer(x)
    assert (o._fusion(p[:-1], x, skip) - want2).abs().max().item() < 1e-4 * want2.abs().max().item()
    assert (o._fusion(p[:-1], x) - want1).abs().max().item() < 1e-4 * want1.abs().max().item()


def test_reassemble_against_the_transformers_dpt_stage():
    """Token maps -> (x4 transposed conv, x2 transposed conv, identity, 3x3 stride 2) after a 1x1 projection: transformers'
    DPTReassembleLayer with factors 4, 2, 1, 0.5 loaded with the same tensors."""
    dpt = pytest.importorskip("transformers.models.dpt.modeling_dpt")
    from transformers import DPTConfig

    sd = dust3r_state_dict(5, CFG)
    o = DUSt3ROracle(sd, CFG)
    g = torch.Generator().manual_seed(6)
    h, w = 4, 6
    toks = [torch.randn(1, h * w, 128, generator=g) for _ in range(5)]
    p = "downstream_head1.dpt."
    maps = []
    for k, hook in enumerate(o.hooks):
        conf = DPTConfig(hidden_size=128)
        lay = dpt.DPTReassembleLayer(conf, channels=(96, 192, 384, 768)[k], factor=(4, 2, 1, 0.5)[k]).eval()
        tens = {"projection.weight": sd[f"{p}act_postprocess.{k}.0.weight"], "projection.bias": sd[f"{p}act_postprocess.{k}.0.bias"]}
        if k != 2:
            tens["resize.weight"] = sd[f"{p}act_postprocess.{k}.1.weight"]
            tens["resize.bias"] = sd[f"{p}act_postprocess.{k}.1.bias"]
        lay.load_state_dict(tens)
        with torch.no_grad():
            m = lay(toks[hook].transpose(1, 2).reshape(1, 128, h, w))
        maps.append(F.conv2d(m, sd[f"{p}scratch.layer_rn.{k}.weight"], None, 1, 1))
    got = o.reassemble(toks, 1, h, w)
    for a, b in zip(got, maps):
        assert a.shape == b.shape and (a - b).abs().max().item() < 1e-4 * b.abs().max().item()


def test_symmetrised_driver_and_postprocessing():
    sd = dust3r_state_dict(7, CFG)
    o = DUSt3ROracle(sd, CFG)
    g = torch.Generator().manual_seed(8)
    a, b = torch.rand(1, 3, 64, 96, generator=g), torch.rand(1, 3, 64, 96, generator=g)
    r = o.inference_symmetrized(a, b, return_intermediates=True)
    assert r["pred1"]["pts3d"].shape == (2, 64, 96, 3) and r["pred2"]["pts3d_in_other_view"].shape == (2, 64, 96, 3)
    assert (r["pred1"]["conf"] > 1).all()
    raw = r["_passes"][0][0]["_raw"]
    d = raw[..., :3].norm(dim=-1)
    assert torch.allclose(r["pred1"]["pts3d"][:1].norm(dim=-1), torch.expm1(d), rtol=1e-5, atol=1e-6)  # |pts| = expm1(|xyz|)
    # the swapped call is the swapped pair
    s = o.inference_symmetrized(b, a)
    assert torch.equal(s["pred1"]["pts3d"][0], r["pred1"]["pts3d"][1]) and torch.equal(s["pred2"]["conf"][1], r["pred2"]["conf"][0])


def test_views_of_two_sizes():
    """Upstream's `_encode_image_pairs` encodes views of different sizes one after the other and `inference` collates the per-pair
    results as lists.  Properties of the restatement: each view keeps its own size; the encoder states do not depend on the OTHER
    view; exchanging the images exchanges the entries; a view-2 image cropped to whole patches changes view 1 only through the
    cross attention (same shapes, different values)."""
    sd = dust3r_state_dict(7, CFG)
    o = DUSt3ROracle(sd, CFG)
    g = torch.Generator().manual_seed(9)
    a, b = torch.rand(1, 3, 64, 96, generator=g), torch.rand(1, 3, 80, 48, generator=g)
    r = o.inference_symmetrized(a, b, return_intermediates=True)
    assert [tuple(m.shape) for m in r["pred1"]["pts3d"]] == [(80, 48, 3), (64, 96, 3)]
    assert [tuple(m.shape) for m in r["pred2"]["pts3d_in_other_view"]] == [(64, 96, 3), (80, 48, 3)]
    assert [tuple(m.shape) for m in r["pred2"]["conf"]] == [(64, 96), (80, 48)]
    s = o.inference_symmetrized(b, a)
    assert torch.equal(s["pred1"]["pts3d"][0], r["pred1"]["pts3d"][1]) and torch.equal(s["pred2"]["conf"][1], r["pred2"]["conf"][0])
    # the encoder of image a is the same whoever it is paired with (and whichever view it is)
    same = o.inference_symmetrized(a, a[..., :48, :], return_intermediates=True)
    e1 = r["_passes"][1][0]["_enc_layers"][-1]     # a as view 1 of (a, b)
    e2 = same["_passes"][1][0]["_enc_layers"][-1]  # a as view 1 of (a, crop of a)
    e3 = r["_passes"][0][1]["_enc_layers"][-1]     # a as view 2 of (b, a)
    assert torch.allclose(e1, e2, atol=1e-5) and torch.allclose(e1, e3, atol=1e-5)
    assert not torch.allclose(r["pred1"]["pts3d"][1], same["pred1"]["pts3d"][1], atol=1e-3)
    # the decoder's cross attention really runs between sequences of different length: 24 query tokens against 15 keys
    assert r["_passes"][1][0]["_dec"][1].shape[1] == 24 and r["_passes"][1][1]["_dec"][1].shape[1] == 15


def test_pack_dust3r_layout_matches_the_library():
    """The host packer walks the state dict in the order of the C layer table and every shape agrees (no GPU needed)."""
    from imcui_hip.backend import dust3r_cfg_of, pack_dust3r
    from imcui_hip.lib_loader import load_library

    lib = load_library()
    for dd in (0, 24):  # DUSt3R, MASt3R (two more matrices per head)
        cfg = {"enc_dim": 128, "enc_depth": 1, "dec_dim": 64, "dec_depth": 4, "desc_dim": dd}
        sd = dust3r_state_dict(9, cfg)
        assert dust3r_cfg_of(sd) == cfg
        packed, c = pack_dust3r(sd)
        assert c == cfg and packed.numel() == lib.imcui_hip_dust3r_packed_floats(128, 1, 64, 4, dd)
        assert lib.imcui_hip_dust3r_workspace_bytes(128, 1, 64, 4, dd, 2, 2, 96, 128) > 0
    assert lib.imcui_hip_dust3r_num_layers(128, 1, 64, 4, 24) == lib.imcui_hip_dust3r_num_layers(128, 1, 64, 4, 0) + 4
    assert lib.imcui_hip_dust3r_num_layers(100, 1, 64, 4, 0) == 0  # widths must be multiples of 64
    assert lib.imcui_hip_dust3r_workspace_bytes(128, 1, 64, 4, 0, 2, 2, 100, 128) == 0  # sizes must be multiples of 16
    # images of several sizes: the workspace is the one of the largest token grid, the token dump holds R rows per sequence
    import ctypes as C

    c5 = (128, 1, 64, 4, 0)
    same = (C.c_int * 4)(96, 128, 96, 128)
    assert lib.imcui_hip_dust3r_workspace_bytes_sizes(*c5, 2, same, 2) == lib.imcui_hip_dust3r_workspace_bytes(*c5, 2, 2, 96, 128)
    mixed = (C.c_int * 4)(96, 128, 256, 64)  # 48 and 64 tokens -> the 256 x 64 image sizes everything
    assert lib.imcui_hip_dust3r_workspace_bytes_sizes(*c5, 2, mixed, 2) == lib.imcui_hip_dust3r_workspace_bytes(*c5, 2, 2, 256, 64)
    R = 128  # 64 tokens rounded up to the attention tile
    assert lib.imcui_hip_dust3r_token_dump_floats(*c5, 2, mixed, 2) == (1 + 2) * 2 * R * 128 + (4 + 2) * 4 * R * 64
    assert lib.imcui_hip_dust3r_workspace_bytes_sizes(*c5, 2, (C.c_int * 4)(96, 128, 100, 64), 2) == 0  # every size a multiple of 16


def test_mast3r_local_features_are_a_pixel_shuffle_of_the_token_mlp():
    """MASt3R head: descriptor channel c of pixel (y, x) is output (c * 256 + (y % 16) * 16 + x % 16) of the MLP on token
    (y // 16, x // 16); descriptors have unit norm, their confidence is exp(last channel)."""
    from oracle.dust3r import MASt3ROracle

    cfg = {"enc_dim": 128, "enc_depth": 1, "dec_dim": 64, "dec_depth": 4, "desc_dim": 5}
    sd = dust3r_state_dict(11, cfg)
    o = MASt3ROracle(sd, cfg)
    g = torch.Generator().manual_seed(12)
    a, b = torch.rand(1, 3, 32, 48, generator=g), torch.rand(1, 3, 32, 48, generator=g)
    res1, _ = o.forward((a - 0.5) / 0.5, (b - 0.5) / 0.5, return_intermediates=True)
    assert res1["desc"].shape == (1, 32, 48, 5) and res1["desc_conf"].shape == (1, 32, 48)
    assert torch.allclose(res1["desc"].norm(dim=-1), torch.ones(1, 32, 48), atol=1e-5)
    toks = res1["_dec"]
    cat = torch.cat((toks[0], toks[-1]), -1)
    p = "downstream_head1.head_local_features"
    lf = F.linear(F.gelu(F.linear(cat, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"])), sd[p + ".fc2.weight"], sd[p + ".fc2.bias"])[0]  # [T, 6 * 256]
    for y, x in ((0, 0), (17, 5), (31, 47), (16, 32)):
        t = (y // 16) * 3 + x // 16
        v = torch.stack([lf[t, c * 256 + (y % 16) * 16 + x % 16] for c in range(6)])
        assert torch.allclose(res1["desc"][0, y, x], v[:5] / v[:5].norm(), atol=1e-6)
        assert torch.allclose(res1["desc_conf"][0, y, x], v[5].exp(), rtol=1e-5)


def test_fast_reciprocal_nns_restatement():
    """MASt3R's matching step (mast3r.py:68-75): every returned pair is a reciprocal nearest neighbour by dot product, pairs are unique and
    ordered by (position in image 1, position in image 2), a shifted copy of a descriptor field is matched with the shift; the
    plugin's device-tensor loop gives the same result as the numpy-style restatement when both use the same search primitive."""
    from imcui_hip.hloc.matchers.mast3r import fast_reciprocal_nns as plugin_loop
    from oracle.dust3r import fast_reciprocal_nns, nn_dot_first_argmax

    g = torch.Generator().manual_seed(0)
    H, W, D = 40, 56, 24
    base = torch.randn(H + 8, W + 8, D, generator=g)
    d1 = F.normalize(base[:H, :W], dim=-1)
    d2 = F.normalize(base[4 : H + 4, 2 : W + 2] + 0.3 * torch.randn(H, W, D, generator=g), dim=-1)  # d2[y, x] ~ d1[y + 4, x + 2]
    xy1, xy2 = fast_reciprocal_nns(d1, d2, subsample=2)
    assert len(xy1) > 200
    assert ((xy1 - xy2) == torch.tensor([2, 4])).all(1).float().mean().item() > 0.9
    lin1, lin2 = xy1[:, 1] * W + xy1[:, 0], xy2[:, 1] * W + xy2[:, 0]
    key = lin1 * (H * W) + lin2
    assert (key[1:] > key[:-1]).all()  # unique and sorted
    p1, p2 = d1.reshape(-1, D), d2.reshape(-1, D)
    assert torch.equal(nn_dot_first_argmax(p1[lin1], p2), lin2) and torch.equal(nn_dot_first_argmax(p2[lin2], p1), lin1)
    # blocks: a later block only wins with a strictly larger value -> first arg-max (duplicated rows)
    db = torch.cat((p2[:100], p2[:100]), 0)
    assert (nn_dot_first_argmax(p2[:100], db, block=64) == torch.arange(100)).all()
    b1, b2 = plugin_loop(d1, d2, subsample=2, nn=nn_dot_first_argmax)
    assert torch.equal(b1, xy1) and torch.equal(b2, xy2)


def test_encoder_block_against_the_transformers_vit_layer():
    """The CroCo encoder block without its rotary embedding (all positions 0 -> the rotation is the identity) is the standard pre-norm
    ViT block: transformers' ViTLayer loaded with the same tensors (fused qkv split into its q / k / v Linears) gives the same output."""
    vit = pytest.importorskip("transformers.models.vit.modeling_vit")
    from transformers import ViTConfig

    sd = dust3r_state_dict(13, CFG)
    o = DUSt3ROracle(sd, CFG)
    E = CFG["enc_dim"]
    conf = ViTConfig(hidden_size=E, num_attention_heads=E // 64, intermediate_size=4 * E, hidden_act="gelu", layer_norm_eps=1e-6, qkv_bias=True,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    conf._attn_implementation = "eager"
    layer = vit.ViTLayer(conf).eval()
    p = "enc_blocks.1."
    qkv_w, qkv_b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
    tens = {"layernorm_before.weight": sd[p + "norm1.weight"], "layernorm_before.bias": sd[p + "norm1.bias"],
            "layernorm_after.weight": sd[p + "norm2.weight"], "layernorm_after.bias": sd[p + "norm2.bias"]}
    names = set(layer.state_dict())
    if "attention.q_proj.weight" in names:  # current module tree of the port
        out_proj, fc1, fc2, qkv = "attention.o_proj", "mlp.fc1", "mlp.fc2", ("attention.q_proj", "attention.k_proj", "attention.v_proj")
    else:  # older releases
        out_proj, fc1, fc2 = "attention.output.dense", "intermediate.dense", "output.dense"
        qkv = ("attention.attention.query", "attention.attention.key", "attention.attention.value")
    for dst, src in ((out_proj, "attn.proj"), (fc1, "mlp.fc1"), (fc2, "mlp.fc2")):
        tens[dst + ".weight"], tens[dst + ".bias"] = sd[p + src + ".weight"], sd[p + src + ".bias"]
    for i, n in enumerate(qkv):
        tens[n + ".weight"] = qkv_w[i * E : (i + 1) * E]
        tens[n + ".bias"] = qkv_b[i * E : (i + 1) * E]
    layer.load_state_dict(tens)
    g = torch.Generator().manual_seed(14)
    x = torch.randn(2, 35, E, generator=g)
    pos = torch.zeros(2, 35, 2, dtype=torch.long)
    with torch.no_grad():
        want = layer(x)
    want = want[0] if isinstance(want, tuple) else want
    got = o._enc_block(x, pos, "enc_blocks.1")
    assert (got - want).abs().max().item() < 1e-4 * want.abs().max().item()


def test_cross_attention_against_torch_multihead_attention():
    """The decoder's cross attention without rotation (positions 0): queries from x, keys / values from the other view's tokens,
    separate projections with biases, 64-wide heads, output projection -- torch.nn.MultiheadAttention with the same tensors."""
    sd = dust3r_state_dict(15, CFG)
    o = DUSt3ROracle(sd, CFG)
    D = CFG["dec_dim"]
    p = "dec_blocks2.2.cross_attn"
    mha = torch.nn.MultiheadAttention(D, D // 64, bias=True, batch_first=True).eval()
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.cat([sd[f"{p}.proj{n}.weight"] for n in "qkv"], 0))
        mha.in_proj_bias.copy_(torch.cat([sd[f"{p}.proj{n}.bias"] for n in "qkv"], 0))
        mha.out_proj.weight.copy_(sd[p + ".proj.weight"])
        mha.out_proj.bias.copy_(sd[p + ".proj.bias"])
    g = torch.Generator().manual_seed(16)
    x, y = torch.randn(2, 20, D, generator=g), torch.randn(2, 28, D, generator=g)
    zx, zy = torch.zeros(2, 20, 2, dtype=torch.long), torch.zeros(2, 28, 2, dtype=torch.long)
    with torch.no_grad():
        want = mha(x, y, y, need_weights=False)[0]
    got = o._cross_attn(x, y, zx, zy, p)
    assert (got - want).abs().max().item() < 1e-4 * want.abs().max().item()


def test_plugin_host_logic_on_a_mocked_device(monkeypatch):
    """The duster / mast3r plugins' host side with the device call replaced by the oracle (CPU only): `inference_output` lists the
    directed pairs in `make_pairs`' order -- (image1, image0) first, then (image0, image1) -- and `Mast3r._forward` matches the
    descriptors of the (image0, image1) entry, so keypoints0 are pixels of image0 and keypoints1 of image1 (mast3r.py:61-96)."""
    import numpy as np

    from imcui_hip import backend
    from imcui_hip.hloc.matchers.mast3r import Mast3r
    from oracle.dust3r import MASt3ROracle, fast_reciprocal_nns, nn_dot_first_argmax

    cfg = {"enc_dim": 128, "enc_depth": 1, "dec_dim": 64, "dec_depth": 4, "desc_dim": 24}
    sd = dust3r_state_dict(21, cfg)
    ora = MASt3ROracle(sd, cfg)

    def fake_forward(self, packed, net_cfg, images, pairs, dump=False, arith=0):
        norm = (images - 0.5) / 0.5
        res = [ora.forward(norm[a : a + 1], norm[b : b + 1]) for a, b in pairs]
        out = {"pts3d": torch.stack([torch.cat([r[0]["pts3d"] for r in res]), torch.cat([r[1]["pts3d_in_other_view"] for r in res])]),
               "conf": torch.stack([torch.cat([r[0]["conf"] for r in res]), torch.cat([r[1]["conf"] for r in res])]),
               "desc": torch.stack([torch.cat([r[0]["desc"] for r in res]), torch.cat([r[1]["desc"] for r in res])]),
               "desc_conf": torch.stack([torch.cat([r[0]["desc_conf"] for r in res]), torch.cat([r[1]["desc_conf"] for r in res])])}
        return out

    def fake_forward_sizes(self, packed, net_cfg, images, pairs, dump=False, arith=0):
        norm = [(im.reshape((1,) + tuple(im.shape[-3:])) - 0.5) / 0.5 for im in images]
        res = [ora.forward(norm[a], norm[b]) for a, b in pairs]
        return {"pts3d": [[r[0]["pts3d"][0] for r in res], [r[1]["pts3d_in_other_view"][0] for r in res]],
                "conf": [[r[0]["conf"][0] for r in res], [r[1]["conf"][0] for r in res]]}

    monkeypatch.setattr(backend.DUSt3RHIP, "forward", fake_forward)
    monkeypatch.setattr(backend.DUSt3RHIP, "forward_sizes", fake_forward_sizes)
    monkeypatch.setattr(backend, "nn_argmax", lambda q, db, return_best=False, split=False: nn_dot_first_argmax(q, db))
    monkeypatch.setattr(backend, "get_precision", lambda dev: 1)
    model = Mast3r({"state_dict": sd, "max_keypoints": 50}).eval()
    g = torch.Generator().manual_seed(22)
    i0, i1 = torch.rand(1, 3, 48, 64, generator=g), torch.rand(1, 3, 48, 64, generator=g)
    data = {"image0": i0, "image1": i1}
    out = model.inference_output(data)
    ref = ora.inference_symmetrized(i0, i1)
    for pred, keys in (("pred1", ("pts3d", "conf", "desc", "desc_conf")), ("pred2", ("pts3d_in_other_view", "conf", "desc", "desc_conf"))):
        for k in keys:
            assert torch.equal(out[pred][k], ref[pred][k]), (pred, k)
    assert out["view1"]["idx"] == [1, 0] and out["view2"]["idx"] == [0, 1]
    assert torch.equal(out["view1"]["img"][1], ((i0 - 0.5) / 0.5)[0])  # entry 1: image0 is view 1
    # the matcher: image0's descriptors as seen in the pair (image0, image1), against image1's
    r1, r2 = ora.forward((i0 - 0.5) / 0.5, (i1 - 0.5) / 0.5)
    k0, k1 = fast_reciprocal_nns(r1["desc"][0], r2["desc"][0], subsample=2)
    if len(k0) > 50:
        keep = np.round(np.linspace(0, len(k0) - 1, 50)).astype(int)
        k0, k1 = k0[keep], k1[keep]
    pred = model(data)
    assert len(k0) > 5 and torch.equal(pred["keypoints0"], k0) and torch.equal(pred["keypoints1"], k1)


def test_decoder_is_symmetric_under_exchanging_views_and_weight_sets():
    """Both decoder stacks read the OTHER view's tokens as they were BEFORE the block.  Then exchanging the two images together with
    `dec_blocks` <-> `dec_blocks2` and `downstream_head1` <-> `downstream_head2` exchanges the outputs exactly; an implementation in
    which the second stack saw the first stack's already-updated tokens would break this."""
    sd = dust3r_state_dict(17, CFG)
    swapped = {}
    for k, v in sd.items():
        if k.startswith("dec_blocks2."):
            swapped["dec_blocks." + k[len("dec_blocks2.") :]] = v
        elif k.startswith("dec_blocks."):
            swapped["dec_blocks2." + k[len("dec_blocks.") :]] = v
        elif k.startswith("downstream_head1."):
            swapped["downstream_head2." + k[len("downstream_head1.") :]] = v
        elif k.startswith("downstream_head2."):
            swapped["downstream_head1." + k[len("downstream_head2.") :]] = v
        else:
            swapped[k] = v
    g = torch.Generator().manual_seed(18)
    a, b = torch.randn(1, 3, 48, 64, generator=g), torch.randn(1, 3, 48, 64, generator=g)
    r1, r2 = DUSt3ROracle(sd, CFG).forward(a, b)
    s1, s2 = DUSt3ROracle(swapped, CFG).forward(b, a)
    assert torch.allclose(r1["pts3d"], s2["pts3d_in_other_view"], rtol=1e-5, atol=1e-6) and torch.allclose(r2["pts3d_in_other_view"], s1["pts3d"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(r1["conf"], s2["conf"], rtol=1e-5) and torch.allclose(r2["conf"], s1["conf"], rtol=1e-5)


def test_packed_gemm_planes_reconstruct_the_state_dict():
    """Every GEMM layer of the packed DUSt3R / MASt3R buffer, read back through imcui_hip_dust3r_layer_offsets: (hi + lo) x scale in the
    fragment-major order [ceil(N/32)][K/16][2][32][8] is the matrix the packer was given (to 2^-21 of its largest entry), the bias
    follows, and the scale of a `dec_blocks2` layer also sits right behind its `dec_blocks` twin's (one launch serves both sides)."""
    import ctypes as C

    import numpy as np

    from imcui_hip.backend import dust3r_matrices, pack_dust3r
    from imcui_hip.lib_loader import load_library

    lib = load_library()
    cfg = {"enc_dim": 128, "enc_depth": 2, "dec_dim": 64, "dec_depth": 4, "desc_dim": 24}
    sd = dust3r_state_dict(31, cfg)
    packed, c = pack_dust3r(sd)
    _, ws, bs, _ = dust3r_matrices(sd)
    c5 = (128, 2, 64, 4, 24)
    raw = packed.numpy()
    halves = raw.view(np.float16)
    off = [C.c_size_t() for _ in range(4)]
    kind = C.c_int()
    scales = {}
    checked = 0
    for i, (w, b) in enumerate(zip(ws, bs)):
        assert lib.imcui_hip_dust3r_layer_offsets(*c5, i, *[C.byref(o) for o in off], C.byref(kind)) == 0
        ob, oh, ol, osc = (o.value for o in off)
        N, K = w.shape
        scales[i] = raw[osc]
        if b is not None:
            assert np.array_equal(raw[ob : ob + N], b.numpy())
        if kind.value != 0:
            continue
        nfr = (N + 31) // 32
        n = nfr * 32 * K
        rec = (halves[2 * oh : 2 * oh + n].astype(np.float64) + halves[2 * ol : 2 * ol + n].astype(np.float64)) * float(raw[osc])
        full = rec.reshape(nfr, K // 16, 2, 32, 8).transpose(0, 3, 1, 2, 4).reshape(nfr * 32, K)
        assert np.abs(full[:N] - w.numpy().astype(np.float64)).max() <= np.abs(w.numpy()).max() * 2.0**-21, i
        assert np.all(full[N:] == 0.0)
        checked += 1
    assert checked > 40
    # decoder twins: layer index of (side 0, block i, j) = 2 + 4 enc_depth + 7 i + j, side 1 = + 7 dec_depth
    base, nd = 2 + 4 * 2, 4
    for i in range(nd):
        for j in range(7):
            l0 = base + 7 * i + j
            assert lib.imcui_hip_dust3r_layer_offsets(*c5, l0, *[C.byref(o) for o in off], C.byref(kind)) == 0
            assert raw[off[3].value + 1] == scales[l0 + 7 * nd]


def test_duster_forward_with_a_mocked_aligner(monkeypatch):
    """`Duster._forward` end to end on CPU: the device call is replaced by the oracle network, `global_aligner` (cv2 PnP-RANSAC,
    host geometry that stays upstream's) by a scene that hands the network's own point maps back -- view 1 of (image0, image1) and
    view 2 expressed in view 1's frame, which is what PairViewer returns up to its re-projection -- with masks conf > median.
    The plugin's mask / pixel-grid / reciprocal-3-D-neighbour / linspace steps (KD-trees) must equal the brute-force restatement
    (`oracle/dust3r.py: duster_matches_from_scene`, imcui/hloc/matchers/duster.py:76-108)."""
    import numpy as np

    from imcui_hip import backend
    from imcui_hip.hloc.matchers.duster import Duster, find_reciprocal_matches, xy_grid
    from oracle.dust3r import duster_matches_from_scene

    cfg = {"enc_dim": 128, "enc_depth": 1, "dec_dim": 64, "dec_depth": 4}
    sd = dust3r_state_dict(41, cfg)
    ora = DUSt3ROracle(sd, cfg)

    def fake_forward(self, packed, net_cfg, images, pairs, dump=False, arith=0):
        norm = (images - 0.5) / 0.5
        res = [ora.forward(norm[a : a + 1], norm[b : b + 1]) for a, b in pairs]
        return {"pts3d": torch.stack([torch.cat([r[0]["pts3d"] for r in res]), torch.cat([r[1]["pts3d_in_other_view"] for r in res])]),
                "conf": torch.stack([torch.cat([r[0]["conf"] for r in res]), torch.cat([r[1]["conf"] for r in res])])}

    class Scene:
        def __init__(self, output):
            # batch entry 1 of the inference dictionary is the directed pair (image0, image1)
            self.imgs = [np.zeros(tuple(output[v]["img"][1].shape[-2:]) + (3,), dtype=np.float32) for v in ("view1", "view2")]
            self._pts = [output["pred1"]["pts3d"][1], output["pred2"]["pts3d_in_other_view"][1]]
            c = [output["pred1"]["conf"][1], output["pred2"]["conf"][1]]
            self._masks = [ci > ci.median() for ci in c]

        def get_masks(self):
            return self._masks

        def get_pts3d(self):
            return self._pts

    def fake_forward_sizes(self, packed, net_cfg, images, pairs, dump=False, arith=0):
        norm = [(im.reshape((1,) + tuple(im.shape[-3:])) - 0.5) / 0.5 for im in images]
        res = [ora.forward(norm[a], norm[b]) for a, b in pairs]
        return {"pts3d": [[r[0]["pts3d"][0] for r in res], [r[1]["pts3d_in_other_view"][0] for r in res]],
                "conf": [[r[0]["conf"][0] for r in res], [r[1]["conf"][0] for r in res]]}

    monkeypatch.setattr(backend.DUSt3RHIP, "forward", fake_forward)
    monkeypatch.setattr(backend.DUSt3RHIP, "forward_sizes", fake_forward_sizes)
    monkeypatch.setattr(Duster, "aligner", staticmethod(lambda output, device: Scene(output)))
    model = Duster({"state_dict": sd, "max_keypoints": 40}).eval()
    g = torch.Generator().manual_seed(42)
    i0, i1 = torch.rand(1, 3, 48, 64, generator=g), torch.rand(1, 3, 48, 64, generator=g)
    pred = model({"image0": i0, "image1": i1})
    sc = Scene(model.inference_output({"image0": i0, "image1": i1}))
    k0, k1 = duster_matches_from_scene(sc.imgs, [m.numpy() for m in sc.get_masks()], [p.numpy() for p in sc.get_pts3d()], 40)
    assert pred["keypoints0"].shape == pred["keypoints1"].shape and pred["keypoints0"].shape[1] == 2
    assert 5 < len(k0) <= 40
    assert np.array_equal(pred["keypoints0"].numpy(), k0) and np.array_equal(pred["keypoints1"].numpy(), k1)
    # unlimited: every reciprocal pair, no sub-sampling
    model.conf["max_keypoints"] = None
    full = model({"image0": i0, "image1": i1})
    f0, f1 = duster_matches_from_scene(sc.imgs, [m.numpy() for m in sc.get_masks()], [p.numpy() for p in sc.get_pts3d()], None)
    assert np.array_equal(full["keypoints0"].numpy(), f0) and np.array_equal(full["keypoints1"].numpy(), f1) and len(f0) >= len(k0)
    # building blocks: grid convention and the reciprocity rule on a hand-made case
    assert xy_grid(3, 2).tolist() == [[[0, 0], [1, 0], [2, 0]], [[0, 1], [1, 1], [2, 1]]]
    P1 = np.array([[0.0, 0, 0], [10, 0, 0], [5, 5, 0]])
    P2 = np.array([[0.1, 0, 0], [9, 0, 0], [9.5, 0, 0], [100, 0, 0]])
    rec, nn, n = find_reciprocal_matches(P1, P2)
    assert nn.tolist() == [0, 1, 1, 1] and rec.tolist() == [True, False, True, False] and n == 2
    # an empty second cloud is the reference's "Matched 0 points" branch
    empty = model.matches_from_scene(sc.imgs, [sc.get_masks()[0], torch.zeros_like(sc.get_masks()[1])], sc.get_pts3d())
    assert empty["keypoints0"].shape == (0, 2) and empty["keypoints1"].shape == (0, 2)
    # images of two sizes (the reference's drivers resize each image on its own): lists of per-pair maps, as upstream collates them;
    # key-points of image0 / image1 live on their own pixel grids
    j1 = torch.rand(1, 3, 64, 80, generator=g)
    out2 = model.inference_output({"image0": i0, "image1": j1})
    assert [tuple(m.shape) for m in out2["pred1"]["pts3d"]] == [(64, 80, 3), (48, 64, 3)]
    assert [tuple(m.shape) for m in out2["pred2"]["pts3d_in_other_view"]] == [(48, 64, 3), (64, 80, 3)]
    sc2 = Scene(out2)
    assert [im.shape for im in sc2.imgs] == [(48, 64, 3), (64, 80, 3)]
    model.conf["max_keypoints"] = 40
    pred2 = model({"image0": i0, "image1": j1})
    m0, m1 = duster_matches_from_scene(sc2.imgs, [m.numpy() for m in sc2.get_masks()], [p.numpy() for p in sc2.get_pts3d()], 40)
    assert np.array_equal(pred2["keypoints0"].numpy(), m0) and np.array_equal(pred2["keypoints1"].numpy(), m1) and len(m0) > 5
    assert pred2["keypoints0"][:, 0].max() < 64 and pred2["keypoints0"][:, 1].max() < 48
    assert pred2["keypoints1"][:, 0].max() < 80 and pred2["keypoints1"][:, 1].max() < 64
    with pytest.raises(ValueError, match="multiples of the patch size"):
        model({"image0": i0, "image1": torch.rand(1, 3, 60, 64)})
