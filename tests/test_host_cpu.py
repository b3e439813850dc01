"""Host-side logic that needs no GPU: the C-ABI library loads and exports every declared
symbol, weight packing is a pure permutation, plugins honour the reference seam."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_header_symbol_is_exported_and_bound(lib):
    from imcui_hip.lib_loader import SIGNATURES

    hdr = open(os.path.join(ROOT, "include", "imcui_hip.h")).read()
    declared = set(re.findall(r"\b(imcui_hip_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/imcui_hip.h but not exported"
        assert name in SIGNATURES, f"{name} has no ctypes signature"
    assert lib.imcui_hip_version() >= 100


def test_superpoint_packing_is_a_permutation(lib):
    from imcui_hip import backend

    sd = superpoint_state_dict(1)
    packed = backend.pack_superpoint(sd).numpy()
    n32 = 1300928  # f32 region (weights + biases, 64-float aligned); f16 split planes follow
    assert np.isclose(np.abs(packed[:n32]).sum(dtype=np.float64), sum(v.abs().double().sum().item() for v in sd.values()), rtol=1e-9)
    # spot-check the conv3x3 layout: [ch][tap][cq][cout][4]
    w = sd["conv1b.weight"].numpy()
    off = 9 * 64 + 64  # conv1a weights (576) + bias (64), both already 64-aligned
    blk = packed[off : off + 64 * 64 * 9].reshape(2, 9, 8, 64, 4)
    assert blk[1, 5, 3, 17, 2] == w[17, 32 + 3 * 4 + 2, 5 // 3, 5 % 3]


def test_lightglue_packing_deinterleaves_qkv(lib):
    from imcui_hip import backend

    lsd = lightglue_state_dict(2)
    names = backend.lightglue_tensor_names()
    assert len(names) == 251 and names[0] == "posenc.Wr.weight" and set(names) <= set(lsd)
    packed = backend.pack_lightglue(lsd).numpy()
    n32 = 11851712  # f32 region; f16 split planes follow
    # out_proj / to_out are folded into the right half of ffn.0 at pack time: W1b' = W1b @ Wo, b1' = b1 + W1b @ bo
    expect = dict(lsd)
    for i in range(9):
        for blk, proj in (("self_attn", "out_proj"), ("cross_attn", "to_out")):
            pre = f"transformers.{i}.{blk}."
            w1, b1 = lsd[pre + "ffn.0.weight"].double(), lsd[pre + "ffn.0.bias"].double()
            wo, bo = lsd[pre + proj + ".weight"].double(), lsd[pre + proj + ".bias"].double()
            expect[pre + "ffn.0.weight"] = torch.cat([w1[:, :256], w1[:, 256:] @ wo], 1).float()
            expect[pre + "ffn.0.bias"] = (b1 + w1[:, 256:] @ bo).float()
    assert np.isclose(np.abs(packed[:n32]).sum(dtype=np.float64), sum(expect[n].abs().double().sum().item() for n in names), rtol=1e-9)
    # Wqkv rows of layer 0: packed row t*256 + h*64 + d  <-  upstream row h*192 + d*3 + t
    w = lsd["transformers.0.self_attn.Wqkv.weight"].numpy()
    base = 64
    q_row = packed[base + (1 * 256 + 2 * 64 + 5) * 256 : base + (1 * 256 + 2 * 64 + 5) * 256 + 256]
    assert np.array_equal(q_row, w[2 * 192 + 5 * 3 + 1])
    # old-style checkpoint keys are accepted (renamed on load upstream)
    old = {}
    for k, v in lsd.items():
        m = re.match(r"transformers\.(\d+)\.(self_attn|cross_attn)\.(.*)", k)
        old[f"{m.group(2)}.{m.group(1)}.{m.group(3)}" if m else k] = v
    assert np.array_equal(backend.pack_lightglue(old).numpy(), packed)


def test_host_weight_split_is_accurate(lib):
    """w * 2^e == hi + lo to ~2^-22 relative, hi/lo valid f16, scale a power of two."""
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(64, 32, 3, 3, generator=g) * 0.07).numpy()
    hi = np.zeros(w.size, dtype=np.uint16)
    lo = np.zeros(w.size, dtype=np.uint16)
    inv = lib.imcui_hip_conv3x3_pack_split(w.ctypes.data, 64, 32, hi.ctypes.data, lo.ctypes.data)
    assert inv > 0 and np.log2(inv) == np.round(np.log2(inv))
    rec = (hi.view(np.float16).astype(np.float64) + lo.view(np.float16).astype(np.float64)) * inv
    # layout [(ch*9+tap)*4+oc][cout][8]  <-  w[co][oc*8+j][tap]
    ref = w.reshape(64, 1, 4, 8, 9).transpose(1, 4, 2, 0, 3).reshape(-1).astype(np.float64)
    assert np.abs(rec - ref).max() <= np.abs(ref).max() * 2.0**-21
    assert np.isfinite(hi.view(np.float16)).all() and np.isfinite(lo.view(np.float16)).all()


def test_linear_split_planes_are_fragment_major(lib):
    """imcui_hip_linear_pack_split: element (nf, ks, h, r, j) of a plane is W[32 nf + r][16 ks + 8 h + j] (one 1 KiB
    block = the 64-lane MFMA operand fragment of a wave), rows >= N are zero, hi + lo reconstructs w * 2^e."""
    from imcui_hip import backend

    g = torch.Generator().manual_seed(3)
    N, K = 65, 96  # three row fragments (the last one holds a single row), six k-steps
    w = torch.randn(N, K, generator=g) * 0.3
    hi, lo, inv = backend.pack_linear_split(w)
    assert inv > 0 and np.log2(inv) == np.round(np.log2(inv))
    rec = (hi.view(np.float16).astype(np.float64) + lo.view(np.float16).astype(np.float64)) * inv
    rec = rec.reshape(3, K // 16, 2, 32, 8)  # [nf][ks][h][r][j]
    full = rec.transpose(0, 3, 1, 2, 4).reshape(96, K)  # rows 32 nf + r, columns 16 ks + 8 h + j
    assert np.abs(full[:N] - w.numpy().astype(np.float64)).max() <= np.abs(w.numpy()).max() * 2.0**-21
    assert np.all(full[N:] == 0.0)
    assert np.isfinite(hi.view(np.float16)).all() and np.isfinite(lo.view(np.float16)).all()
    with pytest.raises(Exception):
        backend.pack_linear_split(torch.zeros(8, 24))  # K % 16 != 0


def test_header_is_c99_and_library_binds_from_plain_c(lib, tmp_path):
    """The boundary is a C ABI (plain pointers and sizes): a C99 program compiled against include/imcui_hip.h
    dlopens the library and calls the entry points that need no GPU."""
    import subprocess

    from imcui_hip.build import LIB_PATH

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_abi_smoke"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_abi_smoke.c"),
                    "-o", str(exe), "-ldl"], check=True)  # fmt: skip
    out = subprocess.run([str(exe), LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "version=400" in out and "lg_tensors=251" in out and "first=posenc.Wr.weight" in out, out


def test_plugins_follow_the_reference_seam(lib):
    import imcui_hip.hloc.extractors as extractors
    import imcui_hip.hloc.matchers as matchers
    from imcui_hip.hloc.utils.base_model import BaseModel, dynamic_load

    SP = dynamic_load(extractors, "superpoint")
    LG = dynamic_load(matchers, "lightglue")
    NN = dynamic_load(matchers, "nearest_neighbor")
    DS = dynamic_load(matchers, "dual_softmax")
    assert issubclass(SP, BaseModel) and issubclass(LG, BaseModel) and issubclass(NN, BaseModel) and issubclass(DS, BaseModel)
    assert DS.default_conf == {"match_threshold": 0.2, "inv_temperature": 20} and DS.required_inputs == ["descriptors0", "descriptors1"]
    empty = DS({})({"descriptors0": torch.zeros(1, 128, 5), "descriptors1": torch.zeros(1, 128, 0)})  # no GPU needed
    assert empty["matches0"].shape == (1, 128) and int(empty["matches0"].max()) == -1  # the reference's (B, C) quirk
    # default_conf of the reference wrappers (superpoint.py:34-41, lightglue.py:15-25)
    assert SP.default_conf["nms_radius"] == 4 and SP.default_conf["max_keypoints"] == -1 and SP.detection_noise == 2.0
    assert LG.default_conf["depth_confidence"] == 0.95 and LG.default_conf["width_confidence"] == 0.99
    sp = SP({"max_keypoints": 100, "state_dict": superpoint_state_dict(0)})
    assert sp.conf["max_keypoints"] == 100 and sp.conf["nms_radius"] == 4 and "state_dict" not in sp.conf
    # packed weights are a registered buffer (the UI model cache sums buffers and calls .to())
    assert sum(b.numel() for b in sp.buffers()) >= 1300865
    lg = LG({"match_threshold": 0.3, "state_dict": lightglue_state_dict(0)})
    assert lg.conf["filter_threshold"] == 0.3
    with pytest.raises(AssertionError):
        lg({"image0": torch.zeros(1, 1, 8, 8)})  # missing required inputs
    # no CPU fallback: the product path must fail loudly without a ROCm device tensor
    from imcui_hip import ImcuiHipError

    with pytest.raises(ImcuiHipError):
        sp({"image": torch.zeros(1, 1, 64, 64)})
    nn_model = NN({})
    out = nn_model({"descriptors0": torch.zeros(1, 128, 5), "descriptors1": torch.zeros(1, 128, 0)})
    assert (out["matches0"] == -1).all()
    # SuperGlue wrapper (imcui/hloc/matchers/superglue.py:14-29)
    from imcui_hip.synth_weights import superglue_state_dict

    SG = dynamic_load(matchers, "superglue")
    assert issubclass(SG, BaseModel)
    assert SG.default_conf == {"weights": "outdoor", "model_name": "superglue_outdoor.pth", "sinkhorn_iterations": 100, "match_threshold": 0.2}
    assert SG.required_inputs == ["image0", "keypoints0", "scores0", "descriptors0", "image1", "keypoints1", "scores1", "descriptors1"]
    sg = SG({"sinkhorn_iterations": 50, "state_dict": superglue_state_dict(0)})
    assert sg.conf["sinkhorn_iterations"] == 50 and sg.conf["match_threshold"] == 0.2 and "state_dict" not in sg.conf
    assert sum(b.numel() for b in sg.buffers()) >= 12_000_000  # ~12 M parameters + their split planes
    img = torch.zeros(1, 1, 48, 64)
    data = {"image0": img, "image1": img, "keypoints0": torch.zeros(1, 5, 2), "keypoints1": torch.zeros(1, 0, 2), "scores0": torch.zeros(1, 5),
            "scores1": torch.zeros(1, 0), "descriptors0": torch.zeros(1, 256, 5), "descriptors1": torch.zeros(1, 256, 0)}  # fmt: skip
    out = sg(data)  # no key-points: the upstream early return needs no GPU
    assert out["matches0"].dtype == torch.int32 and (out["matches0"] == -1).all() and out["matches1"].shape == (1, 0)
    data["keypoints1"], data["scores1"], data["descriptors1"] = torch.zeros(1, 4, 2), torch.zeros(1, 4), torch.zeros(1, 256, 4)
    with pytest.raises(ImcuiHipError):
        sg(data)  # CPU tensors: no fallback


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import imcui_hip.lib_loader as ll

    monkeypatch.setattr(ll, "_lib", None)
    with pytest.raises(ll.ImcuiHipError):
        ll.load_library(str(tmp_path / "nope.so"))


def _sg_layout_offsets():
    """Python mirror of csrc/superglue.hip sg_layout() (f32 region only)."""
    off = 0

    def take(n):
        nonlocal off
        r = off
        off += (n + 63) // 64 * 64
        return r

    kenc = [3, 32, 64, 128, 256, 256]
    lay = {"k0": take(32 * 4), "kw": [], "kb": [], "L": []}
    for i in range(4):
        lay["kw"].append(take(kenc[i + 2] * kenc[i + 1]))
        lay["kb"].append(take(kenc[i + 2]))
    for _ in range(18):
        lay["L"].append({k: take(n) for k, n in [("wqkv", 768 * 256), ("bqkv", 768), ("w1", 512 * 512), ("b1", 512), ("w2", 256 * 512), ("b2", 256)]})
    lay["wfinal"], lay["bfinal"], lay["bin"] = take(256 * 256), take(256), take(64)
    return lay


def test_superglue_packing_folds_bn_merge_and_heads(lib):
    """The packed SuperGlue weights (BatchNorm and attn.merge folded, heads de-interleaved) evaluated with plain
    torch reproduce the oracle: checks the host packer without a GPU."""
    from imcui_hip import backend
    from oracle.superglue import SuperGlueOracle, log_optimal_transport, normalize_keypoints
    from imcui_hip.synth_weights import superglue_state_dict

    torch.set_num_threads(4)
    sd = superglue_state_dict(3)
    names = backend.superglue_tensor_names()
    assert len(names) == 26 + 16 * 18 + 3 and names[0] == "kenc.encoder.0.weight" and names[-1] == "bin_score"
    assert set(names) == {k for k in sd if not k.endswith("num_batches_tracked")}
    P = backend.pack_superglue(sd)
    lay = _sg_layout_offsets()
    g = torch.Generator().manual_seed(0)
    n0, n1 = 150, 170
    k0 = torch.rand(1, n0, 2, generator=g) * torch.tensor([640.0, 480.0])
    k1 = torch.rand(1, n1, 2, generator=g) * torch.tensor([640.0, 480.0])
    d0 = torch.nn.functional.normalize(torch.randn(1, 256, n0, generator=g), dim=1)
    d1 = torch.nn.functional.normalize(torch.cat([d0[:, :, :100] + 0.05 * torch.randn(1, 256, 100, generator=g), torch.randn(1, 256, n1 - 100, generator=g)], 2), dim=1)
    s0, s1 = torch.rand(1, n0, generator=g), torch.rand(1, n1, generator=g)
    img = torch.zeros(1, 1, 480, 640)
    data = {"image0": img, "image1": img, "keypoints0": k0, "keypoints1": k1, "scores0": s0, "scores1": s1, "descriptors0": d0, "descriptors1": d1}
    ref = SuperGlueOracle(sd, {"sinkhorn_iterations": 20})(data)

    def mat(off, n, k):
        return P[off : off + n * k].view(n, k).double()

    def vec(off, n):
        return P[off : off + n].double()

    def enc(kp, sc, dsc):
        kn = normalize_keypoints(kp, img.shape)[0].double()
        w0 = mat(lay["k0"], 32, 4)
        e = torch.relu(kn @ w0[:, :2].T + sc[0].double()[:, None] * w0[:, 2] + w0[:, 3])
        dims = [32, 64, 128, 256, 256]
        for i in range(4):
            e = e @ mat(lay["kw"][i], dims[i + 1], dims[i]).T + vec(lay["kb"][i], dims[i + 1])
            if i < 3:
                e = torch.relu(e)
        return dsc[0].T.double() + e

    x = [enc(k0, s0, d0), enc(k1, s1, d1)]
    for li, L in enumerate(lay["L"]):
        qkv = [(xi @ mat(L["wqkv"], 768, 256).T + vec(L["bqkv"], 768)).view(-1, 3, 4, 64) for xi in x]
        new = []
        for a in range(2):
            src = a ^ (li & 1)
            q, k, v = qkv[a][:, 0], qkv[src][:, 1], qkv[src][:, 2]
            att = torch.softmax(torch.einsum("nhd,mhd->hnm", q, k) / 8.0, -1)
            ctx = torch.einsum("hnm,mhd->nhd", att, v).reshape(-1, 256)
            hid = torch.relu(torch.cat([x[a], ctx], 1) @ mat(L["w1"], 512, 512).T + vec(L["b1"], 512))
            new.append(x[a] + hid @ mat(L["w2"], 256, 512).T + vec(L["b2"], 256))
        x = new
    md = [xi @ mat(lay["wfinal"], 256, 256).T + vec(lay["bfinal"], 256) for xi in x]
    scores = (md[0] @ md[1].T / 16.0)[None].float()
    Z = log_optimal_transport(scores, P[lay["bin"]], 20)
    mx = Z[:, :-1, :-1].max(2)
    mutual = torch.arange(n0)[None] == Z[:, :-1, :-1].max(1).indices.gather(1, mx.indices)
    ms = torch.where(mutual, mx.values.exp(), torch.zeros(()))
    m0 = torch.where(mutual & (ms > 0.2), mx.indices, torch.tensor(-1))
    assert (ref["matches0"] > -1).sum() > 60
    assert torch.equal(m0, ref["matches0"])
    assert (ms - ref["matching_scores0"]).abs().max().item() < 1e-4


def test_eloftr_upstream_checkpoint_names_round_trip():
    """The zoo's `eloftr` entry downloads `eloftr_outdoor.ckpt` and loads `["state_dict"]` with the UPSTREAM module names
    (imcui/hloc/matchers/eloftr.py:56-61).  The plugin's name map must be a bijection onto the names the packer reads: port
    state dict -> upstream names (with the Lightning `matcher.` prefix, inside a {"state_dict": ...} container, plus keys a
    training checkpoint carries) -> plugin -> the SAME packed buffer as the port's own names give (missing #3a)."""
    import torch
    from transformers import EfficientLoFTRConfig, EfficientLoFTRForKeypointMatching

    from imcui_hip import backend
    from imcui_hip.hloc.matchers import eloftr as E

    torch.manual_seed(0)
    net = EfficientLoFTRForKeypointMatching(EfficientLoFTRConfig())
    with torch.no_grad():  # non-trivial BatchNorm statistics so that a swapped branch cannot hide
        for name, buf in net.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(torch.randn_like(buf) * 0.1)
            elif name.endswith("running_var"):
                buf.copy_(torch.rand_like(buf) + 0.5)
    sd = net.state_dict()
    up = E.port_to_upstream_names(sd)
    assert len(up) == len(sd) == 447 and all(k.startswith("matcher.") for k in up)
    assert "matcher.backbone.layer0.rbr_dense.conv.weight" in up and "matcher.backbone.layer3.13.rbr_identity.running_var" in up
    assert "matcher.loftr_coarse.layers.7.mlp.2.weight" in up and "matcher.fine_preprocess.layer1_outconv2.3.weight" in up
    back = E.upstream_to_port_names(up)
    assert set(back) == set(sd) and all(back[k] is sd[k] for k in sd)
    # as the reference loads it: a Lightning container with extra keys, names with and without the prefix
    ckpt = {"state_dict": {**up, "matcher.pos_encoding.sin": torch.zeros(3)}, "epoch": 29}
    want = backend.pack_eloftr(sd)
    for container in (ckpt, {k[len("matcher."):]: v for k, v in up.items()}):
        model_sd = container["state_dict"] if "state_dict" in container else container
        got = backend.pack_eloftr(E.to_port_names(model_sd))
        assert torch.equal(got, want)
    with pytest.raises(KeyError, match="unrecognised tensors"):
        E.upstream_to_port_names({**up, "matcher.backbone.layer9.rbr_dense.conv.weight": torch.zeros(1)})
    with pytest.raises(KeyError, match="must use the upstream names"):
        E.to_port_names({"foo.weight": torch.zeros(1)})


def test_lightglue_filter_threshold_is_frozen_at_init_like_the_reference():
    """imcui/hloc/matchers/lightglue.py:50-51 copies match_threshold into upstream's conf once (`LG(**conf)`); the UI's later mutation
    of a cached model's conf (imcui/ui/utils.py:921-922) never reaches `filter_matches`.  Default = that behaviour; the opt-in
    conf["runtime_match_threshold"] = True re-reads the conf on every call."""
    from imcui_hip.hloc.matchers.lightglue import LightGlue
    from imcui_hip.synth_weights import lightglue_state_dict

    seen = []

    class Spy:
        def forward(self, packed, k0, k1, d0, d1, n0, n1, s0, s1, depth, width, thr, **kw):
            seen.append(thr)
            return {}

    z = torch.zeros(1, 4, 2), torch.zeros(1, 4, 2), torch.zeros(1, 4, 256), torch.zeros(1, 4, 256), torch.zeros(1, dtype=torch.int32), torch.zeros(1, dtype=torch.int32)
    lg = LightGlue({"match_threshold": 0.3, "state_dict": lightglue_state_dict(0)})
    lg._impl = Spy()
    lg.forward_batched(*z, (640, 480), (640, 480))
    lg.conf["match_threshold"] = 0.05  # what run_matching does to a cached matcher
    lg.forward_batched(*z, (640, 480), (640, 480))
    assert seen == [0.3, 0.3]
    lg.conf["runtime_match_threshold"] = True
    lg.forward_batched(*z, (640, 480), (640, 480))
    assert seen[-1] == 0.05
    assert LightGlue.default_conf["runtime_match_threshold"] is False
