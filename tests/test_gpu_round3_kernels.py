"""Round-3 kernels against the round-2 kernels they replace and against float64 (GPU box only, through the C ABI).

* gemm_wreg_kernel (weights in registers) must be BITWISE equal to gemm_split_kernel: same K order, same three products per
  k-step, same epilogue expressions -- so every parity statement made for the old kernel carries over.
* (LightGlue's assignment ran on a materialised similarity here, with its statistics from a stand-alone pass or from the GEMM's epilogue;
  round 5 replaced both by the matrix-free kernel of csrc/simred.hip: tests/test_gpu_simred.py, tests/test_gpu_lightglue.py.)
* attention priority variants are scheduling only: bitwise equal outputs.
"""
import pytest
import torch

from imcui_hip.synth_weights import lightglue_state_dict
from parity_utils import synthetic_matching_problem

pytestmark = pytest.mark.gpu
LSD = lightglue_state_dict(0)


from imcui_hip import backend  # noqa: E402  (ctypes bindings only: importing it does not load the library)

dev = torch.device("cuda:0")  # the A/B switches live in the per-device handle: backend.option(dev, name=value)


@pytest.mark.parametrize("M,N,K,relu", [(300, 256, 256, False), (4800, 768, 256, True), (1000, 512, 512, False), (130, 256, 32, False),
                                       (2048, 128, 96, True), (129, 1024, 64, False), (5000, 256, 1152, False), (640, 3072, 1024, False)])
def test_gemm_wreg_bitwise_equals_split_kernel(M, N, K, relu):
    from imcui_hip import backend

    dev = torch.device("cuda:0")
    backend.set_precision(dev, 1)
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev)
    w = torch.randn(N, K, generator=g) * (1.0 / K**0.5) + torch.arange(N).float()[:, None] * 1e-3
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    with backend.option(dev, gemm_wreg=0):
        old = backend.linear_split_f32(a, w, b, relu).cpu()
    with backend.option(dev, gemm_wreg=2):
        new = backend.linear_split_f32(a, w, b, relu).cpu()
    ref = a.cpu().double() @ w.double().t() + b.cpu().double()
    ref = torch.relu(ref) if relu else ref
    assert (new.double() - ref).abs().max().item() / ref.abs().max().item() < 4e-6
    assert torch.equal(old, new), (old - new).abs().max().item()


def _batch(problems):
    B = len(problems)
    ncap = max(max(p[0].shape[0], p[1].shape[0]) for p in problems)
    k0, k1 = torch.zeros(B, ncap, 2), torch.zeros(B, ncap, 2)
    d0, d1 = torch.zeros(B, ncap, 256), torch.zeros(B, ncap, 256)
    n0, n1 = torch.zeros(B, dtype=torch.int32), torch.zeros(B, dtype=torch.int32)
    for b, (a, c, e, f) in enumerate(problems):
        k0[b, : len(a)], k1[b, : len(c)], d0[b, : len(a)], d1[b, : len(c)] = a, c, e, f
        n0[b], n1[b] = len(a), len(c)
    return [t.cuda() for t in (k0, k1, d0, d1, n0, n1)]


def _run(dc, wc, problems, dump=False):
    from imcui_hip.hloc.matchers.lightglue import LightGlue

    model = LightGlue({"depth_confidence": dc, "width_confidence": wc, "match_threshold": 0.1, "state_dict": LSD}).eval().to("cuda:0")
    out = model.forward_batched(*_batch(problems), (640, 480), (640, 480), layer_dump=dump)
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in out.items()}


PROBLEMS = [(700, 650, 150), (2048, 1900, 300), (130, 257, 30), (1024, 900, 300)]


@pytest.mark.parametrize("dc,wc", [(-1, -1), (0.95, 0.99)])
def test_lightglue_projection_kernels_agree(dc, wc):
    """QKV / cross projections on gemm_wreg_kernel vs gemm_split_kernel over a whole forward (ragged batch, pruning on and
    off): decisions identical, token states and scores to round-off (the rotary encoding is a different sequence of fused
    multiply-adds in the two kernels; everything else is bitwise equal, see test_attention_layout_projection_planes)."""
    problems = [synthetic_matching_problem(60 + i, n, m, o) for i, (n, m, o) in enumerate(PROBLEMS)]
    with backend.option(dev, gemm_wreg=0):
        old = _run(dc, wc, problems, dump=True)
    with backend.option(dev, gemm_wreg=2):
        new = _run(dc, wc, problems, dump=True)
    for b, (n, m, _) in enumerate(PROBLEMS):
        for li in range(old["_layers"].shape[0]):
            for s, cnt in enumerate((n, m)):
                if wc > 0:
                    continue  # pruned layouts: rows beyond the live count hold stale data, compared through the outputs below
                a, c = old["_layers"][li, 2 * b + s, :cnt], new["_layers"][li, 2 * b + s, :cnt]
                assert (a - c).abs().max().item() <= 2e-5 * a.abs().max().item(), (b, li, s)
    for k in ("matches0", "matches1", "stop", "prune0", "prune1"):
        assert torch.equal(old[k], new[k]), k
    for k in ("matching_scores0", "matching_scores1"):
        assert (old[k] - new[k]).abs().max().item() < 2e-5, k


@pytest.mark.parametrize("cross", [False, True])
def test_attention_layout_projection_planes(cross):
    """imcui_hip_qkv_split_f32 on both GEMM kernels at the bench's size (64 pairs x 2 x 2048 tokens): three runs of each
    kernel bitwise repeatable (this test caught a packed-f32 code-generation hazard in round 3), V^T and the un-rotated
    planes bitwise equal between the kernels, rotated q / k within 1e-6 of the magnitude, q against float64."""
    import numpy as np

    from imcui_hip import backend

    dev = torch.device("cuda:0")
    backend.set_precision(dev, 1)
    nseq, R = 128, 2048
    g = torch.Generator().manual_seed(11 + cross)
    x = torch.randn(nseq * R, 256, generator=g).to(dev)
    N = 512 if cross else 768
    w = torch.randn(N, 256, generator=g) / 16.0
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    ang = torch.rand(nseq * R, 32, generator=g) * 6.28
    cos, sin = torch.cos(ang).to(dev), torch.sin(ang).to(dev)
    cnt = torch.full((nseq,), R, dtype=torch.int32)
    cnt[5], cnt[6] = 1000, 0
    cnt = cnt.to(dev)
    res = {}
    for mode in (0, 2):
        with backend.option(dev, gemm_wreg=mode):
            runs = [backend.qkv_split_f32(x, w, b, cos, sin, cnt, R, 0.18, cross) for _ in range(3)]
        torch.cuda.synchronize()
        for r in runs[1:]:
            for a, c in zip(runs[0], r):
                assert torch.equal(a, c), f"mode {mode}: not repeatable"
        res[mode] = runs[0]

    def val(t):
        a = t.cpu().numpy().view(np.float16).astype(np.float64)
        return a[0] + a[1]

    assert torch.equal(res[0][2], res[2][2])  # V^T
    if cross:
        assert torch.equal(res[0][0], res[2][0])
    else:
        for i in (0, 1):
            a, c = val(res[0][i]), val(res[2][i])
            assert np.abs(a - c).max() <= 1e-6 * np.abs(a).max()
    ref = x[:256].cpu().double() @ w[:256].double().t() + b[:256].cpu().double()
    ref = ref.reshape(256, 4, 64)
    if not cross:
        c2, s2 = cos[:256].cpu().double(), sin[:256].cpu().double()
        e, o = ref[..., 0::2], ref[..., 1::2]
        ref = torch.stack((e * c2[:, None, :] - o * s2[:, None, :], o * c2[:, None, :] + e * s2[:, None, :]), -1).reshape(256, 4, 64)
    ref = (ref * 0.18).permute(1, 0, 2).numpy()
    got = val(res[2][0])[0, :, :256]
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-6


@pytest.mark.parametrize("cross", [False, True])
def test_projection_token_tiles_are_bitwise_equal(cross):
    """The weights-in-registers projection GEMM picks 128 / 64 / 32-token workgroup tiles by token count (option `wreg_tile`; one
    LightGlue pair no longer leaves two thirds of the CUs idle): every plane bitwise equal across the tiles, ragged counts included."""
    backend.set_precision(dev, 1)
    nseq, R = 6, 2048
    g = torch.Generator().manual_seed(23 + cross)
    x = torch.randn(nseq * R, 256, generator=g).to(dev)
    N = 512 if cross else 768
    w = torch.randn(N, 256, generator=g) / 16.0
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    ang = torch.rand(nseq * R, 32, generator=g) * 6.28
    cos, sin = torch.cos(ang).to(dev), torch.sin(ang).to(dev)
    cnt = torch.tensor([R, 1000, 0, 77, R, 1900], dtype=torch.int32).to(dev)
    res = {}
    for tile in (128, 64, 32, 0):
        with backend.option(dev, gemm_wreg=2, wreg_tile=tile):
            res[tile] = [t.clone() for t in backend.qkv_split_f32(x, w, b, cos, sin, cnt, R, 0.18, cross)]
        torch.cuda.synchronize()
    cn = cnt.cpu().tolist()
    for tile in (64, 32, 0):
        for i, (a, c) in enumerate(zip(res[128], res[tile])):
            # a tile is skipped when its first row is past the sequence's count, so the written rows past the count depend on the tile: live rows only
            if a.shape[-1] == 64:  # [2 planes, seq, head, R, 64]
                for z, n in enumerate(cn):
                    assert torch.equal(a[:, z, :, :n], c[:, z, :, :n]), (tile, i, z)
            else:  # V^T [2 planes, seq, head, 64, R]
                for z, n in enumerate(cn):
                    assert torch.equal(a[:, z, :, :, :n], c[:, z, :, :, :n]), (tile, i, z)


@pytest.mark.parametrize("M,N", [(300, 256), (4096, 768), (9000, 256)])
def test_plain_projection_token_tiles_are_bitwise_equal(M, N):
    backend.set_precision(dev, 1)
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, 256, generator=g).to(dev)
    w = torch.randn(N, 256, generator=g) / 16.0
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    outs = {}
    for tile in (128, 64, 32, 0):
        with backend.option(dev, gemm_wreg=2, wreg_tile=tile):
            outs[tile] = backend.linear_split_f32(a, w, b, False).cpu()
    for tile in (64, 32, 0):
        assert torch.equal(outs[tile], outs[128]), tile


def test_attention_priority_variants_bitwise():
    from imcui_hip import backend

    dev = torch.device("cuda:0")
    backend.set_precision(dev, 1)
    g = torch.Generator().manual_seed(9)
    S, Hh, R = 4, 4, 512
    cnt = torch.tensor([512, 400, 77, 300], dtype=torch.int32).to(dev)
    q = (torch.randn(S, Hh, R, 64, generator=g) * 0.5).to(dev)
    k = torch.randn(S, Hh, R, 64, generator=g).to(dev)
    v = torch.randn(S, Hh, R, 64, generator=g).to(dev)
    outs = []
    for var in (0, 1, 2, 3, 8):  # scheduling only: wave priorities, the pipelined K.Q^T of round 4 (the default)
        with backend.option(dev, attn_variant=var):
            outs.append(backend.attention_f32(q, k, v, cnt, True, True).cpu())
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


def test_split_range_check_reports_saturation():
    """the 3 x f16 split saturates silently above 65504.  With the opt-in range check on, a projection
    fed an activation of 1e5 sets bit 0, a NaN sets bit 1, ordinary data (a whole LightGlue forward with the strong weight set)
    leaves the word at 0; padding rows of ragged batches are not scanned."""
    from imcui_hip import backend

    dev = torch.device("cuda:0")
    backend.set_precision(dev, 1)
    backend.set_range_check(dev, True)
    try:
        assert backend.range_status(dev) == 0
        g = torch.Generator().manual_seed(3)
        a = torch.randn(300, 256, generator=g).to(dev)
        w = torch.randn(256, 256, generator=g) / 16.0
        backend.linear_split_f32(a, w, None)
        assert backend.range_status(dev) == 0
        a[17, 5] = 1.0e5
        backend.linear_split_f32(a, w, None)
        assert backend.range_status(dev) == 1
        assert backend.range_status(dev) == 0  # reading clears the word
        a[17, 5] = float("nan")
        backend.linear_split_f32(a, w, None)
        assert backend.range_status(dev) == 2
        problems = [synthetic_matching_problem(80 + i, n, m, o) for i, (n, m, o) in enumerate(PROBLEMS)]
        out = _run(-1, -1, problems)
        assert (out["matches0"] > -1).sum() > 100
        assert backend.range_status(dev) == 0
    finally:
        backend.set_range_check(dev, False)
