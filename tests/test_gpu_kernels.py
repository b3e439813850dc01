"""Building-block kernels vs plain PyTorch fp32 (GPU box only, through the C ABI)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "needs the MI355X"
    return torch.device("cuda:0")


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize("M,N,K,relu", [(256, 256, 256, False), (1000, 65, 256, False), (4800, 768, 256, True), (130, 130, 512, False)])
def test_linear_f32(M, N, K, relu, precision):
    from imcui_hip import backend

    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, K, generator=g)
    # asymmetric operands (row/col dependent) so a transposed write cannot pass
    w = torch.randn(N, K, generator=g) + torch.arange(N).float()[:, None] * 0.01
    b = torch.randn(N, generator=g)
    ref = F.linear(a.double(), w.double(), b.double())
    ref = (F.relu(ref) if relu else ref).float()
    out = backend.linear_f32(a.to(_dev()), w.to(_dev()), b.to(_dev()), relu).cpu()
    assert out.shape == ref.shape
    assert _rel(out, ref) < (2e-6 if precision == 0 else 4e-6)


@pytest.mark.parametrize("M,N,K,relu", [(300, 65, 256, False), (4800, 768, 256, True), (1000, 512, 512, False), (130, 256, 32, False),
                                       (2048, 128, 96, True), (129, 1024, 64, False), (5000, 256, 1152, False)])
def test_linear_split_prepacked_weights(M, N, K, relu):
    """The GEMM path of every network projection in the split mode: fragment-major pre-split weights, 128-column
    tiles (registers) for N % 256 != 0 and 256-column tiles (LDS-DMA) otherwise, odd and single K-tile counts,
    ragged M / N edges."""
    from imcui_hip import backend

    dev = torch.device("cuda:0")
    backend.set_precision(dev, 1)
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * (1.0 / K**0.5)
    b = torch.randn(N, generator=g) * 0.1
    ref = a.double() @ w.double().t() + b.double()
    ref = torch.relu(ref) if relu else ref
    out = backend.linear_split_f32(a.to(dev), w, b.to(dev), relu).cpu()
    assert out.shape == ref.shape
    assert (out.double() - ref).abs().max().item() / ref.abs().max().item() < 4e-6
    out2 = backend.linear_split_f32(a.to(dev), w, b.to(dev), relu).cpu()
    assert torch.equal(out, out2)  # bitwise repeatable


@pytest.mark.parametrize("B,H,W,Cin,Cout,pool", [(2, 16, 32, 64, 64, False), (1, 24, 40, 64, 128, True), (2, 60, 80, 128, 256, False), (1, 15, 21, 32, 64, False)])
def test_conv3x3_f32(B, H, W, Cin, Cout, pool, precision):
    from imcui_hip import backend

    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    if pool:
        ref = F.max_pool2d(ref, 2, 2)
    ref = ref.float().permute(0, 2, 3, 1).contiguous()
    out = backend.conv3x3_f32(x.permute(0, 2, 3, 1).contiguous().to(_dev()), w, b, relu=True, pool=pool).cpu()
    assert out.shape == ref.shape
    assert _rel(out, ref) < (2e-6 if precision == 0 else 4e-6)


@pytest.mark.parametrize("B,H,W,Cin,Cout,pool", [(2, 60, 80, 128, 128, False), (1, 120, 160, 64, 128, True), (2, 60, 80, 128, 256, False), (8, 120, 160, 128, 128, False)])
def test_conv3x3_channel_tiles_are_bitwise_equal(B, H, W, Cin, Cout, pool):
    """A launch whose 128-output-channel tiling gives fewer than 256 workgroups runs 64-channel tiles instead (a single pair's
    SuperPoint layers at 1/4 and 1/8 resolution; option `conv_narrow`: 0 = that rule, 1 = always, 2 = never): same arithmetic per output."""
    from imcui_hip import backend

    backend.set_precision(_dev(), 1)
    g = torch.Generator().manual_seed(H + W + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).to(_dev())
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    outs = {}
    for mode in (2, 1, 0):
        with backend.option(_dev(), conv_narrow=mode):
            outs[mode] = backend.conv3x3_f32(x, w, b, relu=True, pool=pool).cpu()
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("cross", [False, True])
def test_attention_f32(cross, precision):
    from imcui_hip import backend

    g = torch.Generator().manual_seed(5)
    S, Hh, R = 4, 4, 256
    cnt = torch.tensor([256, 200, 77, 130], dtype=torch.int32)
    q = torch.randn(S, Hh, R, 64, generator=g) * 0.5
    k = torch.randn(S, Hh, R, 64, generator=g)
    v = torch.randn(S, Hh, R, 64, generator=g)
    # poison the padding rows: they must never leak into valid outputs
    for s in range(S):
        k[s, :, cnt[s]:] = float("nan")
        v[s, :, cnt[s]:] = float("inf")
    out = backend.attention_f32(q.to(_dev()), k.to(_dev()), v.to(_dev()), cnt.to(_dev()), cross).cpu().view(S, R, Hh, 64)
    for s in range(S):
        ks = s ^ 1 if cross else s
        nq, nk = int(cnt[s]), int(cnt[ks])
        att = torch.softmax(q[s, :, :nq].double() @ k[ks, :, :nk].double().transpose(-1, -2), -1)
        ref = (att @ v[ks, :, :nk].double()).float().permute(1, 0, 2)  # [nq, heads, 64]
        got = out[s, :nq]
        assert torch.isfinite(got).all()
        assert (got - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("log2_domain", [False, True], ids=["natural-log", "base-2"])
def test_attention_sharp_softmax(log2_domain, precision):
    """Large logits: one key dominates; exercises the running-max rescale across tiles (base 2: the deferred-maximum path --
    a spike far above the reference must take the rescale branch, growth below 1.5 must not change the result)."""
    from imcui_hip import backend

    g = torch.Generator().manual_seed(6)
    S, Hh, R = 2, 4, 384
    cnt = torch.tensor([384, 300], dtype=torch.int32)
    q = torch.randn(S, Hh, R, 64, generator=g) * 3.0
    k = torch.randn(S, Hh, R, 64, generator=g) * 3.0
    k[:, :, 290] *= 4.0  # spike in the last tile
    v = torch.randn(S, Hh, R, 64, generator=g)
    out = backend.attention_f32(q.to(_dev()), k.to(_dev()), v.to(_dev()), cnt.to(_dev()), False, log2_domain).cpu().view(S, R, Hh, 64)
    for s in range(S):
        n = int(cnt[s])
        att = torch.softmax(q[s, :, :n].double() @ k[s, :, :n].double().transpose(-1, -2), -1)
        ref = (att @ v[s, :, :n].double()).float().permute(1, 0, 2)
        # logits reach several hundred: fp32 round-off of q.k alone is ~1e-4 relative in P
        assert (out[s, :n] - ref).abs().max().item() < 5e-4


@pytest.mark.parametrize("r", [0, 1, 3, 4])
def test_simple_nms_bit_exact(r):
    from imcui_hip import backend
    from oracle.superpoint import simple_nms

    g = torch.Generator().manual_seed(r)
    s = torch.rand(2, 100, 141, generator=g)
    s[0, 10:20, 10:30] = 0.5  # plateau of exact ties
    s[1] = (s[1] * 8).round() / 8  # heavy quantisation -> many ties
    ref = simple_nms(s, r)
    out = backend.simple_nms(s.to(_dev()), r).cpu()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("B,H,W", [(1, 480, 640), (3, 30, 44), (2, 1, 4)])
def test_rgb_to_gray_bit_exact(B, H, W):
    """Step before the path: device grey conversion + /255 vs the CPU restatement, bit for bit."""
    import numpy as np

    from imcui_hip import backend
    from oracle.preprocess import preprocess_gray

    rng = np.random.default_rng(B * 1000 + H + W)
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    img[0, 0, :4] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 0, 255]][: min(4, W)]
    out = backend.rgb_to_gray(torch.from_numpy(img).cuda()).cpu().numpy()
    ref = preprocess_gray(img)
    assert out.shape == ref.shape and np.array_equal(out, ref)


@pytest.mark.parametrize("log2_domain", [False, True], ids=["natural-log", "base-2"])
@pytest.mark.parametrize("cross", [False, True])
def test_attention_long_ragged_sequences(cross, log2_domain, precision):
    """The bench shape (2048 rows per sequence, 32 key tiles: the steady-state, hand-interleaved iterations of
    attn_split_pipe_kernel run 14 times) with ragged counts -- full, one key short of a tile, one key into a tile, a
    single tile, odd and even tile counts -- poisoned padding and a spiked key that moves the running maximum late."""
    from imcui_hip import backend

    g = torch.Generator().manual_seed(9)
    S, Hh, R = 6, 4, 2048
    cnt = torch.tensor([2048, 1999, 1025, 64, 1984, 1857], dtype=torch.int32)
    q = torch.randn(S, Hh, R, 64, generator=g) * 0.6
    k = torch.randn(S, Hh, R, 64, generator=g)
    v = torch.randn(S, Hh, R, 64, generator=g)
    k[:, :, 1000] *= 3.0  # late spike: the O / l rescale branch fires in the middle of the sequence
    for s in range(S):
        k[s, :, cnt[s]:] = float("nan")
        v[s, :, cnt[s]:] = float("inf")
    out = backend.attention_f32(q.to(_dev()), k.to(_dev()), v.to(_dev()), cnt.to(_dev()), cross, log2_domain).cpu().view(S, R, Hh, 64)
    for s in range(S):
        ks = s ^ 1 if cross else s
        nq, nk = int(cnt[s]), int(cnt[ks])
        att = torch.softmax(q[s, :, :nq].double() @ k[ks, :, :nk].double().transpose(-1, -2), -1)
        ref = (att @ v[ks, :, :nk].double()).float().permute(1, 0, 2)
        got = out[s, :nq]
        assert torch.isfinite(got).all(), f"sequence {s}"
        assert (got - ref).abs().max().item() < 3e-5, f"sequence {s}: {(got - ref).abs().max().item():.2e}"


@pytest.mark.parametrize("log2_domain", [False, True], ids=["natural-log", "base-2"])
def test_attention_valid_rows_do_not_depend_on_padding(log2_domain, precision):
    """The rows past a sequence's count are unwritten capacity.  What they hold (zeros, huge values, another batch's
    tokens) must not change a single bit of the valid rows: a pair matched alone and the same pair inside a batch go
    through different capacities.  The deferred-maximum branch of the base-2 kernel is wave-wide, so this is the case
    that catches a decision leaking from a padded query lane into its neighbours."""
    from imcui_hip import backend

    g = torch.Generator().manual_seed(21)
    S, Hh, R = 2, 4, 512
    cnt = torch.tensor([300, 211], dtype=torch.int32)  # both end inside a 32-query wave
    q = torch.randn(S, Hh, R, 64, generator=g) * 0.7
    k = torch.randn(S, Hh, R, 64, generator=g)
    v = torch.randn(S, Hh, R, 64, generator=g)
    k[:, :, 150] *= 2.5  # a late spike close to the deferral threshold for many queries
    outs = []
    for fill in (0.0, 40.0, -40.0):
        qq, kk, vv = q.clone(), k.clone(), v.clone()
        for s in range(S):
            qq[s, :, cnt[s]:] = fill * torch.randn(R - int(cnt[s]), 64, generator=g) if fill else 0.0
            kk[s, :, cnt[s]:] = fill
            vv[s, :, cnt[s]:] = fill
        o = backend.attention_f32(qq.to(_dev()), kk.to(_dev()), vv.to(_dev()), cnt.to(_dev()), False, log2_domain).cpu().view(S, R, Hh, 64)
        outs.append(torch.cat([o[s, : int(cnt[s])].reshape(-1) for s in range(S)]))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def _wide_range(shape, g, lo_exp=-7.0, hi_exp=4.7):
    """Random signs, magnitudes log-uniform over many decades (1e-7 .. 5e4): what real checkpoints feed the kernels --
    post-ReLU feature maps with a few very large channels next to near-zero ones."""
    mag = 10.0 ** (torch.rand(shape, generator=g) * (hi_exp - lo_exp) + lo_exp)
    return mag * (torch.randint(0, 2, shape, generator=g).float() * 2 - 1)


def test_split_mode_dynamic_range():
    """The 3 x f16 split keeps a value as hi = f16(x) + lo = f16(x - hi): exact to 2^-22 relative while lo is a normal
    f16, and to an ABSOLUTE 3e-8 (half the smallest f16 subnormal) once |x| < ~1e-3 makes lo subnormal; above 65504 hi
    saturates and the value keeps f16-class precision up to 2 x 65504.  Activations are not rescaled (weights are), so this
    checks the matrix kernels on inputs spanning 1e-7 .. 5e4 -- far beyond the seeded synthetic data -- against fp64:
    the error stays at fp32 round-off level relative to the magnitude of the outputs."""
    from imcui_hip import backend

    dev = _dev()
    backend.set_precision(dev, 1)
    g = torch.Generator().manual_seed(77)
    # GEMM, pre-split weights (every network projection)
    a = _wide_range((384, 256), g)
    w = torch.randn(512, 256, generator=g) / 16.0
    ref = a.double() @ w.double().t()
    out = backend.linear_split_f32(a.to(dev), w, None).cpu()
    assert (out - ref.float()).abs().max().item() / ref.abs().max().item() < 4e-6
    # small-magnitude rows must not be flushed: compare them on their own scale
    tiny = a.abs().max(dim=1).values.argmin()
    a2 = a.clone()
    a2[tiny] = _wide_range((256,), g, -7.0, -4.0)  # a row entirely below 1e-4: its lo parts are all subnormal
    ref2 = a2[tiny].double() @ w.double().t()
    out2 = backend.linear_split_f32(a2.to(dev), w, None).cpu()[tiny]
    assert (out2 - ref2.float()).abs().max().item() < 64 * 3e-8 * 256 ** 0.5  # absolute bound from the subnormal lo parts
    # both operands activations (similarity products): f32 weights split on the fly
    b = _wide_range((320, 256), g, -5.0, 2.5)
    ref = a.double() @ b.double().t()
    out = backend.linear_f32(a.to(dev), b.to(dev), None).cpu()
    assert (out - ref.float()).abs().max().item() / ref.abs().max().item() < 4e-6
    # 3x3 convolution
    x = _wide_range((1, 64, 24, 40), g, -6.0, 3.5).abs()  # post-ReLU like
    wc = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    bc = torch.randn(64, generator=g)
    ref = F.relu(F.conv2d(x.double(), wc.double(), bc.double(), padding=1)).float().permute(0, 2, 3, 1)
    out = backend.conv3x3_f32(x.permute(0, 2, 3, 1).contiguous().to(dev), wc, bc, relu=True, pool=False).cpu()
    assert (out - ref).abs().max().item() / ref.abs().max().item() < 4e-6
    # beyond the f16 maximum the split degrades gracefully, it does not overflow: hi saturates at 65504 and lo takes the
    # rest with f16 precision, so values up to 2 x 65504 stay finite and accurate to ~2^-11 (no network on this path
    # produces such activations; the soft-max numbers are bounded by 2^15.5 by construction)
    big = torch.full((128, 256), 0.0)
    big[:, 0] = 1.2e5
    big[:, 1] = -9.0e4
    ref = big.double() @ w.double().t()
    out = backend.linear_split_f32(big.to(dev), w, None).cpu()
    assert torch.isfinite(out).all() and (out - ref.float()).abs().max().item() / ref.abs().max().item() < 1e-3


def _ffn_reference(x, ctx, w1, b1, gamma, beta, w2, b2, act=None):
    h = torch.cat([x, ctx], -1).double() @ w1.double().t() + b1.double()
    if act in (2, 3):  # the dense matchers: activation, second Linear, LayerNorm(256), residual
        h = F.leaky_relu(h, 0.01) if act == 2 else torch.relu(h)
        y = h @ w2.double().t() + b2.double()
        return x.double() + F.layer_norm(y, (256,), gamma.double(), beta.double(), 1e-5)
    if gamma is None:
        h = torch.relu(h)
    else:
        h = torch.nn.functional.gelu(torch.nn.functional.layer_norm(h, (512,), gamma.double(), beta.double(), 1e-5))
    return x.double() + h @ w2.double().t() + b2.double()


@pytest.mark.parametrize("M,scale", [(128, 1.0), (1024, 1.0), (4096, 6.0)])
def test_fused_ffn_vs_fp64(M, scale):
    """The LightGlue FFN as one kernel (GEMM 512x512 -> LayerNorm -> GELU -> GEMM 256x512 -> residual) against float64
    torch: upstream lightglue.py TransformerLayer.ffn on cat([x, message]).  `scale` widens the hidden activations so
    that the GELU polynomial is exercised over its whole range (|y| up to ~10)."""
    from imcui_hip import backend

    backend.set_precision(_dev(), 1)
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, 256, generator=g)
    ctx = torch.randn(M, 256, generator=g) * 0.7
    w1 = torch.randn(512, 512, generator=g) / 512 ** 0.5
    b1 = torch.randn(512, generator=g) * 0.1
    gamma = (1.0 + 0.3 * torch.randn(512, generator=g)) * scale
    beta = torch.randn(512, generator=g) * 0.2 * scale
    w2 = torch.randn(256, 512, generator=g) / 512 ** 0.5
    b2 = torch.randn(256, generator=g) * 0.1
    ffn = backend.FusedFFN(w1, b1, gamma, beta, w2, b2, _dev())
    out = ffn(x.to(_dev()), ctx.to(_dev())).cpu()
    ref = _ffn_reference(x, ctx, w1, b1, gamma, beta, w2, b2)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    assert torch.isfinite(out).all() and err < 3e-6, err
    # in place (out aliases x), the way the LightGlue layers call it
    xd = x.to(_dev())
    ffn(xd, ctx.to(_dev()), out=xd)
    assert torch.equal(xd.cpu(), out)


def test_fused_ffn_relu_mode_vs_fp64():
    """act = 1: SuperGlue's MLP([512, 512, 256]) on cat([x, message]) with the BatchNorm folded (ReLU between the GEMMs)."""
    from imcui_hip import backend

    backend.set_precision(_dev(), 1)
    g = torch.Generator().manual_seed(77)
    M = 2048
    x = torch.randn(M, 256, generator=g)
    ctx = torch.randn(M, 256, generator=g)
    w1 = torch.randn(512, 512, generator=g) / 512 ** 0.5
    b1 = torch.randn(512, generator=g) * 0.3
    w2 = torch.randn(256, 512, generator=g) / 512 ** 0.5
    b2 = torch.randn(256, generator=g) * 0.1
    out = backend.FusedFFN(w1, b1, None, None, w2, b2, _dev())(x.to(_dev()), ctx.to(_dev())).cpu()
    ref = _ffn_reference(x, ctx, w1, b1, None, None, w2, b2)
    assert (out.double() - ref).abs().max().item() / ref.abs().max().item() < 3e-6


@pytest.mark.parametrize("act", [0, 1, 2, 3])
@pytest.mark.parametrize("M", [37, 2400, 4096])
def test_fused_ffn_token_tiles_are_bitwise_equal(act, M):
    """The fused FFN's 128 / 64 / 32-token workgroup tiles (option `ffn_tile`; picked by token count otherwise, so that a single pair
    still fills the CUs) run the same arithmetic per token: bitwise equal outputs, for every activation mode and a ragged last tile."""
    from imcui_hip import backend

    backend.set_precision(_dev(), 1)
    g = torch.Generator().manual_seed(act * 1000 + M)
    x = torch.randn(M, 256, generator=g)
    ctx = torch.randn(M, 256, generator=g) * 0.8
    w1 = torch.randn(512, 512, generator=g) / 512 ** 0.5
    w2 = torch.randn(256, 512, generator=g) / 512 ** 0.5
    b1, b2 = torch.randn(512, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1
    nf = 512 if act == 0 else 256
    gamma, beta = 1.0 + 0.3 * torch.randn(nf, generator=g), 0.2 * torch.randn(nf, generator=g)
    if act == 1:
        ffn = backend.FusedFFN(w1, b1, None, None, w2, b2, _dev())
    elif act == 0:
        ffn = backend.FusedFFN(w1, b1, gamma, beta, w2, b2, _dev())
    else:
        ffn = backend.FusedFFN(w1, torch.zeros(512), gamma, beta, w2, torch.zeros(256), _dev(), act=act)
    outs = {}
    for tile in (128, 64, 32, 0):
        with backend.option(_dev(), ffn_tile=tile):
            outs[tile] = ffn(x.to(_dev()), ctx.to(_dev())).cpu()
    assert torch.isfinite(outs[128]).all()
    for tile in (64, 32, 0):
        assert torch.equal(outs[tile], outs[128]), tile


@pytest.mark.parametrize("act,M", [(2, 4800), (3, 2400), (3, 16384)])
def test_fused_ffn_post_layernorm_modes_vs_fp64(act, M):
    """act 2 / 3: x + LayerNorm(fc2(act(fc1([x | message])))) -- the coarse MLPs of EfficientLoFTR (LeakyReLU) and LoFTR
    (ReLU).  M = 4800 / 2400 are not multiples of the 128-token tile: the last tile is partly masked."""
    from imcui_hip import backend

    backend.set_precision(_dev(), 1)
    g = torch.Generator().manual_seed(act * 100 + M)
    x = torch.randn(M, 256, generator=g)
    ctx = torch.randn(M, 256, generator=g) * 0.8
    w1 = torch.randn(512, 512, generator=g) / 512 ** 0.5
    w2 = torch.randn(256, 512, generator=g) / 512 ** 0.5
    gamma = 1.0 + 0.3 * torch.randn(256, generator=g)
    beta = 0.2 * torch.randn(256, generator=g)
    zero1, zero2 = torch.zeros(512), torch.zeros(256)
    out = backend.FusedFFN(w1, zero1, gamma, beta, w2, zero2, _dev(), act=act)(x.to(_dev()), ctx.to(_dev())).cpu()
    ref = _ffn_reference(x, ctx, w1, zero1, gamma, beta, w2, zero2, act=act)
    assert torch.isfinite(out).all() and (out.double() - ref).abs().max().item() / ref.abs().max().item() < 3e-6
