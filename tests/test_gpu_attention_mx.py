"""Attention variant 9, an OPT-IN (csrc/attention_mx.hip; GPU box only): K.Q^T in three f16 products, P.V as ONE f16 product plus two block-scaled
fp6 (e2m3, one e8m0 scale per 32 keys) correction products on v_mfma_scale_f32_32x32x64_f8f6f4 -- nothing dropped, 36 matrix instructions per
64-key tile instead of 48.  Checked at the kernel (float64 soft-max attention on ragged / poisoned / spiked inputs, independence of the
padding, closeness to the three-product kernel) and through LightGlue (the per-layer / match / score bars of tests/test_gpu_lightglue.py on
the three weight sets at N = M = 2048, with EVERY attention launch on variant 9).  The arithmetic was accepted on the CPU first:
tools/mx_corrections_probe.py, profiles/r06_lab_mx_corrections.txt."""
import pytest
import torch

from oracle.lightglue import LightGlueOracle
from parity_utils import assert_matches_equal_or_tied, synthetic_matching_problem
from test_gpu_lightglue import WEIGHTS, _batch, _check_layers, _model, _oracle_pair

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(q, k, v, cnt, cross):
    S = q.shape[0]
    out = []
    for s in range(S):
        ks = s ^ 1 if cross else s
        nq, nk = int(cnt[s]), int(cnt[ks])
        att = torch.softmax(q[s, :, :nq].double() @ k[ks, :, :nk].double().transpose(-1, -2), -1)
        out.append((att @ v[ks, :, :nk].double()).float().permute(1, 0, 2))  # [nq, heads, 64]
    return out


@pytest.mark.parametrize("cross", [False, True])
def test_variant9_vs_float64_ragged_poisoned_spiked(cross):
    from imcui_hip import backend

    backend.set_precision(torch.device(DEV), 1)
    g = torch.Generator().manual_seed(9)
    S, Hh, R = 6, 4, 2048
    cnt = torch.tensor([2048, 1999, 1025, 64, 1984, 1857], dtype=torch.int32)  # full, a key short of a tile, one key into a tile, one tile, odd / even tile counts
    q = torch.randn(S, Hh, R, 64, generator=g) * 0.6
    k = torch.randn(S, Hh, R, 64, generator=g)
    v = torch.randn(S, Hh, R, 64, generator=g) * (10.0 ** torch.randint(-3, 3, (S, Hh, 1, 64), generator=g).float())  # feature scales over six decades: the block scales matter
    k[:, :, 1000] *= 3.0  # late spike: the O / l rescale branch fires in the middle of the sequence
    for s in range(S):
        k[s, :, cnt[s]:] = float("nan")
        v[s, :, cnt[s]:] = float("inf")
    dev = lambda t: t.to(DEV)  # noqa: E731
    out = backend.attention_mx_f32(dev(q), dev(k), dev(v), dev(cnt), cross).cpu().view(S, R, Hh, 64)
    base = backend.attention_f32(dev(q), dev(k), dev(v), dev(cnt), cross, True).cpu().view(S, R, Hh, 64)  # the handle's default variant (three products)
    for s, ref in enumerate(_ref(q, k, v, cnt, cross)):
        nq = int(cnt[s])
        got, b3 = out[s, :nq], base[s, :nq]
        assert torch.isfinite(got).all(), f"sequence {s}"
        scale = ref.abs().amax((0, 1), keepdim=True).clamp_min(1e-30)  # per feature: the values span six decades
        e9, e3 = ((got - ref).abs() / scale).max().item(), ((b3 - ref).abs() / scale).max().item()
        print(f"[variant 9] cross={cross} sequence {s}: error / feature scale {e9:.2e} (three f16 products: {e3:.2e})")
        # measured 3.0e-5 .. 3.4e-5 on this deliberately peaked soft-max (no averaging) with V spread over six decades; the CPU emulation of the
        # scheme gives 3.3e-5 on the same tensors, dropping both corrections 9e-4, the three-product kernel 1.8e-6
        assert e9 < 6e-5, (s, e9)


def test_variant9_sharp_softmax_and_padding_independence():
    from imcui_hip import backend

    backend.set_precision(torch.device(DEV), 1)
    g = torch.Generator().manual_seed(6)
    S, Hh, R = 2, 4, 384
    cnt = torch.tensor([384, 300], dtype=torch.int32)
    q = torch.randn(S, Hh, R, 64, generator=g) * 3.0
    k = torch.randn(S, Hh, R, 64, generator=g) * 3.0
    k[:, :, 290] *= 4.0  # spike in the last tile
    v = torch.randn(S, Hh, R, 64, generator=g)
    out = backend.attention_mx_f32(q.to(DEV), k.to(DEV), v.to(DEV), cnt.to(DEV), False).cpu().view(S, R, Hh, 64)
    for s, ref in enumerate(_ref(q, k, v, cnt, False)):
        assert (out[s, : int(cnt[s])] - ref).abs().max().item() < 5e-4  # logits reach several hundred (tests/test_gpu_kernels.py: same bound)
    # what the rows past a sequence's count hold must not change a bit of the valid rows (the fp6 block scales included)
    g = torch.Generator().manual_seed(21)
    S, Hh, R = 2, 4, 512
    cnt = torch.tensor([300, 211], dtype=torch.int32)
    q = torch.randn(S, Hh, R, 64, generator=g) * 0.7
    k = torch.randn(S, Hh, R, 64, generator=g)
    v = torch.randn(S, Hh, R, 64, generator=g)
    k[:, :, 150] *= 2.5
    outs = []
    for fill in (0.0, 40.0, -4.0e4):
        qq, kk, vv = q.clone(), k.clone(), v.clone()
        for s in range(S):
            qq[s, :, cnt[s]:] = fill * torch.randn(R - int(cnt[s]), 64, generator=g) if fill else 0.0
            kk[s, :, cnt[s]:] = fill
            vv[s, :, cnt[s]:] = fill
        o = backend.attention_mx_f32(qq.to(DEV), kk.to(DEV), vv.to(DEV), cnt.to(DEV), False).cpu().view(S, R, Hh, 64)
        outs.append(torch.cat([o[s, : int(cnt[s])].reshape(-1) for s in range(S)]))
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("weights", ["damped", "strong", "random"])
def test_lightglue_on_variant9_vs_oracle(weights):
    """All 18 attention launches on variant 9, N = M = 2048, nine layers: the bars of test_lightglue_full_size_vs_oracle."""
    from imcui_hip import backend

    dev = torch.device(DEV)
    backend.set_precision(dev, 1)
    torch.set_num_threads(16)
    sd = WEIGHTS[weights]
    problems = [synthetic_matching_problem(40, 2048, 2048, 300), synthetic_matching_problem(41, 2048, 1900, 250)]
    k0, k1, d0, d1, n0, n1 = _batch(problems)
    model = _model(-1, -1, sd=sd)
    with backend.option(dev, attn_variant=9):
        out = model.forward_batched(k0.cuda(), k1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480), layer_dump=True)
        torch.cuda.synchronize()
    dump = out.pop("_layers")
    out = {k: v.cpu() for k, v in out.items()}
    ora = LightGlueOracle(sd, dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.1))
    for b, (a, c, e, f) in enumerate(problems):
        ref = _oracle_pair(ora, a, c, e, f)
        na = len(a)
        tag = f"variant 9, {weights} pair {b}"
        worst = _check_layers(dump, b, ref, tag)
        tol = 1e-4 * max(1.0, ref["_sim"].abs().max().item() / 100.0)
        ties = assert_matches_equal_or_tied(out["matches0"][b, :na], ref["_log_assignment"][0], ref["matches0"][0], 0.1, tol=tol, tag=tag)
        same = out["matches0"][b, :na].long() == ref["matches0"][0]
        d0s = (out["matching_scores0"][b, :na] - ref["matching_scores0"][0]).abs()
        print(f"[parity] {tag}: worst layer error {worst:.2e}, matches {(ref['matches0'] > -1).sum().item()}, ties {ties}, score error {d0s[same].max().item():.2e}")
        # Measured (round 6): damped 1.9e-6 / 9.1e-6, strong 1.5e-5 / 6.1e-5, random 1.3e-5 / 4.1e-4.  On the full-strength weight sets that is
        # ABOVE the bar the default arithmetic is held to (layer 1e-5; the audit rule of tools/attn_mix_audit.py: 1.2e-5 / 5e-5) and the kernel
        # is only 11 % faster than variant 8 (the non-matrix instructions of a key tile now outweigh its 36 matrix instructions): variant 9
        # stays an OPT-IN like variants 6 / 7.  The bounds below pin its class, they are not the parity bar.
        assert worst < {"damped": 1e-5, "strong": 3e-5, "random": 3e-5}[weights], (tag, worst)
        assert d0s[same].max().item() < {"damped": 5e-5, "strong": 1.5e-4, "random": 1e-3}[weights], (tag, d0s[same].max().item())
