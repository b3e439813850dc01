"""The persistent similarity-and-reduce kernel (csrc/simred.hip) through its C-ABI test hook, against float64 torch on the host
(GPU box only).  Every mode -- nearest neighbours, soft-max statistics, dual-softmax confidences, LightGlue's log assignment -- in both
arithmetics, for every supported width (64 / 128 / 256 = the three kernel geometries), at sizes that are not multiples of the 128 x 128
tile, with device-side row / column counts below the static sizes, with one and with several column chunks, and with tile flags.
The matchers built on it have their own parity tests (mutual-NN goldens, LoFTR / EfficientLoFTR / dual-softmax against the oracle);
this file pins the kernel itself: what a row or a column reduces to must be what the full matrix would have given."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
NN, LSE, DSBEST, LGBEST, NN1 = 0, 1, 2, 3, 4
BIG = 0x7FFFFFFF


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def run(mode, A, Bm, alpha=1.0, mcnt=None, ncnt=None, nchunk=0, stats=None, l01=None, flags=None):
    """One launch -> dict of the raw outputs (row slots [batch, nchunk, M], column slots [batch, nrb, N])."""
    from imcui_hip import backend

    hd = backend.get_handle(torch.device(DEV))
    lib = hd.lib
    B, M, K = A.shape
    N = Bm.shape[1]
    nch = nchunk or lib.imcui_hip_simred_chunks(B, M, N)
    nrb = (M + 127) // 128
    f = lambda *s: torch.full(s, float("nan"), device=DEV)
    i = lambda *s: torch.full(s, -7, dtype=torch.int32, device=DEV)
    out = dict(r0=f(B, nch, M), r1=f(B, nch, M), ri=i(B, nch, M), c0=f(B, nrb, N), c1=f(B, nrb, N), ci=i(B, nrb, N))
    ws = torch.empty(lib.imcui_hip_simred_debug_workspace_bytes(B, M, N, K), dtype=torch.uint8, device=DEV)
    st = stats or (None,) * 4
    ll = l01 or (None, None)
    with torch.cuda.device(DEV):
        rc = lib.imcui_hip_simred_debug(hd.h, mode, _p(A), _p(Bm), B, M, N, K, _p(mcnt), _p(ncnt), C.c_float(alpha), nch, _p(out["r0"]), _p(out["r1"]),
                                        _p(out["ri"]), _p(out["c0"]), _p(out["c1"]), _p(out["ci"]), _p(st[0]), _p(st[1]), _p(st[2]), _p(st[3]), _p(ll[0]), _p(ll[1]),
                                        _p(flags), _p(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream))  # fmt: skip
        hd.check(rc, "imcui_hip_simred_debug")
    torch.cuda.synchronize()
    return {k: v.cpu() for k, v in out.items()}, nch


def problem(seed, B, M, N, K, spread=4.0):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(B, M, K, generator=g) * (spread / K**0.5)
    b = torch.randn(B, N, K, generator=g)
    return a, b


def sim64(a, b, alpha):
    return alpha * (a.double() @ b.double().transpose(1, 2))


SHAPES = [(2, 300, 517), (1, 128, 128), (3, 1, 5), (1, 1000, 77), (2, 129, 257)]


def _merge_lse(m, s):
    """slots [B, P, n] of (max, sum exp) -> log-sum-exp [B, n]."""
    m64, s64 = m.double(), s.double()
    mx = m64.max(1).values
    return mx + torch.log((s64 * torch.exp(m64 - mx[:, None])).sum(1))


@pytest.mark.parametrize("K", [64, 128, 256])
@pytest.mark.parametrize("B,M,N", SHAPES)
def test_softmax_statistics(precision, K, B, M, N):
    a, b = problem(K + M + N, B, M, N, K)
    ref = sim64(a, b, 0.7)
    for nchunk in (0, 1, min(3, (N + 127) // 128)):
        out, nch = run(LSE, a.to(DEV), b.to(DEV), alpha=0.7, nchunk=nchunk)
        nct = (N + 127) // 128
        tpc = (nct + nch - 1) // nch
        live = [c for c in range(nch) if c * tpc < nct]  # (a chunk past the last tile writes the neutral (-inf, 0))
        rl = _merge_lse(out["r0"][:, live], out["r1"][:, live])
        cl = _merge_lse(out["c0"], out["c1"])
        assert (rl - torch.logsumexp(ref, 2)).abs().max() < 5e-5, (K, nchunk)
        assert (cl - torch.logsumexp(ref, 1)).abs().max() < 5e-5, (K, nchunk)
        # the maxima are maxima of the SAME similarities in either direction
        assert torch.equal(out["r0"][:, live].max(1).values.max(1).values, out["c0"].max(1).values.max(1).values)


@pytest.mark.parametrize("K", [64, 128, 256])
@pytest.mark.parametrize("B,M,N", SHAPES)
def test_nearest_neighbours(precision, K, B, M, N):
    a, b = problem(3 * K + M + N, B, M, N, K)
    ref = sim64(a, b, 1.0)
    out, nch = run(NN, a.to(DEV), b.to(DEV), nchunk=min(2, (N + 127) // 128))
    for name, val, idx, sec, r in (("rows", out["r0"], out["ri"], out["r1"], ref), ("columns", out["c0"], out["ci"], out["c1"], ref.transpose(1, 2))):
        # fold the slots: larger value, lowest index on ties -- what nn_merge_kernel does
        v, k = val.double(), idx.long()
        best = v.max(1).values
        isb = v == best[:, None]
        bi = torch.where(isb, k, torch.full_like(k, BIG)).min(1).values
        n = r.shape[2]
        top2 = r.topk(min(2, n), dim=2)
        assert (best - top2.values[..., 0]).abs().max() < 2e-5, name
        agree = bi == top2.indices[..., 0]
        if not agree.all():  # only where the two best similarities are closer than the arithmetic resolves
            gap = (top2.values[..., 0] - top2.values[..., 1])[~agree]
            assert gap.max() < 1e-5, (name, gap.max())
        if n > 1:
            # second best over the slots: the best of the other slots or the winning slot's own second
            allv = torch.cat([v, sec.double()], 1)
            s2 = allv.clone()
            # remove ONE occurrence of the best (the winning slot's best)
            first = (s2 == best[:, None]).double().argmax(1)
            s2.scatter_(1, first[:, None], -np.inf)
            assert (s2.max(1).values - top2.values[..., 1]).abs().max() < 2e-5, name


@pytest.mark.parametrize("K", [64, 128, 256])
@pytest.mark.parametrize("B,M,N", SHAPES)
def test_nearest_neighbours_without_second_best(precision, K, B, M, N):
    """SR_NN1 (find_nn without a ratio test) = SR_NN's best values and first indices, bit for bit; the second-best slots stay untouched."""
    a, b = problem(5 * K + M + N, B, M, N, K)
    a[:, ::7] = a[:, :1].clone()  # ties between rows (and so between columns' candidates): the FIRST index must win in both modes
    nchunk = min(2, (N + 127) // 128)
    full, _ = run(NN, a.to(DEV), b.to(DEV), nchunk=nchunk)
    one, _ = run(NN1, a.to(DEV), b.to(DEV), nchunk=nchunk)
    nct = (N + 127) // 128
    tpc = (nct + nchunk - 1) // nchunk
    live = [c for c in range(nchunk) if c * tpc < nct]
    for key in ("r0", "ri"):
        assert torch.equal(full[key][:, live], one[key][:, live]), key
    for key in ("c0", "ci"):
        assert torch.equal(full[key], one[key]), key
    assert torch.isnan(one["r1"]).all() and torch.isnan(one["c1"]).all()


def _dual_softmax_conf(ref):
    return torch.softmax(ref, 1) * torch.softmax(ref, 2)


@pytest.mark.parametrize("K", [64, 128, 256])
@pytest.mark.parametrize("B,M,N", SHAPES)
def test_dual_softmax_confidence(precision, K, B, M, N):
    a, b = problem(5 * K + M + N, B, M, N, K, spread=12.0)
    ref = sim64(a, b, 1.3)
    conf = _dual_softmax_conf(ref)
    rmax, cmax = ref.max(2).values, ref.max(1).values
    rsum, csum = torch.exp(ref - rmax[..., None]).sum(2), torch.exp(ref - cmax[:, None]).sum(1)
    stats = [t.float().contiguous().to(DEV) for t in (rmax, rsum, cmax, csum)]
    for nchunk in (1, min(2, (N + 127) // 128)):
        out, nch = run(DSBEST, a.to(DEV), b.to(DEV), alpha=1.3, nchunk=nchunk, stats=stats)
        rv = out["r0"].double().max(1).values
        cv = out["c0"].double().max(1).values
        # |sim| reaches ~60 here: fp32 round-off of a similarity (1e-6 relative) is 6e-5 in each of the two exponents of a confidence near 1
        assert (rv - conf.max(2).values).abs().max() < 2.5e-4
        assert (cv - conf.max(1).values).abs().max() < 2.5e-4
        # first column attaining the row maximum: the chunk that holds the maximum reports it
        v, k = out["r0"].double(), out["ri"].long()
        bj = torch.where(v == rv[:, None], k, torch.full_like(k, BIG)).min(1).values
        want = conf.argmax(2)
        bad = bj != want
        if bad.any():
            top2 = conf.topk(min(2, N), 2).values
            assert ((top2[..., 0] - top2[..., -1])[bad] < 2.5e-4).all()
        # mutual maxima compare IDENTICAL numbers: where the double-precision matrix has a clear mutual maximum the device's row best equals its column best
        mut = (conf == conf.max(2, keepdim=True).values) & (conf == conf.max(1, keepdim=True).values) & (conf > 0.05)
        bb, ii, jj = torch.nonzero(mut, as_tuple=True)
        ok = bj[bb, ii] == jj
        assert (out["r0"].max(1).values[bb[ok], ii[ok]] == out["c0"].max(1).values[bb[ok], jj[ok]]).all()


def test_dual_softmax_tile_flags():
    """Only flagged tiles are evaluated: rows / columns of unflagged tiles keep their neutral results, flagged ones equal the full run."""
    from imcui_hip import backend

    backend.set_precision(torch.device(DEV), 1)
    B, M, N, K = 2, 700, 900, 256
    a, b = problem(11, B, M, N, K, spread=12.0)
    ref = sim64(a, b, 1.0)
    rmax, cmax = ref.max(2).values, ref.max(1).values
    rsum, csum = torch.exp(ref - rmax[..., None]).sum(2), torch.exp(ref - cmax[:, None]).sum(1)
    stats = [t.float().contiguous().to(DEV) for t in (rmax, rsum, cmax, csum)]
    full, _ = run(DSBEST, a.to(DEV), b.to(DEV), nchunk=1, stats=stats)
    nrb, nct = (M + 127) // 128, (N + 127) // 128
    g = torch.Generator().manual_seed(3)
    flags = (torch.rand(B, nrb, nct, generator=g) < 0.4).to(torch.uint8)
    flags[0, 1] = 0  # a row block without any tile
    part, _ = run(DSBEST, a.to(DEV), b.to(DEV), nchunk=1, stats=stats, flags=flags.to(DEV))
    conf = _dual_softmax_conf(ref).float()
    for bb in range(B):
        for rb in range(nrb):
            rows = slice(rb * 128, min(M, rb * 128 + 128))
            cols = torch.cat([torch.arange(ct * 128, min(N, ct * 128 + 128)) for ct in range(nct) if flags[bb, rb, ct]] or [torch.zeros(0, dtype=torch.long)])
            if len(cols) == 0:
                assert (part["r0"][bb, 0, rows] == -1).all() and (part["ri"][bb, 0, rows] == BIG).all()
                continue
            sub = conf[bb, rows][:, cols]
            assert (part["r0"][bb, 0, rows] - sub.max(1).values).abs().max() < 2.5e-4
            for ct in range(nct):
                cc = slice(ct * 128, min(N, ct * 128 + 128))
                if flags[bb, rb, ct]:
                    assert torch.equal(part["c0"][bb, rb, cc], full["c0"][bb, rb, cc])  # the same tile, the same instructions: bitwise


@pytest.mark.parametrize("K", [64, 128, 256])
@pytest.mark.parametrize("B,M,N", SHAPES)
def test_lightglue_log_assignment(precision, K, B, M, N):
    a, b = problem(7 * K + M + N, B, M, N, K, spread=10.0)
    ref = sim64(a, b, 1.0)
    g = torch.Generator().manual_seed(M)
    l0, l1 = -torch.rand(B, M, generator=g) * 3, -torch.rand(B, N, generator=g) * 3
    score = torch.log_softmax(ref, 2) + torch.log_softmax(ref, 1) + l0.double()[:, :, None] + l1.double()[:, None, :]
    rmax, cmax = ref.max(2).values, ref.max(1).values
    rls, cls = torch.log(torch.exp(ref - rmax[..., None]).sum(2)), torch.log(torch.exp(ref - cmax[:, None]).sum(1))
    stats = [t.float().contiguous().to(DEV) for t in (rmax, rls, cmax, cls)]
    out, nch = run(LGBEST, a.to(DEV), b.to(DEV), nchunk=min(2, (N + 127) // 128), stats=stats, l01=(l0.to(DEV), l1.to(DEV)))
    for name, val, idx, r in (("rows", out["r0"], out["ri"], score), ("columns", out["c0"], out["ci"], score.transpose(1, 2))):
        v, k = val.double(), idx.long()
        best = v.max(1).values
        bi = torch.where(v == best[:, None], k, torch.full_like(k, BIG)).min(1).values
        assert (best - r.max(2).values).abs().max() < 2e-4, name  # (a sum of two log-soft-maxes of similarities up to ~50)
        bad = bi != r.argmax(2)
        if bad.any():
            top2 = r.topk(min(2, r.shape[2]), 2).values
            assert ((top2[..., 0] - top2[..., -1])[bad] < 2e-4).all(), name


@pytest.mark.parametrize("mode", [NN, LSE])
def test_device_side_counts(precision, mode):
    """Rows / columns beyond the per-batch counts hold NaN: nothing of them may reach a result (they are zeroed while the operands are packed and
    masked in the epilogue); results equal the run on the cropped matrices bit for bit."""
    B, M, N, K = 3, 400, 390, 128
    a, b = problem(21, B, M, N, K)
    mc, nc = torch.tensor([400, 129, 7], dtype=torch.int32), torch.tensor([1, 390, 200], dtype=torch.int32)
    for z in range(B):
        a[z, mc[z] :] = float("nan")
        b[z, nc[z] :] = float("nan")
    out, nch = run(mode, a.to(DEV), b.to(DEV), mcnt=mc.to(DEV), ncnt=nc.to(DEV), nchunk=1)
    for z in range(B):
        m, n = int(mc[z]), int(nc[z])
        one, _ = run(mode, a[z : z + 1, :m].contiguous().to(DEV), b[z : z + 1, :n].contiguous().to(DEV), nchunk=1)
        nrb = (m + 127) // 128
        for key in ("r0", "r1") + (("ri",) if mode == NN else ()):
            assert torch.equal(out[key][z, 0, :m], one[key][0, 0]), (z, key)
        for key in ("c0", "c1") + (("ci",) if mode == NN else ()):
            assert torch.equal(out[key][z, :nrb, :n], one[key][0]), (z, key)


def test_repeatable_and_equal_to_the_tile_gemm_path():
    """Two runs give the same bits; and the mutual-NN matcher on the persistent kernel returns exactly what the round-4 tile GEMM with the reducing
    epilogue returns (same products in the same order per similarity): matches AND scores bitwise, 5000 x 5000 x 128 and 2048 x 2048 x 256."""
    from imcui_hip import backend

    dev = torch.device(DEV)
    backend.set_precision(dev, 1)
    for n, m, d in ((5000, 5000, 128), (2048, 1900, 256), (300, 5000, 64)):
        g = torch.Generator().manual_seed(n + d)
        d0 = torch.nn.functional.normalize(torch.randn(2, n, d, generator=g), dim=2).to(dev)
        d1 = torch.nn.functional.normalize(torch.randn(2, m, d, generator=g), dim=2).to(dev)
        with backend.option(dev, simred=1):
            m_new, s_new = backend.mutual_nn(d0, d1, 0.999, None, True)
            m_again, s_again = backend.mutual_nn(d0, d1, 0.999, None, True)
            m_dn, s_dn = backend.mutual_nn_dn(d0.permute(0, 2, 1).contiguous(), d1.permute(0, 2, 1).contiguous(), 0.999, None, True)
        with backend.option(dev, simred=0):
            m_old, s_old = backend.mutual_nn(d0, d1, 0.999, None, True)
        assert torch.equal(m_new, m_again) and torch.equal(s_new, s_again)
        assert torch.equal(m_new, m_dn) and torch.equal(s_new, s_dn)
        assert torch.equal(m_new, m_old), (n, m, d, (m_new != m_old).sum().item())
        assert torch.equal(s_new, s_old), (n, m, d, (s_new - s_old).abs().max().item())
        assert (m_new > -1).sum() > 20, (m_new > -1).sum()  # (mutual neighbours that pass the 0.999 ratio test exist: the comparison is not vacuous)


def test_nan_descriptors_do_not_index_out_of_range():
    """a descriptor of NaNs (an unnormalisable zero vector) leaves its row without a maximum; the matcher reports it unmatched instead
    of the sentinel index."""
    from imcui_hip import backend

    dev = torch.device(DEV)
    for prec in (1, 0):
        backend.set_precision(dev, prec)
        g = torch.Generator().manual_seed(5)
        d0 = torch.nn.functional.normalize(torch.randn(1, 300, 128, generator=g), dim=2)
        d1 = torch.nn.functional.normalize(torch.randn(1, 280, 128, generator=g), dim=2)
        d0[0, 17] = float("nan")
        for simred in (1, 0):
            with backend.option(dev, simred=simred):
                m, s = backend.mutual_nn(d0.to(dev), d1.to(dev), None, None, False)
            m = m.cpu()
            assert m[0, 17] == -1 and (m >= -1).all() and (m < 280).all(), (prec, simred, m[0, 17])
    backend.set_precision(dev, 1)
