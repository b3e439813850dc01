"""SuperGlue HIP path vs the CPU oracle on identical seeded inputs (GPU box only).

Bar: matches0 / matches1 bit-exact, matching scores within 1e-4.  Ragged batches: several pairs of different
sizes go through ONE HIP call and are compared with per-pair oracle runs.
"""
import pytest
import torch

from oracle.superglue import SuperGlueOracle
from imcui_hip.synth_weights import superglue_state_dict
from parity_utils import synthetic_matching_problem

pytestmark = pytest.mark.gpu

SSD = superglue_state_dict(0)
SIZES = [(700, 650, 150), (512, 512, 100), (130, 257, 30), (1024, 900, 300)]


def _model(iters, th=0.2):
    from imcui_hip.hloc.matchers.superglue import SuperGlue

    return SuperGlue({"sinkhorn_iterations": iters, "match_threshold": th, "state_dict": SSD}).eval().to("cuda:0")


def _scores(seed, n):
    return torch.rand(n, generator=torch.Generator().manual_seed(seed))


def _batch(problems):
    B = len(problems)
    ncap = max(max(p[0].shape[0], p[1].shape[0]) for p in problems)
    k0, k1 = torch.zeros(B, ncap, 2), torch.zeros(B, ncap, 2)
    d0, d1 = torch.zeros(B, ncap, 256), torch.zeros(B, ncap, 256)
    s0, s1 = torch.zeros(B, ncap), torch.zeros(B, ncap)
    n0, n1 = torch.zeros(B, dtype=torch.int32), torch.zeros(B, dtype=torch.int32)
    for b, (a, c, e, f, sa, sc) in enumerate(problems):
        k0[b, : len(a)], k1[b, : len(c)], d0[b, : len(a)], d1[b, : len(c)] = a, c, e, f
        s0[b, : len(a)], s1[b, : len(c)] = sa, sc
        n0[b], n1[b] = len(a), len(c)
    return k0, k1, s0, s1, d0, d1, n0, n1


def _problem(seed, n, m, o):
    a, c, e, f = synthetic_matching_problem(seed, n, m, o)
    return a, c, e, f, _scores(seed + 1000, n), _scores(seed + 2000, m)


def _oracle(ora, p):
    a, c, e, f, sa, sc = p
    img = torch.zeros(1, 1, 480, 640)
    return ora({"image0": img, "image1": img, "keypoints0": a[None], "keypoints1": c[None], "scores0": sa[None], "scores1": sc[None],
                "descriptors0": e.t()[None], "descriptors1": f.t()[None]})  # fmt: skip


@pytest.mark.parametrize("iters", [5, 50])
def test_superglue_ragged_batch_vs_oracle(iters, precision):
    torch.set_num_threads(8)
    problems = [_problem(40 + i, n, m, o) for i, (n, m, o) in enumerate(SIZES)]
    k0, k1, s0, s1, d0, d1, n0, n1 = _batch(problems)
    model = _model(iters)
    out = model.forward_batched(k0.cuda(), k1.cuda(), s0.cuda(), s1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480))
    torch.cuda.synchronize()
    out = {k: v.cpu() for k, v in out.items()}
    ora = SuperGlueOracle(SSD, {"sinkhorn_iterations": iters, "match_threshold": 0.2})
    for b, p in enumerate(problems):
        ref = _oracle(ora, p)
        na, nc = len(p[0]), len(p[1])
        tag = f"pair {b} (n={na},{nc}) iters={iters}"
        assert (ref["matches0"] > -1).sum() > 10, tag
        assert (out["matching_scores0"][b, :na] - ref["matching_scores0"][0]).abs().max().item() < 1e-4, tag
        assert (out["matching_scores1"][b, :nc] - ref["matching_scores1"][0]).abs().max().item() < 1e-4, tag
        assert torch.equal(out["matches0"][b, :na].long(), ref["matches0"][0]), tag
        assert torch.equal(out["matches1"][b, :nc].long(), ref["matches1"][0]), tag
        assert (out["matches0"][b, na:] == -1).all() and (out["matching_scores0"][b, na:] == 0).all()
        assert (out["matches1"][b, nc:] == -1).all() and (out["matching_scores1"][b, nc:] == 0).all()


def test_superglue_plugin_contract_and_empty():
    """Flat hloc dict in (descriptors [B,256,N]) -> reference keys out; an empty side -> all -1 (int32 upstream)."""
    p = _problem(3, 300, 280, 60)
    a, c, e, f, sa, sc = p
    model = _model(20)
    img = torch.zeros(1, 1, 480, 640)
    data = {"image0": img, "image1": img, "keypoints0": a[None].cuda(), "keypoints1": c[None].cuda(),
            "scores0": sa[None].cuda(), "scores1": sc[None].cuda(),
            "descriptors0": e.t()[None].contiguous().cuda(), "descriptors1": f.t()[None].contiguous().cuda()}  # fmt: skip
    with torch.no_grad():
        pred = model(data)
    assert set(pred) == {"matches0", "matches1", "matching_scores0", "matching_scores1"}
    assert pred["matches0"].dtype == torch.int64 and pred["matches0"].shape == (1, 300) and pred["matches1"].shape == (1, 280)
    ref = _oracle(SuperGlueOracle(SSD, {"sinkhorn_iterations": 20, "match_threshold": 0.2}), p)
    assert torch.equal(pred["matches0"].cpu(), ref["matches0"])
    assert torch.equal(pred["matches1"].cpu(), ref["matches1"])
    assert (pred["matching_scores0"].cpu() - ref["matching_scores0"]).abs().max().item() < 1e-4
    # the UI mutates the threshold on the live model (imcui/ui/utils.py:921-922): like the reference's `SG(conf)` copy, nothing changes ...
    model.conf["match_threshold"] = 0.9
    with torch.no_grad():
        same = model(data)
    assert torch.equal(same["matches0"], pred["matches0"]) and torch.equal(same["matching_scores0"], pred["matching_scores0"])
    # ... unless the plugin's opt-in says the conf is to be re-read on every call
    model.conf["runtime_match_threshold"] = True
    with torch.no_grad():
        strict = model(data)
    ref9 = _oracle(SuperGlueOracle(SSD, {"sinkhorn_iterations": 20, "match_threshold": 0.9}), p)
    assert torch.equal(strict["matches0"].cpu(), ref9["matches0"]) and (strict["matches0"] > -1).sum() < (pred["matches0"] > -1).sum()
    # empty second image
    data["keypoints1"] = torch.zeros(1, 0, 2).cuda()
    data["descriptors1"] = torch.zeros(1, 256, 0).cuda()
    data["scores1"] = torch.zeros(1, 0).cuda()
    with torch.no_grad():
        pred = model(data)
    assert (pred["matches0"] == -1).all() and pred["matches0"].dtype == torch.int32 and pred["matches1"].shape == (1, 0)
    assert (pred["matching_scores0"] == 0).all()


def test_superglue_batch_with_an_empty_pair_and_one_iteration():
    """A pair whose second image has no key-points inside a batch returns all -1 / 0 (per-pair early return) and does
    not disturb its neighbours; a single Sinkhorn round (the potentials start at u = v = 0)."""
    torch.set_num_threads(8)
    problems = [_problem(70, 260, 300, 50), _problem(71, 200, 128, 20)]
    k0, k1, s0, s1, d0, d1, n0, n1 = _batch(problems)
    n1e = n1.clone()
    n1e[1] = 0
    model = _model(1, th=0.0)
    out = model.forward_batched(k0.cuda(), k1.cuda(), s0.cuda(), s1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1e.cuda(), (640, 480), (640, 480))
    torch.cuda.synchronize()
    out = {k: v.cpu() for k, v in out.items()}
    ref = _oracle(SuperGlueOracle(SSD, {"sinkhorn_iterations": 1, "match_threshold": 0.0}), problems[0])
    na, nc = len(problems[0][0]), len(problems[0][1])
    assert torch.equal(out["matches0"][0, :na].long(), ref["matches0"][0])
    assert torch.equal(out["matches1"][0, :nc].long(), ref["matches1"][0])
    assert (out["matching_scores0"][0, :na] - ref["matching_scores0"][0]).abs().max().item() < 1e-4
    assert (out["matches0"][1] == -1).all() and (out["matches1"][1] == -1).all()
    assert (out["matching_scores0"][1] == 0).all() and (out["matching_scores1"][1] == 0).all()


def test_superglue_full_size_replicas_are_identical():
    """2048 key-points per image (the size of BASELINE configs[2]), 8 pairs = 2 distinct problems replicated 4x:
    every replica must give the same matches and scores bit for bit (slot / CU / timing independence), and the
    optimal transport must be a valid partial assignment (a column is used by at most one row)."""
    problems = [_problem(90 + (i % 2), 2048, 2048, 500) for i in range(8)]
    k0, k1, s0, s1, d0, d1, n0, n1 = _batch(problems)
    model = _model(20)
    out = model.forward_batched(k0.cuda(), k1.cuda(), s0.cuda(), s1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480))
    torch.cuda.synchronize()
    m0, m1, ms0 = out["matches0"].cpu(), out["matches1"].cpu(), out["matching_scores0"].cpu()
    for b in range(2, 8):
        assert torch.equal(m0[b], m0[b % 2]) and torch.equal(m1[b], m1[b % 2]) and torch.equal(ms0[b], ms0[b % 2]), b
    for b in range(2):
        v = m0[b][m0[b] > -1]
        assert len(v) > 1000 and len(torch.unique(v)) == len(v)
        idx = torch.where(m0[b] > -1)[0]
        assert torch.equal(m1[b][v.long()].long(), idx)  # mutual consistency
        assert (ms0[b][idx] > 0.2).all() and (ms0[b] <= 1.0 + 1e-5).all()


def test_superglue_more_than_2048_keypoints_uses_the_streaming_rounds():
    """Above 2048 key-points per image the Sinkhorn rounds run as separate row / column passes (the fused
    register-resident round covers R <= 2048): same parity bar."""
    torch.set_num_threads(16)
    p = _problem(77, 2100, 2060, 400)
    k0, k1, s0, s1, d0, d1, n0, n1 = _batch([p])
    model = _model(5)
    out = model.forward_batched(k0.cuda(), k1.cuda(), s0.cuda(), s1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480))
    torch.cuda.synchronize()
    out = {k: v.cpu() for k, v in out.items()}
    ref = _oracle(SuperGlueOracle(SSD, {"sinkhorn_iterations": 5, "match_threshold": 0.2}), p)
    assert (ref["matches0"] > -1).sum() > 1000
    assert (out["matching_scores0"][0, :2100] - ref["matching_scores0"][0]).abs().max().item() < 1e-4
    assert torch.equal(out["matches0"][0, :2100].long(), ref["matches0"][0])
    assert torch.equal(out["matches1"][0, :2060].long(), ref["matches1"][0])
