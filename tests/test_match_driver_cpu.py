"""Host logic of the batched pair-matching driver (imcui_hip/hloc/match_features.py) on CPU with a stand-in matcher:
pair-list parsing / de-duplication as in the reference (imcui/hloc/utils/parsers.py:43-63,
imcui/hloc/match_features.py:118-138), grouping by image size, fixed-stride collation, per-pair slicing and the
on-disk dtypes of `writer_fn` (:73-83)."""
import numpy as np
import pytest
import torch

from imcui_hip.hloc import match_features as mf


class _StubMatcher(torch.nn.Module):
    """Mutual nearest neighbour on the valid rows of every pair; the key-point normalisation by the image size is
    mimicked by offsetting the scores with W0 + W1 so that a wrong size grouping shows up in the output."""

    def __init__(self, with_scores=False):
        super().__init__()
        self.calls = []
        self.with_scores = with_scores
        if with_scores:
            self.forward_batched = self._fb_scores
        else:
            self.forward_batched = self._fb

    def _run(self, d0, d1, n0, n1, size0, size1, s0=None):
        B, ncap = d0.shape[0], d0.shape[1]
        self.calls.append((B, ncap, tuple(size0), tuple(size1)))
        m0 = torch.full((B, ncap), -1, dtype=torch.int32)
        sc = torch.zeros(B, ncap)
        for b in range(B):
            a, c = int(n0[b]), int(n1[b])
            if a == 0 or c == 0:
                continue
            sim = d0[b, :a] @ d1[b, :c].t()
            j = sim.argmax(1)
            i = sim.argmax(0)
            mutual = i[j] == torch.arange(a)
            m0[b, :a] = torch.where(mutual, j.int(), torch.tensor(-1, dtype=torch.int32))
            sc[b, :a] = torch.where(mutual, sim.max(1).values, torch.zeros(())) + (s0[b, :a] if s0 is not None else 0) * 0
            sc[b, :a] += (size0[0] + size1[0]) * 1e-3
        return {"matches0": m0, "matching_scores0": sc}

    def _fb(self, k0, k1, d0, d1, n0, n1, size0, size1):
        return self._run(d0, d1, n0, n1, size0, size1)

    def _fb_scores(self, k0, k1, scores0, scores1, d0, d1, n0, n1, size0, size1):
        assert scores0.shape == k0.shape[:2]
        return self._run(d0, d1, n0, n1, size0, size1, scores0)


def _features(seed, names, sizes):
    g = np.random.default_rng(seed)
    feats = {}
    for name, (n, wh) in zip(names, sizes):
        feats[name] = {
            "keypoints": g.random((n, 2)).astype(np.float16) * 100,  # the reference stores halves (as_half)
            "scores": g.random(n).astype(np.float16),
            "descriptors": g.standard_normal((64, n)).astype(np.float16),
            "image_size": np.array(wh),
        }
    return feats


def test_pair_list_parsing_and_deduplication(tmp_path):
    p = tmp_path / "pairs.txt"
    p.write_text("a/1.jpg b/2.jpg\na/1.jpg c.jpg\n\nb/2.jpg a/1.jpg\nc.jpg d.jpg\n")
    ret = mf.parse_retrieval(p)
    assert ret == {"a/1.jpg": ["b/2.jpg", "c.jpg"], "b/2.jpg": ["a/1.jpg"], "c.jpg": ["d.jpg"]}
    pairs = [(q, r) for q, rs in ret.items() for r in rs]
    uniq = mf.find_unique_new_pairs(pairs)
    assert uniq == [("a/1.jpg", "b/2.jpg"), ("a/1.jpg", "c.jpg"), ("c.jpg", "d.jpg")]  # the reversed duplicate is dropped
    assert mf.names_to_pair("a/1.jpg", "b/2.jpg") == "a-1.jpg/b-2.jpg" and mf.names_to_pair_old("a/1.jpg", "c.jpg") == "a-1.jpg_c.jpg"
    done = {"c.jpg/a-1.jpg", "c.jpg_d.jpg"}  # one stored in the other order, one under the old key style
    assert mf.find_unique_new_pairs(pairs, done) == [("a/1.jpg", "b/2.jpg")]


@pytest.mark.parametrize("with_scores", [False, True])
def test_batched_driver_equals_per_pair_calls(with_scores):
    names = [f"img{i}" for i in range(7)]
    sizes = [(50, (640, 480)), (37, (640, 480)), (64, (640, 480)), (0, (640, 480)), (41, (320, 240)), (29, (320, 240)), (33, (640, 480))]
    feats = _features(0, names, sizes)
    store = mf.DictFeatureStore(feats)
    pairs = [("img0", "img1"), ("img1", "img2"), ("img0", "img4"), ("img4", "img5"), ("img2", "img3"), ("img6", "img0"), ("img5", "img6")]
    model = _StubMatcher(with_scores)
    sink = mf.DictMatchSink()
    assert mf.match_from_pairs(model, pairs, store, store, sink, batch_size=2, device=torch.device("cpu")) == len(pairs)
    # grouping: (640x480, 640x480) x4 -> 2 calls of 2; (640,320) x1; (320,320) x1; (320,640) x1
    assert sorted(c[0] for c in model.calls) == [1, 1, 1, 2, 2]
    assert all(c[2] != c[3] or c[2] in ((640, 480), (320, 240)) for c in model.calls)
    ref_model = _StubMatcher(with_scores)
    for q, r in pairs:
        one = mf.DictMatchSink()
        mf.match_from_pairs(ref_model, [(q, r)], store, store, one, batch_size=1, device=torch.device("cpu"))
        key = mf.names_to_pair(q, r)
        got, exp = sink.matches[key], one.matches[key]
        assert got["matches0"].dtype == np.int16 and got["matching_scores0"].dtype == np.float16
        assert got["matches0"].shape == (feats[q]["keypoints"].shape[0],)
        assert np.array_equal(got["matches0"], exp["matches0"]) and np.array_equal(got["matching_scores0"], exp["matching_scores0"])
    assert (sink.matches["img2/img3"]["matches0"] == -1).all()  # empty second image
    assert (sink.matches["img0/img1"]["matches0"] > -1).sum() > 5
