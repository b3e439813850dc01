"""The batched pair-matching driver (imcui_hip/hloc/match_features.py) on the real plugins: B pairs per C-ABI call
must give exactly what the reference's one-pair-per-call loop gives (imcui/hloc/match_features.py:172-185 +
`writer_fn` :73-83), for LightGlue and SuperGlue, with images of two sizes in the pair list."""
import numpy as np
import pytest
import torch

from imcui_hip.hloc import match_features as mf
from imcui_hip.synth_weights import lightglue_state_dict, superglue_state_dict
from parity_utils import synthetic_matching_problem

pytestmark = pytest.mark.gpu


def _store():
    feats = {}
    sizes = [(640, 480), (640, 480), (640, 480), (320, 240), (320, 240)]
    counts = [300, 280, 257, 200, 190]
    k_all, d_all = [], []
    a, c, e, f = synthetic_matching_problem(5, 300, 300, 60)
    base_k, base_d = torch.cat([a, c]), torch.cat([e, f])
    g = torch.Generator().manual_seed(9)
    for i, ((w, h), n) in enumerate(zip(sizes, counts)):
        idx = torch.randperm(len(base_k), generator=g)[:n]
        kp = base_k[idx] * torch.tensor([w / 640.0, h / 480.0])
        feats[f"db/img{i}.jpg"] = {"keypoints": kp.numpy(), "scores": torch.rand(n, generator=g).numpy(),
                                   "descriptors": base_d[idx].t().contiguous().numpy(), "image_size": np.array([w, h])}  # fmt: skip
    return mf.DictFeatureStore(feats)


@pytest.mark.parametrize("name", ["lightglue", "superglue"])
def test_driver_equals_one_pair_per_call(name):
    if name == "lightglue":
        from imcui_hip.hloc.matchers.lightglue import LightGlue

        model = LightGlue({"depth_confidence": 0.95, "width_confidence": 0.99, "match_threshold": 0.1, "state_dict": lightglue_state_dict(0)})
    else:
        from imcui_hip.hloc.matchers.superglue import SuperGlue

        model = SuperGlue({"sinkhorn_iterations": 20, "match_threshold": 0.2, "state_dict": superglue_state_dict(0)})
    model = model.eval().to("cuda:0")
    store = _store()
    n = [f"db/img{i}.jpg" for i in range(5)]
    pairs = [(n[0], n[1]), (n[1], n[2]), (n[1], n[0]), (n[3], n[4]), (n[0], n[3]), (n[2], n[0])]
    pairs = mf.find_unique_new_pairs(pairs)
    assert len(pairs) == 5
    sink = mf.DictMatchSink()
    assert mf.match_from_pairs(model, pairs, store, store, sink, batch_size=4) == 5
    for q, r in pairs:
        fq, fr = store.get(q), store.get(r)
        data = {}
        for side, f in (("0", fq), ("1", fr)):
            for k in ("keypoints", "scores", "descriptors"):
                data[k + side] = torch.from_numpy(f[k])[None].float().cuda()
            data["image" + side] = torch.empty((1, 1) + tuple(int(v) for v in f["image_size"])[::-1])
        with torch.no_grad():
            pred = model(data)
        got = sink.matches[mf.names_to_pair(q, r)]
        assert np.array_equal(got["matches0"], pred["matches0"][0].cpu().short().numpy()), (q, r)
        assert np.array_equal(got["matching_scores0"], pred["matching_scores0"][0].cpu().half().numpy()), (q, r)
    assert (sink.matches["db-img0.jpg/db-img1.jpg"]["matches0"] > -1).sum() > 20
