"""Host-side PairViewer restatement (imcui_hip/hloc/matchers/pair_viewer.py; the step imcui/hloc/matchers/duster.py:74-79 takes from
upstream's `global_aligner`) on synthetic two-camera scenes with a KNOWN focal length and relative pose.  PARITY UNPINNED: upstream's
dust3r.cloud_opt and cv2 are not installed, so these tests check the geometry against ground truth, not against the original code."""
import numpy as np
import torch

from imcui_hip.hloc.matchers.pair_viewer import PairViewerScene, estimate_focal_knowing_depth, solve_pnp_ransac


def _rot(ax, ang):
    ax = np.asarray(ax, dtype=np.float64) / np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K


def _scene(H=48, W=64, f=70.0, seed=0, conf_hi=(8.0, 6.0), conf2_scale=(1.0, 1.3)):
    """Two pinhole cameras (focal f, principal point (W/2, H/2)) looking at smooth random depth surfaces.  Returns the inference
    dictionary a perfect DUSt3R would produce for the symmetrised pair -- edges (1, 0), (0, 1): view 1's points in its own frame,
    view 2's points in view 1's frame -- and the ground truth (pose of camera 1 in camera 0's frame)."""
    g = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(W), np.arange(H))

    def surface():
        z = 4.0 + 0.6 * np.sin(u / 9.0 + g.uniform(0, 6)) + 0.5 * np.cos(v / 7.0 + g.uniform(0, 6)) + 0.3 * np.sin((u + v) / 5.0)
        return np.stack(((u - W / 2) * z / f, (v - H / 2) * z / f, z), -1)

    P = [surface(), surface()]  # points seen by camera k, in camera k's own frame
    R01, t01 = _rot((0.2, 1.0, 0.1), 0.25), np.array([0.6, -0.1, 0.2])  # camera 1 -> camera 0:  X0 = R01 X1 + t01
    to0 = lambda X1: X1 @ R01.T + t01  # noqa: E731
    to1 = lambda X0: (X0 - t01) @ R01  # noqa: E731
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
    conf = [np.full((H, W), conf_hi[0]), np.full((H, W), conf_hi[1])]
    conf[0][:6], conf[1][:, :5] = 1.5, 1.2  # low-confidence bands: masked out (threshold 3)
    imgs = [torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(seed + k)) * 2 - 1 for k in range(2)]
    out = {
        "view1": {"img": torch.cat((imgs[1], imgs[0])), "idx": [1, 0], "true_shape": torch.tensor([[H, W], [H, W]])},
        "view2": {"img": torch.cat((imgs[0], imgs[1])), "idx": [0, 1], "true_shape": torch.tensor([[H, W], [H, W]])},
        "pred1": {"pts3d": torch.stack((T(P[1]), T(P[0]))), "conf": torch.stack((T(conf[1]), T(conf[0])))},
        # (view-2 confidences of the edges (1, 0) and (0, 1), scaled so that the two edges differ: upstream anchors the frame on the edge
        # with the larger mean(conf_i) * mean(conf_j))
        "pred2": {"pts3d_in_other_view": torch.stack((T(to1(P[0])), T(to0(P[1])))), "conf": torch.stack((T(conf[0] * conf2_scale[0]), T(conf[1] * conf2_scale[1])))},
    }
    return out, (R01, t01, P, f, conf)


def test_focal_and_pnp_on_exact_data():
    out, (R01, t01, P, f, _) = _scene()
    fe = estimate_focal_knowing_depth(out["pred1"]["pts3d"][1], torch.tensor((32.0, 24.0)))
    assert abs(fe - f) / f < 1e-4
    H, W = 48, 64
    pixels = np.mgrid[:W, :H].T.astype(np.float32).reshape(-1, 2)
    X = out["pred2"]["pts3d_in_other_view"][1].numpy().reshape(-1, 3)  # image 1's points in camera 0's frame
    g = np.random.default_rng(1)
    bad = g.choice(len(X), len(X) // 5, replace=False)  # 20 % gross outliers
    Xo = X.copy()
    Xo[bad] += g.normal(0, 1.0, (len(bad), 3))
    R, t, inl = solve_pnp_ransac(Xo, pixels, f, (W / 2, H / 2))
    # world (camera 0) -> camera 1 is the inverse of (R01, t01)
    assert np.abs(R - R01.T).max() < 2e-3 and np.abs(t + R01.T @ t01).max() < 5e-3
    good = np.ones(len(X), bool)
    good[bad] = False
    assert inl[good].mean() > 0.99 and inl[bad].mean() < 0.2
    assert solve_pnp_ransac(X[:5], pixels[:5], f, (W / 2, H / 2)) is None  # fewer than six points: the caller keeps the identity


def test_pair_viewer_scene_recovers_the_geometry():
    for scale, anchor in (((1.0, 1.3), 0), ((1.3, 1.0), 1), ((1.0, 1.0), 1)):  # (a tie takes the `else` branch, as upstream's `if confs[0] > confs[1]`)
        out, (R01, t01, P, f, conf) = _scene(conf2_scale=scale)
        sc = PairViewerScene(out)
        H, W = 48, 64
        assert [im.shape for im in sc.imgs] == [(H, W, 3), (H, W, 3)]
        masks = [m.numpy() for m in sc.get_masks()]
        assert np.array_equal(masks[0], conf[0] * max(1.0, scale[0]) > 3) and np.array_equal(masks[1], conf[1] * max(1.0, scale[1]) > 3)
        assert all(abs(fk - f) / f < 1e-3 for fk in sc.get_focals())
        assert (sc.confs[0] > sc.confs[1]) == (anchor == 0)
        pts = [p.numpy() for p in sc.get_pts3d()]
        poses = [p.numpy() for p in sc.get_im_poses()]
        assert np.allclose(poses[anchor], np.eye(4))
        if anchor == 0:  # cloud expressed in camera 0
            want = [P[0], P[1] @ R01.T + t01]
            assert np.abs(poses[1][:3, :3] - R01).max() < 2e-3 and np.abs(poses[1][:3, 3] - t01).max() < 5e-3
        else:  # in camera 1
            want = [(P[0] - t01) @ R01, P[1]]
            assert np.allclose(poses[1], np.eye(4))
            assert np.abs(poses[0][:3, :3] - R01.T).max() < 2e-3 and np.abs(poses[0][:3, 3] + R01.T @ t01).max() < 5e-3
        for k in range(2):
            assert pts[k].shape == (H, W, 3) and pts[k].dtype == np.float32
            assert np.abs(pts[k] - want[k]).max() < 2e-2, np.abs(pts[k] - want[k]).max()


def test_duster_forward_stands_alone_without_upstream(monkeypatch):
    """`Duster._forward` with NO upstream package: the network is replaced by a perfect synthetic prediction, the aligner is the
    restatement, and the matches returned are pixel pairs that see the same 3-D point (checked against the ground-truth geometry)."""
    from imcui_hip import backend
    from imcui_hip.hloc.matchers.duster import Duster
    from imcui_hip.synth_weights import dust3r_state_dict

    out, (R01, t01, P, f, conf) = _scene(H=48, W=64)
    # one consistent world: image 1 sees the points of surface 0 moved into its frame where they project inside the image
    H, W = 48, 64

    def fake_forward(self, packed, net_cfg, images, pairs, dump=False, arith=0):
        return {"pts3d": torch.stack((out["pred1"]["pts3d"], out["pred2"]["pts3d_in_other_view"])), "conf": torch.stack((out["pred1"]["conf"], out["pred2"]["conf"]))}

    monkeypatch.setattr(backend.DUSt3RHIP, "forward", fake_forward)
    cfg = {"enc_dim": 128, "enc_depth": 1, "dec_dim": 64, "dec_depth": 4}
    model = Duster({"state_dict": dust3r_state_dict(41, cfg), "max_keypoints": 50}).eval()
    pred = model({"image0": torch.rand(1, 3, H, W), "image1": torch.rand(1, 3, H, W)})
    k0, k1 = pred["keypoints0"].numpy(), pred["keypoints1"].numpy()
    assert k0.shape == k1.shape and k0.shape[1] == 2 and len(k0) <= 50
    # every returned pair joins two pixels whose ground-truth 3-D points (camera 0's frame) are mutual nearest neighbours among the
    # confident points of the two clouds (brute force; the recovered cloud differs from the truth by ~1e-2, so near-ties may flip)
    m0, m1 = conf[0] * 1.0 > 3, conf[1] * 1.3 > 3
    C0, C1 = P[0][m0], (P[1] @ R01.T + t01)[m1]
    X0 = P[0][k0[:, 1], k0[:, 0]]
    X1 = (P[1] @ R01.T + t01)[k1[:, 1], k1[:, 0]]
    assert len(k0) > 10 and m0[k0[:, 1], k0[:, 0]].all() and m1[k1[:, 1], k1[:, 0]].all()
    d01 = np.linalg.norm(X0[:, None] - C1[None], axis=2)
    d10 = np.linalg.norm(X1[:, None] - C0[None], axis=2)
    ok = (np.abs(d01.min(1) - np.linalg.norm(X0 - X1, axis=1)) < 2e-2) & (np.abs(d10.min(1) - np.linalg.norm(X0 - X1, axis=1)) < 2e-2)
    assert ok.mean() > 0.9, ok.mean()
