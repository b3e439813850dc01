"""Host-side restatement of upstream DUSt3R's `PairViewer` -- the scene object `global_aligner(output, device, mode=PairViewer)`
returns for ONE symmetrised pair (imcui/hloc/matchers/duster.py:74-79: `scene.imgs`, `scene.get_masks()`, `scene.get_pts3d()`).

PARITY UNPINNED.  Upstream's `dust3r.cloud_opt` is not in /root/reference (empty submodule) and cv2 is not installed, so nothing here
can be checked against the original; the steps follow the public upstream source (dust3r/cloud_opt/pair_viewer.py,
base_opt.py, dust3r/post_process.py `estimate_focal_knowing_depth`, dust3r/utils/geometry.py) as summarised below and are tested
on synthetic two-camera scenes with a known focal length and pose (tests/test_pair_viewer_cpu.py).  One step cannot be reproduced
even in principle: upstream calls `cv2.solvePnPRansac(..., iterationsCount=100, reprojectionError=5, flags=SOLVEPNP_SQPNP)`, whose
sampling is cv2's own; here the pose comes from a seeded numpy RANSAC over 6-point DLT hypotheses with the same threshold and
iteration count, refined by Gauss-Newton on the inliers -- the same model and the same inlier rule, a different sampler, so poses agree
to the accuracy the data supports, not bit for bit.  `Duster.aligner` uses upstream's package whenever it is importable.

For a symmetrised pair (edges (1, 0) and (0, 1) in `make_pairs`' order), with pred_i[e] = view-1 point map of edge e (own frame),
pred_j[e] = view-2 point map of edge e (in view 1's frame), conf_i / conf_j the confidences:

  * masks:   im_conf[k] = max over the edges of the confidence map of image k;  mask_k = im_conf[k] > min_conf_thr (3).
  * per image k:  conf_k = mean(conf_i[k, 1-k]) * mean(conf_j[k, 1-k]);  focal_k = Weiszfeld focal of pred_i[k, 1-k] about the
    principal point (W/2, H/2);  pose_k = inverse of the PnP pose of pred_j[1-k, k][mask_k] (image k's points in camera 1-k's frame)
    against the pixels of image k with K = [[f, 0, W/2], [0, f, H/2], [0, 0, 1]] -- i.e. camera k -> camera (1-k); identity if PnP fails.
  * the edge with the larger conf_k anchors the frame:  conf_0 > conf_1 -> poses (I, pose_1), depths (z of pred_i[0,1],
    z of pred_j[0,1] moved into camera 1); else poses (pose_0, I), depths (z of pred_j[1,0] moved into camera 0, z of pred_i[1,0]).
  * get_pts3d(): every depth map back-projected through its pinhole (focal_k, principal point) and moved by its pose.
"""
from __future__ import annotations

import numpy as np
import torch

MIN_CONF_THR = 3.0  # BasePCOptimizer(min_conf_thr=3)
PNP_ITERATIONS = 100
PNP_REPROJECTION_ERROR = 5.0


def estimate_focal_knowing_depth(pts3d: torch.Tensor, pp: torch.Tensor, min_focal: float = 0.0, max_focal: float = float("inf")) -> float:
    """`focal_mode='weiszfeld'` of upstream's estimate_focal_knowing_depth: focal = argmin sum |pixel - focal (x, y) / z| over the
    point map [H, W, 3] of a camera looking down +z, pixels taken relative to the principal point; closed-form L2 start, ten
    re-weighted least-squares steps (weights 1 / distance, clipped at 1e-8)."""
    H, W, _ = pts3d.shape
    xs, ys = torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="xy")
    pixels = torch.stack((xs, ys), -1).reshape(-1, 2) - pp.reshape(1, 2).float()
    p = pts3d.reshape(-1, 3).float()
    xy_over_z = torch.nan_to_num(p[:, :2] / p[:, 2:3], posinf=0.0, neginf=0.0)
    dot_xy_px = (xy_over_z * pixels).sum(-1)
    dot_xy_xy = xy_over_z.square().sum(-1)
    focal = dot_xy_px.mean() / dot_xy_xy.mean()
    for _ in range(10):
        dis = (pixels - focal * xy_over_z).norm(dim=-1)
        w = dis.clip(min=1e-8).reciprocal()
        focal = (w * dot_xy_px).mean() / (w * dot_xy_xy).mean()
    focal_base = max(H, W) / (2 * np.tan(np.deg2rad(60) / 2))
    return float(focal.clip(min=min_focal * focal_base, max=max_focal * focal_base))


def _project_to_rotation(M: np.ndarray):
    """Nearest rotation to the 3x3 block of a DLT solution and the scale that was divided out."""
    U, S, Vt = np.linalg.svd(M)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R, S = -R, -S
    return R, float(np.mean(S))


def _dlt_pose(X: np.ndarray, xn: np.ndarray):
    """[R | t] with xn ~ R X + t from >= 6 correspondences (object points X [n,3], normalised image points xn [n,2]): direct
    linear transform, the 3x3 block projected onto SO(3).  None when the system is degenerate."""
    n = len(X)
    A = np.zeros((2 * n, 12))
    Xh = np.concatenate((X, np.ones((n, 1))), 1)
    A[0::2, 0:4] = Xh
    A[0::2, 8:12] = -xn[:, 0:1] * Xh
    A[1::2, 4:8] = Xh
    A[1::2, 8:12] = -xn[:, 1:2] * Xh
    try:
        _, s, Vt = np.linalg.svd(A)
    except np.linalg.LinAlgError:
        return None
    P = Vt[-1].reshape(3, 4)
    R, sc = _project_to_rotation(P[:, :3])
    if not np.isfinite(sc) or abs(sc) < 1e-12:
        return None
    t = P[:, 3] / sc
    if not np.all(np.isfinite(t)) or np.mean((X @ R.T + t)[:, 2]) <= 0:  # (the sign of the null vector went into `sc`: det R = +1)
        return None
    return R, t


def _reprojection_error(R, t, X, x, f, pp):
    Xc = X @ R.T + t
    z = Xc[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        uv = f * Xc[:, :2] / z[:, None] + pp
    err = np.linalg.norm(uv - x, axis=1)
    err[~(z > 1e-9)] = np.inf
    return err


def _rodrigues(w: np.ndarray) -> np.ndarray:
    th = np.linalg.norm(w)
    Kx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + Kx
    return np.eye(3) + np.sin(th) / th * Kx + (1 - np.cos(th)) / th**2 * (Kx @ Kx)


def _refine_pose(R, t, X, x, f, pp, iters: int = 10):
    """Gauss-Newton on the pixel reprojection error over (rotation vector, translation), left-multiplicative rotation update."""
    for _ in range(iters):
        Xc = X @ R.T + t
        z = Xc[:, 2]
        r = (f * Xc[:, :2] / z[:, None] + pp - x).reshape(-1)
        n = len(X)
        J = np.zeros((2 * n, 6))
        iz, iz2 = 1.0 / z, 1.0 / (z * z)
        dpx = np.stack((f * iz, np.zeros(n), -f * Xc[:, 0] * iz2), 1)  # d u / d Xc
        dpy = np.stack((np.zeros(n), f * iz, -f * Xc[:, 1] * iz2), 1)
        # d Xc / d w = -[Xc]x for R <- exp([w]x) R ; d Xc / d t = I
        cx = np.stack((np.zeros(n), Xc[:, 2], -Xc[:, 1], -Xc[:, 2], np.zeros(n), Xc[:, 0], Xc[:, 1], -Xc[:, 0], np.zeros(n)), 1).reshape(n, 3, 3)
        J[0::2, :3] = np.einsum("nk,nkj->nj", dpx, cx)
        J[1::2, :3] = np.einsum("nk,nkj->nj", dpy, cx)
        J[0::2, 3:] = dpx
        J[1::2, 3:] = dpy
        try:
            d = np.linalg.lstsq(J, -r, rcond=None)[0]
        except np.linalg.LinAlgError:
            break
        R = _rodrigues(d[:3]) @ R
        t = t + d[3:]
        if np.linalg.norm(d) < 1e-10:
            break
    return R, t


def solve_pnp_ransac(X: np.ndarray, x: np.ndarray, f: float, pp, iterations: int = PNP_ITERATIONS, reproj: float = PNP_REPROJECTION_ERROR, seed: int = 0):
    """World-to-camera pose (R [3,3], t [3]) with x ~ K (R X + t), K = [[f, 0, ppx], [0, f, ppy], [0, 0, 1]]: seeded RANSAC over
    6-point DLT hypotheses, inliers = reprojection error below `reproj` pixels in front of the camera, the winner refined on its
    inliers (stands in for cv2.solvePnPRansac(..., iterationsCount=100, reprojectionError=5, flags=SOLVEPNP_SQPNP), see the module
    docstring).  Returns (R, t, inlier mask) or None."""
    X, x = np.asarray(X, dtype=np.float64), np.asarray(x, dtype=np.float64)
    pp = np.asarray(pp, dtype=np.float64)
    n = len(X)
    if n < 6:
        return None
    xn = (x - pp) / f
    rng = np.random.default_rng(seed)
    best, best_cnt = None, 0
    for _ in range(iterations):
        pick = rng.choice(n, 6, replace=False)
        sol = _dlt_pose(X[pick], xn[pick])
        if sol is None:
            continue
        inl = _reprojection_error(sol[0], sol[1], X, x, f, pp) < reproj
        cnt = int(inl.sum())
        if cnt > best_cnt:
            best, best_cnt = (sol, inl), cnt
            if cnt == n:
                break
    if best is None or best_cnt < 6:
        return None
    (R, t), inl = best
    for _ in range(2):  # re-estimate on the consensus set, re-collect, refine
        sol = _dlt_pose(X[inl], xn[inl])
        if sol is not None:
            R2, t2 = _refine_pose(sol[0], sol[1], X[inl], x[inl], f, pp)
            inl2 = _reprojection_error(R2, t2, X, x, f, pp) < reproj
            if inl2.sum() >= inl.sum():
                R, t, inl = R2, t2, inl2
    R, t = _refine_pose(R, t, X[inl], x[inl], f, pp)
    return R, t, inl


def _geotrf(T: np.ndarray, pts: np.ndarray) -> np.ndarray:
    return pts @ T[:3, :3].T + T[:3, 3]


def depthmap_to_absolute_camera_coordinates(depth: np.ndarray, f: float, pp, cam2world: np.ndarray) -> np.ndarray:
    """upstream dust3r.utils.geometry: back-project a depth map through the pinhole (f, f, pp) and move it by the camera pose."""
    H, W = depth.shape
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    x = (u - pp[0]) * depth / f
    y = (v - pp[1]) * depth / f
    cam = np.stack((x, y, depth), -1).astype(np.float32)
    return (np.einsum("ik,vuk->vui", cam2world[:3, :3].astype(np.float32), cam) + cam2world[:3, 3].astype(np.float32)).astype(np.float32)


class PairViewerScene:
    """The three things imcui/hloc/matchers/duster.py:76-79 reads from the aligner's result."""

    def __init__(self, output: dict, min_conf_thr: float = MIN_CONF_THR, seed: int = 0):
        def entry(v, e):  # batch entry e of a collated tensor, or element e of a `lists=True` collation
            return v[e]

        idx1, idx2 = list(output["view1"]["idx"]), list(output["view2"]["idx"])
        edges = [(int(a), int(b)) for a, b in zip(idx1, idx2)]
        if sorted(edges) != [(0, 1), (1, 0)]:
            raise ValueError(f"PairViewer needs the two directed pairs of one symmetrised image pair, got edges {edges}")
        cpu = lambda t: torch.as_tensor(t).detach().float().cpu()  # noqa: E731
        pred_i = {e: cpu(entry(output["pred1"]["pts3d"], n)) for n, e in enumerate(edges)}
        pred_j = {e: cpu(entry(output["pred2"]["pts3d_in_other_view"], n)) for n, e in enumerate(edges)}
        conf_i = {e: cpu(entry(output["pred1"]["conf"], n)) for n, e in enumerate(edges)}
        conf_j = {e: cpu(entry(output["pred2"]["conf"], n)) for n, e in enumerate(edges)}
        shapes = {}
        for n, (a, b) in enumerate(edges):
            shapes[a], shapes[b] = tuple(pred_i[(a, b)].shape[:2]), tuple(pred_j[(a, b)].shape[:2])
        self.imshapes = [shapes[0], shapes[1]]
        # scene.imgs: only the shapes are read by the wrapper (xy_grid(*imgs[i].shape[:2][::-1])); the pixels are the inputs in [0, 1]
        self.imgs = []
        for k in range(2):
            view, n = ("view1", edges.index((k, 1 - k)))
            im = cpu(entry(output[view]["img"], n))
            self.imgs.append(np.clip(im.permute(1, 2, 0).numpy() * 0.5 + 0.5, 0.0, 1.0))
        im_conf = [torch.zeros(self.imshapes[k]) for k in range(2)]
        for (a, b) in edges:
            im_conf[a] = torch.maximum(im_conf[a], conf_i[(a, b)])
            im_conf[b] = torch.maximum(im_conf[b], conf_j[(a, b)])
        self.im_conf = im_conf
        self.min_conf_thr = float(min_conf_thr)
        masks = [c > self.min_conf_thr for c in im_conf]

        self.focals, self.pp, confs, rel_poses, self.pnp_inliers = [], [], [], [], []
        for k in range(2):
            e, er = (k, 1 - k), (1 - k, k)
            confs.append(float(conf_i[e].mean() * conf_j[e].mean()))
            H, W = self.imshapes[k]
            pp = torch.tensor((W / 2, H / 2))
            focal = estimate_focal_knowing_depth(pred_i[e], pp)
            if not (np.isfinite(focal) and focal > 0):  # a point map that is not a camera's view (untrained weights): the 60-degree default
                focal = max(H, W) / (2 * np.tan(np.deg2rad(60) / 2))
            self.focals.append(focal)
            self.pp.append(pp.numpy())
            pixels = np.mgrid[:W, :H].T.astype(np.float32)  # [H, W, 2] = (x, y)
            pts = pred_j[er].numpy()  # image k's points in camera (1 - k)'s frame
            msk = masks[k].numpy()
            pose = np.eye(4)
            sol = solve_pnp_ransac(pts[msk], pixels[msk], focal, self.pp[-1], seed=seed + k) if msk.sum() >= 6 and np.isfinite(focal) and focal > 0 else None
            if sol is not None:
                R, t, inl = sol
                w2c = np.eye(4)
                w2c[:3, :3], w2c[:3, 3] = R, t
                pose = np.linalg.inv(w2c)  # camera k -> camera (1 - k)
                self.pnp_inliers.append(int(inl.sum()))
            else:
                self.pnp_inliers.append(0)
            rel_poses.append(pose.astype(np.float32))
        self.confs = confs
        if confs[0] > confs[1]:  # the point cloud is expressed in camera 0
            self.im_poses = [np.eye(4, dtype=np.float32), rel_poses[1]]
            self.depth = [pred_i[(0, 1)][..., 2].numpy(), _geotrf(np.linalg.inv(rel_poses[1]), pred_j[(0, 1)].numpy())[..., 2]]
        else:  # in camera 1
            self.im_poses = [rel_poses[0], np.eye(4, dtype=np.float32)]
            self.depth = [_geotrf(np.linalg.inv(rel_poses[0]), pred_j[(1, 0)].numpy())[..., 2], pred_i[(1, 0)][..., 2].numpy()]

    def get_masks(self):
        return [c > self.min_conf_thr for c in self.im_conf]

    def get_pts3d(self):
        return [torch.from_numpy(depthmap_to_absolute_camera_coordinates(np.asarray(d, dtype=np.float32), f, pp, P))
                for d, f, pp, P in zip(self.depth, self.focals, self.pp, self.im_poses)]  # fmt: skip

    def get_focals(self):
        return list(self.focals)

    def get_im_poses(self):
        return [torch.from_numpy(np.asarray(p, dtype=np.float32)) for p in self.im_poses]
