"""Plane-level A/B of the attention-layout projection: gemm_split_kernel (option gemm_wreg = 0) vs gemm_wreg_kernel, repeatability.

The reproducer of the packed-f32 code-generation hazard of round 3 (v_pk_fma_f32 with op_sel in the rotary epilogue: wrong even
elements in lanes 48-63, different from run to run).  tests/test_gpu_round3_kernels.py::test_attention_layout_projection_planes is the
regression test; this script prints WHERE planes differ when it fails."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "image-matching-webui_amd")):
    sys.path.insert(0, p)
from imcui_hip import backend  # noqa: E402

dev = torch.device("cuda:0")
backend.set_precision(dev, 1)


def f16(planes):  # int16 [2, ...] -> float64 hi + lo
    a = planes.cpu().numpy().view(np.float16).astype(np.float64)
    return a[0] + a[1]


def where(d, name):
    idx = np.argwhere(d)
    if len(idx) == 0:
        return f"{name}: equal"
    uniq = [np.unique(idx[:, c]) for c in range(idx.shape[1])]
    return f"{name}: {len(idx)} elems differ; " + " | ".join(f"axis{c}: {len(u)} vals, first {u[:8].tolist()}" for c, u in enumerate(uniq))


for nseq, R, cross in ((64, 2048, 0), (64, 2048, 1), (3, 256, 0)):
    g = torch.Generator().manual_seed(nseq + cross)
    M = nseq * R
    x = torch.randn(M, 256, generator=g).to(dev)
    N = 512 if cross else 768
    w = torch.randn(N, 256, generator=g) / 16.0
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    ang = torch.rand(M, 32, generator=g) * 6.28
    cos, sin = torch.cos(ang).to(dev), torch.sin(ang).to(dev)
    cnt = torch.full((nseq,), R, dtype=torch.int32).to(dev)
    outs = {}
    for mode in ("0", "2"):
        backend.set_option(dev, "gemm_wreg", int(mode))
        runs = []
        for rep in range(3):
            q, k, v = backend.qkv_split_f32(x, w, b, cos, sin, cnt, R, 0.18, bool(cross))
            torch.cuda.synchronize()
            runs.append((q.clone(), k.clone(), v.clone()))
        outs[mode] = runs
        for rep in (1, 2):
            for name, a, c in zip("qkv", runs[0], runs[rep]):
                d = (a != c).cpu().numpy()
                if d.any():
                    print(f"nseq {nseq} cross {cross} WREG={mode} run0 vs run{rep}: " + where(d, name))
        print(f"nseq {nseq} cross {cross} WREG={mode}: repeatability checked")
    for name, a, c in zip("qkv", outs["0"][0], outs["2"][0]):
        fa, fc = f16(a), f16(c)
        err = np.abs(fa - fc)
        print(f"nseq {nseq} cross {cross} old vs wreg {name}: max abs diff {err.max():.3e} (|ref| max {np.abs(fa).max():.3e}); " + where(err > 1e-5 * max(np.abs(fa).max(), 1e-30), name))
    # float64 reference of q (cross: qk) rows 0..255 of sequence 0
    xr = x[:256].cpu().double()
    ref = xr @ w.double().t() + b.cpu().double()
    qr = ref[:, :256].reshape(256, 4, 64)
    if not cross:
        c2, s2 = cos[:256].cpu().double(), sin[:256].cpu().double()
        e, o = qr[..., 0::2], qr[..., 1::2]
        rot = torch.stack((e * c2[:, None, :] - o * s2[:, None, :], o * c2[:, None, :] + e * s2[:, None, :]), -1).reshape(256, 4, 64)
        qr = rot
    qr = (qr * 0.18).permute(1, 0, 2).numpy()
    for mode in ("0", "2"):
        got = f16(outs[mode][0][0])[0, :, :256]
        print(f"nseq {nseq} cross {cross} WREG={mode} q vs float64: max rel {np.abs(got - qr).max() / np.abs(qr).max():.3e}")
