"""CPU probe for VERDICT round 5, item 1, step A: can the two CORRECTION products of the 3 x f16 split,

    x . y  ~  xh . yh  +  xh . yl  +  xl . yh            (v_mfma_f32_32x32x16_f16, three times),

run on the block-scaled MX instruction `v_mfma_scale_f32_32x32x64_f8f6f4` instead -- operands quantised to 8- / 6- / 4-bit elements with
one e8m0 scale per 32 elements along K, at 2 x (fp8) or 4 x (fp6 / fp4) the f16 matrix rate -- without leaving the parity bar?

    x . y  ~  xh . yh  +  mx(xh) . mx(yl)  +  mx(xl) . mx(yh)        1 + 0.5 + 0.5 = 2.0 units (fp8) or 1 + 0.25 + 0.25 = 1.5 units (fp6 / fp4)

Nothing is dropped: the correction terms are 2^-11 of the result, so an element error of 2^-4 (e4m3 / e2m3: 4 significant bits) leaves
2^-15 per product, with random signs.  This script EMULATES that arithmetic in float64 on the CPU oracle's networks, per class of product:

  lin   LightGlue's projections and FFN GEMMs (K = 256 / 512): activations split while staged, weights split at pack time;
  pv    the second attention product O = P.V (P = exp2(s - m + 14) split into f16 planes; blocks of 32 along the keys);
  qk    the first attention product S = Q.K^T (its error is exponentiated);
  conv  SuperPoint's 3 x 3 convolutions (K = (tap, channel), blocks = the 32-channel chunks the kernel stages);

on the three LightGlue weight sets of the attention audits (N = M = 2048, two problems) and on SuperPoint (seeded weights, two synthetic
images), against a float64 evaluation of the same network ("truth") and against the f32 oracle.  Acceptance = the rule of
tools/attn_mix_audit.py (half the parity bar): worst per-layer token error <= 1.2e-5, matching-score error <= 5e-5, matches equal.

Calibration rows: "3xf16" (the arithmetic the library runs today), "f32" (the reference's own arithmetic: plain float32 products) and "2prod"
(the audited two-product P.V of round 5: measured on the GPU at 2.1e-5 / 9.6e-5 in the self blocks, 7.1e-6 / 3.6e-5 in the cross blocks).

Not modelled: the f32 accumulation order of the matrix pipe (1e-7 class), `v_exp_f32`'s last ulp, the deferred soft-max maximum.

    python tools/mx_corrections_probe.py > profiles/r06_lab_mx_corrections.txt          (CPU only, ~25 min on 8 cores)
"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-matching-webui_amd"), os.path.join(ROOT, "tests")]

from imcui_hip.synth import make_pair_batch  # noqa: E402
from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict  # noqa: E402
from oracle.lightglue import LightGlueOracle, apply_cached_rotary_emb, filter_matches, normalize_keypoints, sigmoid_log_double_softmax  # noqa: E402
from oracle.superpoint import SuperPointOracle, simple_nms  # noqa: E402
from parity_utils import synthetic_matching_problem  # noqa: E402

D = torch.float64
# element formats of the f8f6f4 instruction: (mantissa bits, exponent of the smallest normal, largest value)
FMT = {"e4m3": (3, -6, 448.0), "e5m2": (2, -14, 57344.0), "e2m3": (3, 0, 7.5), "e3m2": (2, -2, 28.0), "e2m1": (1, 0, 6.0)}


def q_elem(x, fmt):
    """Round-to-nearest-even onto the grid of an OCP element format (gradual underflow, saturation)."""
    m, emin, mx = FMT[fmt]
    a = x.abs()
    e = torch.floor(torch.log2(a.clamp_min(1e-300))).clamp_min(emin)
    quantum = torch.exp2(e - m)
    return torch.sign(x) * (torch.round(a / quantum) * quantum).clamp_max(mx)


def q_mx(x, fmt, block=32):
    """MX quantisation along the last axis: one power-of-two scale per `block` elements, chosen so the block maximum does not saturate."""
    mx = FMT[fmt][2]
    *lead, k = x.shape
    pad = (-k) % block
    if pad:
        x = F.pad(x, (0, pad))
    xb = x.reshape(*lead, -1, block)
    amax = xb.abs().amax(-1, keepdim=True)
    se = torch.ceil(torch.log2((amax / mx).clamp_min(1e-300)))
    scale = torch.where(amax > 0, torch.exp2(se), torch.ones_like(amax))
    out = (q_elem(xb / scale, fmt) * scale).reshape(*lead, -1)
    return out[..., :k] if pad else out


def q_i8(x, block=0, levels=127):
    """Symmetric integer quantisation along the last axis: one f32 scale per row (block = 0) or per `block` elements, `levels` steps to the maximum
    (v_mfma_i32_32x32x32_i8: i32 accumulation, the scales applied when the i32 sums are folded into the f32 accumulator)."""
    *lead, k = x.shape
    if block:
        pad = (-k) % block
        xb = (F.pad(x, (0, pad)) if pad else x).reshape(*lead, -1, block)
    else:
        xb = x.reshape(*lead, 1, k)
    amax = xb.abs().amax(-1, keepdim=True)
    scale = torch.where(amax > 0, amax / levels, torch.ones_like(amax))
    out = (torch.round(xb / scale).clamp(-levels, levels) * scale).reshape(*lead, -1)
    return out[..., :k]


def f16_rtn(x):
    return x.to(torch.float32).to(torch.float16).to(D)


def f16_rtz(x):
    """v_cvt_pkrtz_f16_f32 on normal values: the f32 mantissa truncated to 10 bits (saturating at 65504)."""
    b = x.to(torch.float32).clamp(-65504.0, 65504.0).view(torch.int32) & ~0x1FFF
    return b.view(torch.float32).to(D)


def split_act(x):
    """common.h split2: hi = rtz f16, lo = nearest f16 of the remainder (activations are not rescaled)."""
    x = x.to(torch.float32).to(D)
    hi = f16_rtz(x)
    return hi, f16_rtn(x - hi), 1.0


def split_weight(w):
    """gemm.hip split_weights_frag_host: scale 2^e puts max|w| into [4096, 8192]; hi = nearest f16, lo = nearest f16 of the remainder."""
    w = w.to(torch.float32).to(D)
    mxv = w.abs().max().item()
    e = max(-8, min(24, math.floor(math.log2(8192.0 / mxv)))) if mxv > 0 else 0
    ws = w * 2.0**e
    hi = f16_rtn(ws)
    return hi, f16_rtn(ws - hi), 2.0**e


def product(x, y, mode, xsplit=split_act, ysplit=split_act):
    """x [..., M, K] . y [..., N, K]^T in the arithmetic `mode`; the result is rounded to float32 (the accumulator)."""
    if mode == "f64":
        return x @ y.transpose(-1, -2)
    if mode == "f32":
        return (x.to(torch.float32) @ y.to(torch.float32).transpose(-1, -2)).to(D)
    xh, xl, sx = xsplit(x)
    yh, yl, sy = ysplit(y)
    yt = lambda t: t.transpose(-1, -2)  # noqa: E731
    if mode == "3xf16":
        r = xh @ yt(yh) + xh @ yt(yl) + xl @ yt(yh)
    elif mode == "2prod":  # round 5's variant 6 / 7: the x operand as ONE nearest f16, both planes of y
        xn = f16_rtn(x.to(torch.float32).to(D) * sx)
        r = xn @ yt(yh) + xn @ yt(yl)
    elif mode == "1xf16":
        r = f16_rtn(x.to(torch.float32).to(D) * sx) @ yt(f16_rtn(y.to(torch.float32).to(D) * sy))
    elif mode.startswith("mx:"):  # mx:<format of the copies of the hi planes>:<format of the lo planes>
        _, fh, fl = mode.split(":")
        r = xh @ yt(yh) + q_mx(xh, fh) @ yt(q_mx(yl, fl)) + q_mx(xl, fl) @ yt(q_mx(yh, fh))
    elif mode.startswith("i8:"):  # i8:<block along K of the scales, 0 = one scale per row>: both correction products on int8 operands
        blk = int(mode.split(":")[1])
        r = xh @ yt(yh) + q_i8(xh, blk) @ yt(q_i8(yl, blk)) + q_i8(xl, blk) @ yt(q_i8(yh, blk))
    elif mode.startswith("i8lo:"):  # only the lo planes on int8, their partners stay f16 (not an instruction: what the lo planes alone cost)
        blk = int(mode.split(":")[1])
        r = xh @ yt(yh) + xh @ yt(q_i8(yl, blk)) + q_i8(xl, blk) @ yt(yh)
    elif mode.startswith("pvfix:"):  # attention's P with CONSTANT scales (2^13 for P, 2^2 for P - ph) instead of one scale per 32 keys; V block-scaled
        _, fh, fl = mode.split(":")
        r = xh @ yt(yh) + (q_elem(x.to(torch.float32).to(D) / 8192.0, fh) * 8192.0) @ yt(q_mx(yl, fl)) + (q_elem(xl / 4.0, fl) * 4.0) @ yt(q_mx(yh, fh))
    else:
        raise ValueError(mode)
    return (r / (sx * sy)).to(torch.float32).to(D)


class Arith:
    def __init__(self, lin="f64", qk="f64", pv="f64", self_pv=None, cross_pv=None, assign=None, conv="f64"):
        self.lin, self.qk, self.assign, self.conv = lin, qk, assign or lin, conv
        self.self_pv, self.cross_pv = self_pv or pv, cross_pv or pv

    def __repr__(self):
        return f"lin={self.lin} qk={self.qk} pv(self)={self.self_pv} pv(cross)={self.cross_pv}"


def attention(q, k, v, qk_mode, pv_mode):
    """soft-max(q k^T) v as attention.hip evaluates it: q carries log2(e) (the 1/sqrt(d) is already in q, k), base-2 exponentials offset by 14
    so that P's f16 lo plane stays normal, the row sum taken over the unsplit P.  q [h, n, d], k, v [h, m, d]."""
    if qk_mode == "f64" and pv_mode == "f64":
        return torch.softmax(q @ k.transpose(-1, -2), -1) @ v
    s = product(q * math.log2(math.e), k, qk_mode)  # [h, n, m]
    mref = s.amax(-1, keepdim=True)
    p = torch.exp2(s - mref + 14.0).to(torch.float32).to(D)
    l = p.sum(-1, keepdim=True)
    o = product(p, v.transpose(-1, -2).contiguous(), pv_mode)  # P [h, n, m] . (V^T [h, d, m])^T
    return (o / l).to(torch.float32).to(D)


class EmuLightGlue(LightGlueOracle):
    """The oracle's blocks in float64 with the products of each class routed through `product`; element-wise work (LayerNorm, GELU, rotary,
    residuals) stays in float64 -- the library does it in float32, an error this probe leaves out on purpose (it is the same in every row)."""

    def __init__(self, sd, arith: Arith):
        super().__init__(sd, dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.1))
        self.sd = {k: v.to(D) for k, v in self.sd.items()}
        self.a = arith
        self._wsplit = {}

    def _w(self, key):
        if key not in self._wsplit:
            self._wsplit[key] = split_weight(self.sd[key])
        return self._wsplit[key]

    def _lin(self, x, prefix, mode=None):
        mode = mode or self.a.lin
        w = self.sd[prefix + ".weight"]
        if w.shape[0] < 8 or mode in ("f64", "f32"):  # (matchability / token confidence heads: one output feature, VALU on the device)
            r = product(x, w, "f64" if mode != "f32" else "f32")
        else:
            r = product(x, w, mode, ysplit=lambda _w, key=prefix + ".weight": self._w(key))
        b = self.sd.get(prefix + ".bias")
        return r if b is None else r + b

    def self_block(self, i, x, encoding):
        p = f"transformers.{i}.self_attn"
        qkv = self._lin(x, p + ".Wqkv").unflatten(-1, (4, -1, 3)).transpose(1, 2)
        q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
        q, k = apply_cached_rotary_emb(encoding, q), apply_cached_rotary_emb(encoding, k)
        ctx = attention(q[0] * 64**-0.5, k[0], v[0], self.a.qk, self.a.self_pv)[None]
        message = self._lin(ctx.transpose(1, 2).flatten(start_dim=-2), p + ".out_proj")
        return x + self._ffn(torch.cat([x, message], -1), p + ".ffn")

    def cross_block(self, i, x0, x1):
        p = f"transformers.{i}.cross_attn"
        qk0, qk1 = self._lin(x0, p + ".to_qk"), self._lin(x1, p + ".to_qk")
        v0, v1 = self._lin(x0, p + ".to_v"), self._lin(x1, p + ".to_v")
        qk0, qk1, v0, v1 = (t.unflatten(-1, (4, -1)).transpose(1, 2)[0] for t in (qk0, qk1, v0, v1))
        qk0, qk1 = qk0 * 64**-0.25, qk1 * 64**-0.25
        m0 = attention(qk0, qk1, v1, self.a.qk, self.a.cross_pv)[None]  # the kernel computes K.Q^T once per direction
        m1 = attention(qk1, qk0, v0, self.a.qk, self.a.cross_pv)[None]
        m0, m1 = (t.transpose(1, 2).flatten(start_dim=-2) for t in (m0, m1))
        m0, m1 = self._lin(m0, p + ".to_out"), self._lin(m1, p + ".to_out")
        return x0 + self._ffn(torch.cat([x0, m0], -1), p + ".ffn"), x1 + self._ffn(torch.cat([x1, m1], -1), p + ".ffn")

    def log_assignment(self, i, desc0, desc1):
        p = f"log_assignment.{i}"
        md0, md1 = self._lin(desc0, p + ".final_proj") / 256**0.25, self._lin(desc1, p + ".final_proj") / 256**0.25
        sim = product(md0, md1, self.a.assign if self.a.assign in ("f64", "f32") else "3xf16")  # (the assignment keeps three f16 products)
        z0, z1 = self._lin(desc0, p + ".matchability"), self._lin(desc1, p + ".matchability")
        return sigmoid_log_double_softmax(sim, z0, z1), sim

    @torch.no_grad()
    def run(self, k0, k1, d0, d1):
        x0, x1 = d0[None].to(D), d1[None].to(D)
        e0 = self.posenc(normalize_keypoints(k0[None].to(D), (640, 480)))
        e1 = self.posenc(normalize_keypoints(k1[None].to(D), (640, 480)))
        layers = []
        for i in range(9):
            x0, x1 = self.self_block(i, x0, e0), self.self_block(i, x1, e1)
            x0, x1 = self.cross_block(i, x0, x1)
            layers.append((x0[0].clone(), x1[0].clone()))
        scores, sim = self.log_assignment(8, x0, x1)
        m0, _, ms0, _ = filter_matches(scores, 0.1)
        return {"layers": layers, "m0": m0[0], "s0": ms0[0], "sim": sim[0]}


def compare(out, ref):
    layer = max(max((a - b).abs().max().item() / max(b.abs().max().item(), 1e-30) for a, b in zip(lo, lr)) for lo, lr in zip(out["layers"], ref["layers"]))
    same = out["m0"] == ref["m0"]
    score = (out["s0"] - ref["s0"]).abs()[same].max().item()
    return layer, score, int((~same).sum())


LG_ROWS = [
    ("f32 (plain float32 products: the reference's arithmetic)", Arith(lin="f32", qk="f32", pv="f32", assign="f32")),
    ("3xf16 everywhere (the library today, cross blocks on three products)", Arith(lin="3xf16", qk="3xf16", pv="3xf16")),
    ("calibration: 2-product P.V in the self blocks", Arith(lin="3xf16", qk="3xf16", pv="3xf16", self_pv="2prod")),
    ("calibration: 2-product P.V in the cross blocks (round 5 default)", Arith(lin="3xf16", qk="3xf16", pv="3xf16", cross_pv="2prod")),
    ("lin mx fp8 (e4m3 hi copies, e4m3 lo)", Arith(lin="mx:e4m3:e4m3", qk="3xf16", pv="3xf16")),
    ("lin mx fp6 (e2m3, e2m3)", Arith(lin="mx:e2m3:e2m3", qk="3xf16", pv="3xf16")),
    ("lin mx fp6 (e3m2, e3m2)", Arith(lin="mx:e3m2:e3m2", qk="3xf16", pv="3xf16")),
    ("lin mx fp4 (e2m1, e2m1)", Arith(lin="mx:e2m1:e2m1", qk="3xf16", pv="3xf16")),
    ("lin ONE f16 product (what dropping both corrections costs)", Arith(lin="1xf16", qk="3xf16", pv="3xf16")),
    ("pv mx fp8, self + cross", Arith(lin="3xf16", qk="3xf16", pv="mx:e4m3:e4m3")),
    ("pv mx fp6 (e2m3), self + cross", Arith(lin="3xf16", qk="3xf16", pv="mx:e2m3:e2m3")),
    ("pv mx fp4 (e2m1), self + cross", Arith(lin="3xf16", qk="3xf16", pv="mx:e2m1:e2m1")),
    ("qk mx fp8", Arith(lin="3xf16", qk="mx:e4m3:e4m3", pv="3xf16")),
    ("qk mx fp6 (e2m3)", Arith(lin="3xf16", qk="mx:e2m3:e2m3", pv="3xf16")),
    ("lin + pv mx fp8", Arith(lin="mx:e4m3:e4m3", qk="3xf16", pv="mx:e4m3:e4m3")),
    ("lin + pv mx fp6 (e2m3)", Arith(lin="mx:e2m3:e2m3", qk="3xf16", pv="mx:e2m3:e2m3")),
    ("lin + pv + qk mx fp8", Arith(lin="mx:e4m3:e4m3", qk="mx:e4m3:e4m3", pv="mx:e4m3:e4m3")),
]


def lightglue_part(n=2048, rows=LG_ROWS):
    weights = {"damped": lightglue_state_dict(0), "strong": lightglue_state_dict(0, damp=0.1, ln_noise=0.1, final_gain=10.0),
               "random": lightglue_state_dict(1, structured=False)}  # fmt: skip
    problems = [synthetic_matching_problem(40, n, n, int(n * 0.146)), synthetic_matching_problem(41, n, n - n // 14, n // 8)]
    print(f"## LightGlue, N = M = {n}, 9 layers, {len(problems)} problems x 3 weight sets; errors against the float64 evaluation (in brackets: against the f32 oracle)")
    print("## layer = worst per-layer max|dx| / max|x|; score = worst |d matching_scores0| over the points matched alike; diff = points whose match differs")
    print("## accept: layer <= 1.2e-5 and score <= 5e-5 and diff = 0 against float64 on every weight set (the rule of tools/attn_mix_audit.py)\n")
    worst = {name: [0.0, 0.0, 0] for name, _ in rows}
    for wname, sd in weights.items():
        truths, f32s = [], []
        for pr in problems:
            truths.append(EmuLightGlue(sd, Arith()).run(*pr))
            ora = LightGlueOracle(sd, dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.1))
            img = torch.zeros(1, 1, 480, 640)
            r = ora({"image0": img, "image1": img, "keypoints0": pr[0][None], "keypoints1": pr[1][None], "descriptors0": pr[2].t()[None], "descriptors1": pr[3].t()[None]},
                    return_intermediates=True)  # fmt: skip
            f32s.append({"layers": [(a[0].to(D), b[0].to(D)) for a, b in r["_layers"]], "m0": r["matches0"][0], "s0": r["matching_scores0"][0].to(D)})
        t = truths[0]
        print(f"# weight set {wname}: |sim| max {t['sim'].abs().max().item():.1f}, matched {int((t['m0'] > -1).sum())} of {n}; the f32 oracle itself vs float64: "
              + ", ".join("layer %.2e score %.2e diff %d" % compare(f, tr) for f, tr in zip(f32s, truths)))
        for name, ar in rows:
            t0 = time.time()
            res, res32 = [0.0, 0.0, 0], [0.0, 0.0, 0]
            for pr, tr, f3 in zip(problems, truths, f32s):
                out = EmuLightGlue(sd, ar).run(*pr)
                for acc, ref in ((res, tr), (res32, f3)):
                    c = compare(out, ref)
                    acc[0], acc[1], acc[2] = max(acc[0], c[0]), max(acc[1], c[1]), acc[2] + c[2]
            w = worst[name]
            w[0], w[1], w[2] = max(w[0], res[0]), max(w[1], res[1]), w[2] + res[2]
            ok = res[0] <= 1.2e-5 and res[1] <= 5e-5 and res[2] == 0
            print(f"{wname:7s} {name:70s} layer {res[0]:.2e} [{res32[0]:.2e}]  score {res[1]:.2e} [{res32[1]:.2e}]  diff {res[2]} [{res32[2]}]  "
                  f"{'ok' if ok else 'FAIL'}   ({time.time() - t0:.0f} s)", flush=True)
        print()
    print("## verdict per row (worst over the three weight sets, against float64)")
    for name, _ in rows:
        w = worst[name]
        print(f"{name:70s} layer {w[0]:.2e}  score {w[1]:.2e}  diff {w[2]}  -> {'ACCEPT' if w[0] <= 1.2e-5 and w[1] <= 5e-5 and w[2] == 0 else 'reject'}")
    print()


# ------------------------------------------------------------------------------------------------------------------- SuperPoint
class EmuSuperPoint(SuperPointOracle):
    def __init__(self, sd, mode):
        super().__init__(sd)
        self.sd = {k: v.to(D) for k, v in self.sd.items()}
        self.mode = mode

    def _conv(self, x, name, relu=True, pad=1):
        w, b = self.sd[name + ".weight"], self.sd[name + ".bias"]
        co, ci, kh, kw = w.shape
        if ci < 32 or self.mode == "f64" or kh == 1 and False:
            y = F.conv2d(x, w, b, padding=pad)  # conv1a (1 -> 64) is a VALU kernel on the device: exact f32 products
        else:
            B, _, H, W = x.shape
            cols = F.unfold(x, (kh, kw), padding=pad).reshape(B, ci, kh * kw, H * W).permute(0, 3, 2, 1).reshape(B, H * W, kh * kw * ci)  # K = (tap, channel)
            wm = w.reshape(co, ci, kh * kw).permute(0, 2, 1).reshape(co, kh * kw * ci)
            y = product(cols, wm, self.mode, ysplit=split_weight)  # [B, HW, co]
            y = y.transpose(1, 2).reshape(B, co, H, W) + b.view(1, -1, 1, 1)
        return F.relu(y) if relu else y

    @torch.no_grad()
    def run(self, image):
        feat = self.encoder(image.to(D))
        return {"dense": self.score_map(feat)[0], "desc": self.dense_descriptors(feat)[0]}


def superpoint_part(h=240, w=320):
    sd = superpoint_state_dict(0)
    img0, img1, _ = make_pair_batch(5, 1, h, w, n_blobs=300)
    rows = ["f32", "3xf16", "mx:e4m3:e4m3", "mx:e2m3:e2m3", "mx:e3m2:e3m2", "mx:e2m1:e2m1", "1xf16"]
    print(f"## SuperPoint, {h} x {w}, two images; every 3 x 3 / 1 x 1 convolution with Cin >= 64 in the named arithmetic (conv1a is exact f32 on the device)")
    print("## dense = max |d score map| (tests: < 2e-5), desc = max |d dense descriptor| (bar 1e-4), kp = key-points (threshold 0.005, NMS 3) that differ from float64's")
    for mode in rows:
        t0 = time.time()
        res = [0.0, 0.0, 0, 0]
        for im in (img0, img1):
            tr = EmuSuperPoint(sd, "f64").run(im)
            out = EmuSuperPoint(sd, mode).run(im)
            res[0] = max(res[0], (out["dense"] - tr["dense"]).abs().max().item())
            res[1] = max(res[1], (out["desc"] - tr["desc"]).abs().max().item())
            ka = simple_nms(out["dense"][None].float(), 3)[0] > 0.005
            kb = simple_nms(tr["dense"][None].float(), 3)[0] > 0.005
            res[2] += int((ka != kb).sum())
            res[3] += int(kb.sum())
        print(f"conv {mode:16s} dense {res[0]:.2e}  desc {res[1]:.2e}  kp differ {res[2]} of {res[3]}   ({time.time() - t0:.0f} s)", flush=True)
    print()


if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("PROBE_THREADS", "8")))
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    n = int(os.environ.get("PROBE_N", "2048"))
    print("# MX-corrections probe (tools/mx_corrections_probe.py): float64 emulation of  xh.yh + mx(xh).mx(yl) + mx(xl).mx(yh)  per class of product")
    if what in ("all", "superpoint"):
        superpoint_part()
    if what in ("all", "lightglue"):
        lightglue_part(n)
