"""Per-launch spending of the attention parity slack: the two-product P.V (variant 7: K.Q^T in three products,
V^T.P^T as (vh + vl) . f16(P)) was rejected as an all-or-nothing default in round 4 -- layer error 2.12e-5 and score error 9.1e-5 on the
adversarial ("strong") weight set.  Its error accumulates per launch and its speed-up is per launch, so this tool measures the MIXES:
the reduced product only in the self blocks, only in the cross blocks, only in layers 0-3, only in layers 5-8 (and everywhere / nowhere),
every other launch on the default three-product kernel (variant 8).

Per mix: LightGlue at N = M = 2048, all 9 layers, two pairs, on the three weight sets of tests/test_gpu_lightglue.py -- worst per-layer
token error against the CPU oracle, matches (equal or audited ties), score error -- and the headline-shaped step time (64 pairs, HIP events).
Acceptance rule (half the parity bar): layer error <= 1.2e-5 AND score error <= 5e-5 on every weight set.

    python tools/attn_mix_audit.py > gpurun_out/.../lab_attention_mix.txt      (GPU box)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-matching-webui_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

from imcui_hip import backend  # noqa: E402
from imcui_hip.hloc.matchers.lightglue import LightGlue  # noqa: E402
from imcui_hip.synth_weights import lightglue_state_dict  # noqa: E402
from oracle.lightglue import LightGlueOracle  # noqa: E402
from parity_utils import assert_matches_equal_or_tied, synthetic_matching_problem  # noqa: E402

dev = torch.device("cuda:0")
backend.set_precision(dev, 1)
torch.set_num_threads(min(os.cpu_count() or 1, 32))
RED = int(os.environ.get("AUDIT_REDUCED", "7"))
MIXES = [("none (default: three products everywhere)", -1, -1, 0x1FF), ("self blocks only", RED, -1, 0x1FF), ("cross blocks only", -1, RED, 0x1FF),
         ("layers 0-3 (self + cross)", RED, RED, 0x00F), ("layers 5-8 (self + cross)", RED, RED, 0x1E0), ("self blocks of layers 0-3", RED, -1, 0x00F),
         ("cross blocks of layers 5-8", -1, RED, 0x1E0), ("everywhere", RED, RED, 0x1FF)]  # fmt: skip


def set_mix(sv, cv, mask):
    backend.set_option(dev, "attn_variant_self", sv)
    backend.set_option(dev, "attn_variant_cross", cv)
    backend.set_option(dev, "attn_mix_layers", mask)


WEIGHTS = {"damped": lightglue_state_dict(0), "strong": lightglue_state_dict(0, damp=0.1, ln_noise=0.1, final_gain=10.0), "random": lightglue_state_dict(1, structured=False)}
IMG = torch.zeros(1, 1, 480, 640)
problems = [synthetic_matching_problem(40, 2048, 2048, 300), synthetic_matching_problem(41, 2048, 1900, 250)]
B = len(problems)
k0, k1, d0, d1 = torch.zeros(B, 2048, 2), torch.zeros(B, 2048, 2), torch.zeros(B, 2048, 256), torch.zeros(B, 2048, 256)
n0, n1 = torch.zeros(B, dtype=torch.int32), torch.zeros(B, dtype=torch.int32)
for b, (a, c, e, f) in enumerate(problems):
    k0[b, : len(a)], k1[b, : len(c)], d0[b, : len(a)], d1[b, : len(c)] = a, c, e, f
    n0[b], n1[b] = len(a), len(c)
results = {m[0]: {"layer": 0.0, "score": 0.0, "abs": 0.0, "match": "equal"} for m in MIXES}
print(f"# reduced-product variant {RED} (two-product P.V), default variant 8; LightGlue N = M = 2048, 9 layers, 2 pairs x 3 weight sets, HIP vs the CPU oracle")
for wname, sd in WEIGHTS.items():
    ora = LightGlueOracle(sd, dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.1))
    refs = [ora({"image0": IMG, "image1": IMG, "keypoints0": a[None], "keypoints1": c[None], "descriptors0": e.t()[None], "descriptors1": f.t()[None]},
                return_intermediates=True) for (a, c, e, f) in problems]  # fmt: skip
    model = LightGlue({"depth_confidence": -1, "width_confidence": -1, "match_threshold": 0.1, "state_dict": sd}).eval().to(dev)
    for name, sv, cv, mask in MIXES:
        set_mix(sv, cv, mask)
        out = model.forward_batched(k0.cuda(), k1.cuda(), d0.cuda(), d1.cuda(), n0.cuda(), n1.cuda(), (640, 480), (640, 480), layer_dump=True)
        torch.cuda.synchronize()
        dump = out.pop("_layers")
        out = {kk: vv.cpu() for kk, vv in out.items()}
        for b, ((a, c, e, f), ref) in enumerate(zip(problems, refs)):
            na = len(a)
            per_layer = []
            for li, (r0, r1) in enumerate(ref["_layers"]):
                errs = [(dump[li, 2 * b + s, : r.shape[0]].cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-30) for s, r in enumerate((r0[0], r1[0]))]
                per_layer.append(max(errs))
            tol = 1e-4 * max(1.0, ref["_sim"].abs().max().item() / 100.0)
            try:
                ties = assert_matches_equal_or_tied(out["matches0"][b, :na], ref["_log_assignment"][0], ref["matches0"][0], 0.1, tol=tol, tag=f"{wname} {name}",
                                                    ind0=ref.get("_ind0"), ind1=ref.get("_ind1"))  # fmt: skip
                mstat = f"matches equal ({ties} ties)"
            except AssertionError as ex:
                mstat = f"MATCHES DIFFER: {str(ex)[:100]}"
                results[name]["match"] = "DIFFER"
            same = out["matches0"][b, :na].long() == ref["matches0"][0]
            serr = (out["matching_scores0"][b, :na] - ref["matching_scores0"][0]).abs()[same].max().item()
            rel = serr / tol  # in units of this weight set's bar (1e-4 scaled by |sim| / 100)
            results[name]["layer"] = max(results[name]["layer"], max(per_layer))
            results[name]["score"] = max(results[name]["score"], rel)
            if wname != "random":  # (plain random weights push |sim| to ~2000: their bar is 2.3e-3 and the default kernel itself measures 1.5e-4 there)
                results[name]["abs"] = max(results[name]["abs"], serr)
            print(f"  {wname:7s} pair {b} {name:42s}: worst layer error {max(per_layer):.2e}; {mstat}; score error {serr:.2e} (bar {tol:.1e})", flush=True)

# ---- speed: the headline's step shape
print("# headline-shaped step (64 pairs, SuperPoint + LightGlue 640x480, all 9 layers), HIP events over 10 steps after 3 warm-ups")
from imcui_hip.pipeline import SuperPointLightGluePipeline  # noqa: E402
from imcui_hip.synth import make_pair_batch  # noqa: E402
from imcui_hip.synth_weights import superpoint_state_dict  # noqa: E402

pipe = SuperPointLightGluePipeline({"nms_radius": 3, "max_keypoints": 2048, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)},
                                   {"depth_confidence": -1.0, "width_confidence": -1.0, "match_threshold": 0.1, "state_dict": lightglue_state_dict(0)}).eval().to(dev)  # fmt: skip
img0, img1, _ = make_pair_batch(1234, 64, 480, 640, distinct=8)
img0, img1 = img0.to(dev), img1.to(dev)
base = None
for name, sv, cv, mask in MIXES:
    set_mix(sv, cv, mask)
    for _ in range(3):
        pipe(img0, img1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        pipe(img0, img1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    base = base or ms
    r = results[name]
    ok = r["layer"] <= 1.2e-5 and r["score"] <= 0.5 and r["abs"] <= 5e-5 and r["match"] == "equal"
    print(f"  {name:42s}: {ms:7.2f} ms / step = {64e3 / ms:7.1f} pairs/s ({100 * (base / ms - 1):+5.2f} %); worst layer error {r['layer']:.2e}, worst score error {r['score']:.2f} x the bar ({r['abs']:.1e} on the shaped sets), "
          f"matches {r['match']} -> {'ACCEPT' if ok else 'reject'} (rule: layer <= 1.2e-5, score <= 5e-5 and <= 0.5 x bar)", flush=True)
set_mix(-1, -1, 0x1FF)
