"""Host logic of bench.py that needs no GPU: which invocations carry the legs of the other BASELINE configs, the leg table, the
locked self-build, the rank-agreement guard on one rank."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    base = dict(workload="splg", no_parity=False, no_cpu_baseline=False, adaptive=False, graph=False, h2d=None, precision=1, batch_given=None, gpus=1, legs=None,
                no_legs=False)  # fmt: skip
    base.update(kw)
    return argparse.Namespace(**base)


def test_which_invocations_carry_the_legs():
    assert bench.legs_enabled(_args())  # the driver's plain command line
    assert not bench.legs_enabled(_args(no_legs=True))
    for kw in (dict(no_parity=True), dict(no_cpu_baseline=True), dict(adaptive=True), dict(graph=True), dict(h2d="raw"), dict(h2d="jpeg"), dict(precision=0),
               dict(batch_given=32), dict(workload="loftr"), dict(gpus=8)):  # fmt: skip
        assert not bench.legs_enabled(_args(**kw)), kw  # profiler passes, A/B legs, other workloads, the scaling runs
    assert bench.legs_enabled(_args(gpus=8, legs="all")) and bench.legs_enabled(_args(workload="loftr", legs="nn"))


def test_leg_table_covers_every_other_baseline_config():
    # configs[0], [1], [3], [4] (+ its bf16-class arithmetic), the reference's own operating points of configs[2] (default adaptive conf; one pair
    # per call through the plugin seam) and the f-row matchers
    assert set(bench.LEGS) == {"splg_adaptive", "splg_b1_seam", "nn", "superpoint", "loftr_1024", "dust3r_512", "dust3r_512_fp16", "eloftr_640x480", "mast3r_512",
                               "superglue"}  # fmt: skip
    assert {w for w, _ in bench.LEGS.values()} == {"splg", "seam", "nn", "superpoint", "loftr", "eloftr", "dust3r", "mast3r", "superglue"}
    assert bench.LEGS["dust3r_512"][1]["arith"] == "fp32" and bench.LEGS["dust3r_512_fp16"][1]["arith"] == "fp16"
    assert bench.LEGS["splg_adaptive"][1]["adaptive"] is True


def test_compact_line_fits_the_drivers_tail():
    """The ONE stdout line must fit the 8 KB tail the driver keeps, with every leg's value in it: a full-size record (long config strings, per-pair
    parity lists, verbose samples, ten legs) is reduced to well under 8 KB and keeps the contract's keys."""
    import json

    leg = {"metric": "m" * 80, "value": 123.456, "unit": "pairs/s", "n_gpus": 1, "steps": 10, "warmup": 3, "ms_per_step": 12.5, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32 via 3xf16 split MFMA, f32 accumulate" + "x" * 200, "data": "synthetic", "config": {"workload": "w" * 600, "pairs_per_step_per_gpu": 16},
           "roofline": {"kernel": "k" * 200, "bound": "mfma", "achieved": 321.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.1284, "traffic": 1.07e9, "traffic_source": "t" * 300,
                        "note": "n" * 300}, "cpu_baseline": {"value": 0.2, "unit": "pairs/s", "cores": 32, "kind": "port", "sample": "s" * 500, "table": {str(i): i for i in range(20)}},
           "parity": {"status": "ok", "checked": "c" * 300, "max_score_error": 2.9e-5, "per_pair": [{"pair": i, "x": "y" * 50} for i in range(8)]}, "status": "ok", "leg_wall_s": 12.3}  # fmt: skip
    full = {**leg, "kernel_time_ms_per_step": {"attention": 26.9, "conv3x3": 17.4, "gemm": 18.9}, "workloads": {f"leg{i}": dict(leg) for i in range(10)}}
    full["workloads"]["broken"] = {"status": "failed", "error": "RuntimeError: " + "e" * 500, "traceback": "t" * 1500}
    c = bench.compact_line(full)
    text = json.dumps(c)
    assert len(text) < 7000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline"):  # fmt: skip
        assert k in c, k
    assert c["roofline"]["frac"] == 0.1284 and c["cpu_baseline"]["cores"] == 32 and c["parity"]["status"] == "ok"
    assert len(c["legs"]) == 11 and c["legs"]["leg3"]["value"] == 123.456 and c["legs"]["leg3"]["frac"] == 0.1284 and c["legs"]["leg3"]["parity"] == "ok"
    assert c["legs"]["leg3"]["cpu"] == 0.2 and c["legs"]["broken"]["status"] == "failed"


def test_ensure_built_is_a_no_op_on_a_fresh_library_and_guard_on_one_rank():
    from imcui_hip import build as b

    bench.ensure_built()
    assert not b.needs_build() and os.path.exists(b.LIB_PATH)
    assert bench.ranks_agree(True, 1, torch.device("cpu")) and not bench.ranks_agree(False, 1, torch.device("cpu"))
