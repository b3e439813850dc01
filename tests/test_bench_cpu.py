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
    assert set(bench.LEGS) == {"nn", "superpoint", "loftr_1024", "dust3r_512", "dust3r_512_fp16"}
    assert {w for w, _ in bench.LEGS.values()} == {"nn", "superpoint", "loftr", "dust3r"}
    assert bench.LEGS["dust3r_512"][1]["arith"] == "fp32" and bench.LEGS["dust3r_512_fp16"][1]["arith"] == "fp16"


def test_ensure_built_is_a_no_op_on_a_fresh_library_and_guard_on_one_rank():
    from imcui_hip import build as b

    bench.ensure_built()
    assert not b.needs_build() and os.path.exists(b.LIB_PATH)
    assert bench.ranks_agree(True, 1, torch.device("cpu")) and not bench.ranks_agree(False, 1, torch.device("cpu"))
