#!/usr/bin/env python
"""bench.py -- image-pairs/s of the SuperPoint+LightGlue hot path on N MI355X (BASELINE.json).

A step = one pass of the hot path over one batch of synthetic 640x480 pairs already resident in
HBM: SuperPoint on 2B images (<= 2048 key-points, nms 3, thr 0.005) then LightGlue on the B
pairs with early stopping and pruning DISABLED (depth = width = -1: all 9 layers on all points,
the fixed-work worst case of SURVEY.md section 8d: 334 GF/pair -- nothing is skipped).  Weights are
seeded random tensors of the real architecture (no checkpoints offline).  One process per GPU;
pairs shard across ranks with no data-path dependency; each step ends with the RCCL all-gather of
the fixed-stride match table.

Prints ONE JSON line (rank 0).  `roofline` = attention kernel (the dominant kernel, 135 of 334
GF/pair), algorithmic flops / HIP-event time measured live in the timed region; `cpu_baseline` =
the torch-CPU oracle on a bounded sample of the same workload on this host's cores.

The plain invocation (what the driver runs) then times short legs of the OTHER BASELINE.json
configs in the same process -- configs[0] mutual-NN, configs[1] SuperPoint, configs[3] LoFTR
1024x1024, configs[4] DUSt3R 512x512 in both arithmetics -- and attaches their lines (value,
ms_per_step, roofline, cpu_baseline, parity) to the one JSON line as "workloads" (`--no-legs`:
headline only; `--workload X`: that workload's line alone).  The library is built under a file
lock first when it is missing or stale, so a clean checkout and an N-rank launch work.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "image-matching-webui_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# Hardware check of the N > 1 code path on a ONE-GPU box (tests/test_gpu_rccl_single_rank.py): under `torch.distributed.run --nproc-per-node 1`
# with IMCUI_BENCH_DIST1=1 a single rank initialises RCCL and runs every barrier / all-gather / max-over-ranks reduction the multi-GPU launch
# runs.  Not set by the driver: a plain `python bench.py` (no WORLD_SIZE) never takes this path.
FORCE_DIST = os.environ.get("IMCUI_BENCH_DIST1") == "1" and os.environ.get("WORLD_SIZE") == "1"


def dist_on(world: int) -> bool:
    """True when this process is part of a torch.distributed job (N > 1, or the forced single-rank check above)."""
    return world > 1 or FORCE_DIST

H, W, MAXK = 480, 640, 2048
SP_GF_PER_IMAGE = 52.10
LG_GF_PER_LAYER_PAIR = 25.23     # @N=M=2048, shared cross similarity
ATTN_GF_PER_LAYER_PAIR = 15.03   # QK^T + PV, self (8.59) + cross with shared sim (6.44)
PEAK_F32_MFMA_TF = 157.3         # MI355X_MICROARCH.md: dense f32 MFMA peak
PEAK_F16_MFMA_TF = 2500.0        # dense f16/bf16 MFMA peak (split mode executes 3 f16 MFMAs per product)
SUSTAINED_F16_MFMA_TF = 1800.0   # tools/clock_lab.hip on this chip: register-resident MFMA loop, random operands,
                                 # power-limited to ~1.82 GHz (2200 TF/s at 2.18 GHz with zero operands)


def cpu_baseline(max_seconds: float = 28.0, min_pairs: int = 10, max_pairs: int = 12):
    """Oracle (= restated reference CPU path) timed on this host's cores, same workload.

    Leg 1 is the reference's own operating point -- one pair per call (imcui/hloc/match_features.py:172-174 never
    batches): median over >= 10 pairs after 2 warm-ups.  Leg 2 batches the extractor over 8 images (SURVEY.md section 8d
    asks for it "for fairness"; LightGlue key-point sets are ragged, the oracle matches them pair by pair).  Bounded to
    about half a minute of CPU work."""
    from imcui_hip.synth import make_pair
    from oracle.lightglue import LightGlueOracle
    from oracle.superpoint import SuperPointOracle
    from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict

    sp = SuperPointOracle(superpoint_state_dict(0))
    lg = LightGlueOracle(lightglue_state_dict(0), dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.1))
    spc = dict(nms_radius=3, max_keypoints=MAXK, keypoint_threshold=0.005, remove_borders=4)

    def match(i0, i1, f0, f1, b0=0, b1=0):
        lg({"image0": i0, "image1": i1, "keypoints0": f0["keypoints"][b0][None], "keypoints1": f1["keypoints"][b1][None],
            "descriptors0": f0["descriptors"][b0][None], "descriptors1": f1["descriptors"][b1][None]})  # fmt: skip

    def one(pair):
        i0, i1 = pair
        match(i0, i1, sp({"image": i0}, spc), sp({"image": i1}, spc))

    ncpu = os.cpu_count() or 1
    pairs = [make_pair(99 + i, H, W)[:2] for i in range(4)]
    torch.set_num_threads(best_cpu_threads())  # fastest of 8 / 16 / 32 / 64 threads on a whole pair, two runs each
    one(pairs[0])  # two warm-ups
    one(pairs[1])
    t_start = time.perf_counter()
    times = []
    while len(times) < max_pairs and (len(times) < min_pairs or (time.perf_counter() - t_start) < 0.6 * max_seconds):
        t0 = time.perf_counter()
        one(pairs[len(times) % 4])
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > max_seconds:
            break
    times.sort()
    med = times[len(times) // 2]
    # leg 2: the extractor on a batch of 8 images (4 pairs), matcher pair by pair
    i0 = torch.cat([p[0] for p in pairs], 0)
    i1 = torch.cat([p[1] for p in pairs], 0)
    t0 = time.perf_counter()
    f = sp({"image": torch.cat([i0, i1], 0)}, spc)
    for b in range(4):
        match(i0[b : b + 1], i1[b : b + 1], f, f, b, 4 + b)
    b8 = 4 / (time.perf_counter() - t0)
    return {"value": 1.0 / med, "unit": "pairs/s", "cores": torch.get_num_threads(), "host_cpus": ncpu, "kind": "port",
            "batched_extractor_value": b8,
            "sample": f"median of {len(times)} synthetic 640x480 pairs after 2 warm-ups, one pair per call (the reference never batches), fp32, "
                      f"SuperPoint(2048 kpts)+LightGlue(9 layers, no early exit), torch {torch.__version__} CPU, {torch.get_num_threads()} of {ncpu} host "
                      f"CPUs (fastest of 8/16/32/64 threads, whole pair, best of two runs each: {_BEST_THREADS[1]} s); batched_extractor_value = same with SuperPoint on a batch of 8 images",
            "thread_selection_seconds_per_pair": _BEST_THREADS[1]}  # fmt: skip


def cpu_baseline_dense(eloftr: bool, Hh: int, Ww: int, sd: dict, img0, img1, max_seconds: float = 30.0):
    """The dense matcher's oracle (= restated reference CPU path, one pair per call as `match_dense` runs it) timed on this host's
    cores on the bench pair: one warm-up, then the median of up to 3 pairs within `max_seconds`."""
    from oracle.eloftr import ELoFTROracle
    from oracle.loftr import LoFTROracle

    ncpu = os.cpu_count() or 1
    torch.set_num_threads(min(ncpu, 32))  # the thread count the sparse baseline settles on for these convolution sizes
    ora = (ELoFTROracle if eloftr else LoFTROracle)(sd, {"match_threshold": 0.2, "max_keypoints": 2000})
    data = {"image0": img0[:1].cpu(), "image1": img1[:1].cpu()}
    t0 = time.perf_counter()
    ref = ora(data)
    warm = time.perf_counter() - t0
    cpu_baseline_dense.last_ref = ref  # the warm-up run doubles as the parity reference of pair 0 (bench_loftr)
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 and (not times or time.perf_counter() - t_start + warm < max_seconds):
        t0 = time.perf_counter()
        ora(data)
        times.append(time.perf_counter() - t0)
    times.sort()
    return {"value": 1.0 / times[len(times) // 2], "unit": "pairs/s", "cores": torch.get_num_threads(), "host_cpus": ncpu, "kind": "port",
            "sample": f"median of {len(times)} synthetic {Ww}x{Hh} pair(s) after 1 warm-up, one pair per call, fp32, "
                      f"{'EfficientLoFTR' if eloftr else 'LoFTR'} oracle (torch {torch.__version__} CPU, {torch.get_num_threads()} of {ncpu} host CPUs)"}  # fmt: skip


def parity_splg(pipe, img0, img1, dc, wc, which=(0,), keypoints_only=()) -> dict:
    """"parity checks run with every benchmark".  After the timed region, the pairs `which` of the bench batch go
    through the CPU oracle (the restated reference path) one by one and are compared with what the HIP pipeline returns for them:
    key-point sets equal or every difference an audited round-off tie, the matcher run on the HIP key-points gives the same stop layer
    and matches (or audited ties of the oracle's own log-assignment) and scores within 1e-4.  Raises on a violation; the returned
    record (totals + the worst pair's tie counts) goes into the JSON line so the number printed is bound to a checked output."""
    from oracle.audit import assert_matches_equal_or_tied, audit_keypoint_differences
    from oracle.lightglue import LightGlueOracle
    from oracle.superpoint import SuperPointOracle
    from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict

    t0 = time.perf_counter()
    torch.set_num_threads(best_cpu_threads())
    spc = dict(nms_radius=3, max_keypoints=MAXK, keypoint_threshold=0.005, remove_borders=4)
    sp = SuperPointOracle(superpoint_state_dict(0))
    lg = LightGlueOracle(lightglue_state_dict(0), dict(depth_confidence=dc, width_confidence=wc, filter_threshold=0.1))
    per_pair = []
    for pi in which:
        i0, i1 = img0[pi : pi + 1], img1[pi : pi + 1]
        f = pipe.extractor.forward_batched(torch.cat([i0, i1]), want_score_map=True)
        out = pipe(i0, i1)
        torch.cuda.synchronize()
        n0, n1 = int(out["num_keypoints0"][0]), int(out["num_keypoints1"][0])
        ties = 0
        for b, (img, k, n) in enumerate(((i0, out["keypoints0"], n0), (i1, out["keypoints1"], n1))):
            ref = sp({"image": img.cpu()}, spc, return_intermediates=True)
            kp, kr = k[0, :n].cpu(), ref["keypoints"][0]
            flat_h, flat_r = (kp[:, 1] * W + kp[:, 0]).long(), (kr[:, 1] * W + kr[:, 0]).long()
            ties += audit_keypoint_differences(flat_h, flat_r, f["score_map"][b].cpu(), ref["_dense_scores"][0], spc, tag=f"bench pair {pi} image {b}")
            if len(set(flat_h.tolist()) & set(flat_r.tolist())) < 0.99 * len(flat_r):
                raise AssertionError(f"bench parity: pair {pi} image {b}: key-point sets differ ({n} vs {len(flat_r)})")
        ref = lg({"image0": i0.cpu(), "image1": i1.cpu(), "keypoints0": out["keypoints0"][0, :n0].cpu()[None], "keypoints1": out["keypoints1"][0, :n1].cpu()[None],
                  "descriptors0": out["descriptors0"][0, :n0].cpu().t()[None], "descriptors1": out["descriptors1"][0, :n1].cpu().t()[None]},
                 return_intermediates=True)  # fmt: skip
        if int(out["stop"][0]) != ref["stop"]:
            raise AssertionError(f"bench parity: pair {pi}: stop layer {int(out['stop'][0])} vs {ref['stop']}")
        m_h = out["matches0"][0, :n0].cpu()
        mt = assert_matches_equal_or_tied(m_h, ref["_log_assignment"][0], ref["matches0"][0], 0.1, tag=f"bench pair {pi}", ind0=ref.get("_ind0"), ind1=ref.get("_ind1"))
        same = m_h.long() == ref["matches0"][0]
        err = (out["matching_scores0"][0, :n0].cpu() - ref["matching_scores0"][0]).abs()[same].max().item()
        if not err < 1e-4:
            raise AssertionError(f"bench parity: pair {pi}: matching score error {err:.2e}")
        per_pair.append({"pair": pi, "keypoints": [n0, n1], "keypoint_ties_audited": ties, "matches": int((ref["matches0"] > -1).sum()), "match_ties_audited": mt,
                         "max_score_error": err, "stop_layer": ref["stop"]})  # fmt: skip
    # the extractor alone on further pairs of the batch (`keypoints_only`): the key-point audit is cheap, so it covers every DISTINCT scene
    # of the batch and the JSON carries the worst tie count seen over all of them
    extra_ties = []
    for pi in keypoints_only:
        ims = torch.cat([img0[pi : pi + 1], img1[pi : pi + 1]])
        f = pipe.extractor.forward_batched(ims, want_score_map=True)
        torch.cuda.synchronize()
        t = 0
        for b in range(2):
            ref = sp({"image": ims[b : b + 1].cpu()}, spc, return_intermediates=True)
            n = int(f["num_keypoints"][b])
            kp, kr = f["keypoints"][b, :n].cpu(), ref["keypoints"][0]
            flat_h, flat_r = (kp[:, 1] * W + kp[:, 0]).long(), (kr[:, 1] * W + kr[:, 0]).long()
            t += audit_keypoint_differences(flat_h, flat_r, f["score_map"][b].cpu(), ref["_dense_scores"][0], spc, tag=f"bench pair {pi} image {b}")
            if len(set(flat_h.tolist()) & set(flat_r.tolist())) < 0.99 * len(flat_r):
                raise AssertionError(f"bench parity: pair {pi} image {b}: key-point sets differ ({n} vs {len(flat_r)})")
        extra_ties.append(t)
    kp_ties_all = [p["keypoint_ties_audited"] for p in per_pair] + extra_ties
    return {"status": "ok", "checked": f"pairs {list(which)} of the bench batch vs the CPU oracle, after the timed region", "pairs_checked": len(per_pair),
            "score_bar": 1e-4, "weights": "synth_weights seed 0, the shaped ('damped') LightGlue set",
            "out_of_bar_stress_case": "tests/test_gpu_lightglue.py also runs an unshaped 'random' weight set (|similarity| ~ 2000) held to 5e-4 on scores: a stress case ABOVE the 1e-4 bar, not a parity claim",
            "keypoint_audit_pairs": len(kp_ties_all), "worst_keypoint_ties_per_pair_all": max(kp_ties_all), "keypoint_ties_audited_all": sum(kp_ties_all),
            "keypoints": per_pair[0]["keypoints"], "keypoint_ties_audited": sum(p["keypoint_ties_audited"] for p in per_pair),
            "worst_keypoint_ties_per_pair": max(p["keypoint_ties_audited"] for p in per_pair),
            "matches": per_pair[0]["matches"], "match_ties_audited": sum(p["match_ties_audited"] for p in per_pair),
            "worst_match_ties_per_pair": max(p["match_ties_audited"] for p in per_pair),
            "max_score_error": max(p["max_score_error"] for p in per_pair), "stop_layer": per_pair[0]["stop_layer"], "per_pair": per_pair,
            "seconds": round(time.perf_counter() - t0, 2)}  # fmt: skip


def parity_superpoint(model, img, which) -> dict:
    """Images `which` of the bench batch through the CPU oracle after the timed region: key-point index sets equal or every difference
    an audited round-off tie (oracle/audit.py), scores of the common points within 1e-4, descriptors of the common points within 1e-4."""
    from oracle.audit import audit_keypoint_differences
    from oracle.superpoint import SuperPointOracle
    from imcui_hip.synth_weights import superpoint_state_dict

    t0 = time.perf_counter()
    torch.set_num_threads(best_cpu_threads())
    spc = dict(nms_radius=3, max_keypoints=MAXK, keypoint_threshold=0.005, remove_borders=4)
    sp = SuperPointOracle(superpoint_state_dict(0))
    sub = img[which]
    f = model.forward_batched(sub, want_score_map=True)
    torch.cuda.synchronize()
    ties, worst_ties, derr, serr, counts = 0, 0, 0.0, 0.0, []
    for b in range(len(which)):
        ref = sp({"image": sub[b : b + 1].cpu()}, spc, return_intermediates=True)
        n = int(f["num_keypoints"][b])
        kp, kr = f["keypoints"][b, :n].cpu(), ref["keypoints"][0]
        flat_h, flat_r = (kp[:, 1] * W + kp[:, 0]).long(), (kr[:, 1] * W + kr[:, 0]).long()
        t = audit_keypoint_differences(flat_h, flat_r, f["score_map"][b].cpu(), ref["_dense_scores"][0], spc, tag=f"bench image {which[b]}")
        ties, worst_ties = ties + t, max(worst_ties, t)
        pos_r = {int(v): i for i, v in enumerate(flat_r.tolist())}
        ih = [i for i, v in enumerate(flat_h.tolist()) if v in pos_r]
        ir = [pos_r[int(flat_h[i])] for i in ih]
        if len(ih) < 0.99 * len(flat_r):
            raise AssertionError(f"bench parity (superpoint): image {which[b]}: key-point sets differ ({n} vs {len(flat_r)}, {len(ih)} common)")
        serr = max(serr, (f["scores"][b, :n].cpu()[ih] - ref["scores"][0][ir]).abs().max().item())
        derr = max(derr, (f["descriptors"][b, :n].cpu()[ih] - ref["descriptors"][0].t()[ir]).abs().max().item())
        counts.append(n)
    if not (serr < 1e-4 and derr < 1e-4):
        raise AssertionError(f"bench parity (superpoint): score error {serr:.2e}, descriptor error {derr:.2e}")
    return {"status": "ok", "checked": f"images {which} of the bench batch vs the CPU oracle, after the timed region", "keypoints": counts, "keypoint_ties_audited": ties,
            "worst_keypoint_ties_per_image": worst_ties, "max_score_error": serr, "max_descriptor_error": derr, "seconds": round(time.perf_counter() - t0, 2)}  # fmt: skip


def cpu_baseline_superpoint(imgs, max_seconds: float = 12.0):
    """The SuperPoint oracle (= restated reference CPU path, one image per call as extract_features.py:203-209 runs it) on this host."""
    from oracle.superpoint import SuperPointOracle
    from imcui_hip.synth_weights import superpoint_state_dict

    torch.set_num_threads(best_cpu_threads())
    spc = dict(nms_radius=3, max_keypoints=MAXK, keypoint_threshold=0.005, remove_borders=4)
    sp = SuperPointOracle(superpoint_state_dict(0))
    sp({"image": imgs[:1]}, spc)
    times, t_start = [], time.perf_counter()
    while len(times) < 24 and (len(times) < 6 or time.perf_counter() - t_start < max_seconds):
        t0 = time.perf_counter()
        sp({"image": imgs[len(times) % len(imgs)][None]}, spc)
        times.append(time.perf_counter() - t0)
    times.sort()
    ncpu = os.cpu_count() or 1
    return {"value": 1.0 / times[len(times) // 2], "unit": "images/s", "cores": torch.get_num_threads(), "host_cpus": ncpu, "kind": "port",
            "sample": f"median of {len(times)} synthetic 640x480 images after 1 warm-up, one image per call, fp32, SuperPoint oracle (2048 kpts), torch {torch.__version__} CPU, "
                      f"{torch.get_num_threads()} of {ncpu} host CPUs"}  # fmt: skip


_BEST_THREADS: list = []


def best_cpu_threads() -> int:
    """The intra-op thread count that runs the oracle fastest on this host (oversubscribing a many-core box is catastrophically slow),
    chosen ONCE per process on a whole SuperPoint + LightGlue pair: every candidate of 8 / 16 / 32 / 64 runs the pair twice (the first
    run also warms the allocator and the thread pool at that width) and is judged by its faster run -- round 3 timed one extractor call
    per candidate and two boxes of the pool settled on different counts."""
    if _BEST_THREADS:
        return _BEST_THREADS[0]
    from imcui_hip.synth import make_pair
    from oracle.lightglue import LightGlueOracle
    from oracle.superpoint import SuperPointOracle
    from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict

    ncpu = os.cpu_count() or 1
    sp = SuperPointOracle(superpoint_state_dict(0))
    lg = LightGlueOracle(lightglue_state_dict(0), dict(depth_confidence=-1, width_confidence=-1, filter_threshold=0.1))
    spc = dict(nms_radius=3, max_keypoints=MAXK, keypoint_threshold=0.005, remove_borders=4)
    i0, i1 = make_pair(98, H, W)[:2]

    def one():
        f0, f1 = sp({"image": i0}, spc), sp({"image": i1}, spc)
        lg({"image0": i0, "image1": i1, "keypoints0": f0["keypoints"][0][None], "keypoints1": f1["keypoints"][0][None],
            "descriptors0": f0["descriptors"][0][None], "descriptors1": f1["descriptors"][0][None]})  # fmt: skip

    best_t, best_dt, table = 1, float("inf"), {}
    for t in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(t)
        d = float("inf")
        for _ in range(2):
            t0 = time.perf_counter()
            one()
            d = min(d, time.perf_counter() - t0)
        table[t] = round(d, 3)
        if d < best_dt:
            best_t, best_dt = t, d
    _BEST_THREADS.extend([best_t, table])
    torch.set_num_threads(best_t)
    return best_t


def bench_nn(args, dev, rank, world):
    """configs[0] (the plumbing config): mutual nearest-neighbour matcher on SIFT-like descriptors -- 5000 x 128-d RootSIFT-style rows
    per image, the matcher zoo's `NN-mutual` conf (imcui/hloc/matchers/nearest_neighbor.py:38-66: ratio test
    off, distance threshold off, mutual check).  One step = B independent pairs through imcui_hip_mutual_nn_dn.  The [5000 x 5000] similarity
    tiles are reduced inside the persistent kernel of csrc/simred.hip and never stored, in either arithmetic -- priced against the MFMA peak."""
    from imcui_hip import backend
    from imcui_hip.hloc.matchers.nearest_neighbor import NearestNeighbor

    failed = None
    try:  # set-up and warm-up: a failure on one rank must not leave the others in the timed loop's barrier
        N, D, B = 5000, 128, args.batch
        g = torch.Generator().manual_seed(4321 + rank)
        d0 = torch.rand(B, D, N, generator=g).sqrt()
        d0 = d0 / d0.norm(dim=1, keepdim=True)  # RootSIFT: non-negative, unit L2 norm
        perm = torch.stack([torch.randperm(N, generator=g) for _ in range(B)])
        d1 = torch.gather(d0, 2, perm[:, None, :].expand(B, D, N)) + 0.05 * torch.randn(B, D, N, generator=g)
        d1 = d1 / d1.norm(dim=1, keepdim=True)
        d0, d1 = d0.to(dev), d1.to(dev)
        model = NearestNeighbor({"ratio_threshold": None, "distance_threshold": None, "do_mutual_check": True}).eval().to(dev)

        def step():
            return model({"descriptors0": d0, "descriptors1": d1})  # B pairs in one C-ABI call (the reference: one pair per call)

        for _ in range(args.warmup):
            out = step()
    except Exception as e:  # noqa: BLE001
        if world == 1:
            raise
        failed = e
    if not ranks_agree(failed is None, world, dev):
        raise LegSkipped(f"set-up failed on a rank: {failed!r}")
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    backend.profile_enable(dev, True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        out = step()
    ev1.record()
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    gemm_ms, gemm_n = backend.profile_read(dev, "gemm")
    backend.profile_enable(dev, False)
    if dist_on(world):
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        split = args.precision == 1
        # Round 5, either arithmetic: both descriptor sets are packed once into MFMA fragments and ONE persistent launch (csrc/simred.hip) reduces
        # the similarity tiles to (best, index, second best) on the spot -- rows in registers across the column tiles, columns per 128-row block;
        # the matrix is never stored.  HBM traffic = the descriptors (+ their packed copies) and 12 B per column and row block: matrix-pipe bound.
        gf = 2.0 * N * N * D / 1e9  # algorithmic GFLOP per pair (the similarity products)
        ach = gf * 1e9 * B * args.steps / (gemm_ms * 1e-3) / 1e12 if gemm_ms else 0.0
        peak = PEAK_F16_MFMA_TF if split else PEAK_F32_MFMA_TF
        roof = {"kernel": "simred_kernel<K/16 = 8, mode NN> (persistent similarity-and-reduce: A rows in registers, B by LDS-DMA through a four-slot ring)", "bound": "mfma",
                "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None, "executed_tflops": (3.0 if split else 1.0) * ach,
                "algorithmic_gflop_per_pair": gf, "launches_per_step": gemm_n / max(args.steps, 1), "kernel_ms_per_step": gemm_ms / max(args.steps, 1),
                "note": "achieved = algorithmic TFLOP of the similarity products / kernel time (HIP events inside the timed loop); K = 128 leaves four "
                        "two-k-step stages between two tile epilogues (per row 32 values / lane, per column a 128 x 128 tile parked in LDS)"}  # fmt: skip
        line = {
            "metric": "image-pairs/sec mutual nearest-neighbour matcher (5000 x 128-d descriptors)", "value": world * B * args.steps / dt, "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 via 3xf16 split MFMA, f32 accumulate" if args.precision == 1 else "f32", "data": "synthetic",
            "config": {"workload": "configs[0] (matcher half): NN-mutual on 5000 x 128-d RootSIFT-like descriptors per image, resident in HBM",
                       "pairs_per_step_per_gpu": B, "parallelism": f"pairs sharded over {world} rank(s) (RCCL world size {dist.get_world_size() if dist_on(world) else 1}), no collective",
                       "matches_pair0": int((out["matches0"][0] > -1).sum())},
            "roofline": roof,
        }  # fmt: skip
        # parity: pair 0 against the oracle (pinned to the reference's own module by tests/golden/nn_*.npz)
        from oracle.mutual_nn import mutual_nn

        nnc = {"ratio_threshold": None, "distance_threshold": None, "do_mutual_check": True}
        ref = mutual_nn({"descriptors0": d0[:1].cpu(), "descriptors1": d1[:1].cpu()}, nnc)
        ok = torch.equal(out["matches0"][:1].cpu().long(), ref["matches0"].long()) and (out["matching_scores0"][:1].cpu() - ref["matching_scores0"]).abs().max().item() < 1e-4
        if not ok:
            raise AssertionError("bench parity (nn): matches / scores differ from the oracle")
        line["parity"] = {"status": "ok", "checked": "pair 0 vs oracle/mutual_nn.py (golden-pinned), matches bit-exact, scores within 1e-4"}
        if not args.no_cpu_baseline and world == 1:
            torch.set_num_threads(min(os.cpu_count() or 1, 32))
            c0, c1 = d0.cpu(), d1.cpu()
            mutual_nn({"descriptors0": c0[:1], "descriptors1": c1[:1]}, nnc)
            t0 = time.perf_counter()
            n = 0
            while time.perf_counter() - t0 < 10.0 and n < 64:
                mutual_nn({"descriptors0": c0[n % B : n % B + 1], "descriptors1": c1[n % B : n % B + 1]}, nnc)
                n += 1
            line["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "pairs/s", "cores": torch.get_num_threads(), "host_cpus": os.cpu_count(), "kind": "port",
                                    "sample": f"{n} pairs of the same workload through oracle/mutual_nn.py (torch CPU), one pair per call"}  # fmt: skip
        return line
    return None


def bench_superpoint(args, dev, rank, world):
    """configs[1]: SuperPoint extractor (max 2048 key-points) on 640x480 batches; images/s, weak scaling."""
    from imcui_hip import backend
    from imcui_hip.hloc.extractors.superpoint import SuperPoint
    from imcui_hip.synth import make_pair_batch
    from imcui_hip.synth_weights import superpoint_state_dict  # seeded weights only

    failed = None
    try:  # set-up and warm-up: a failure on one rank must not leave the others in the timed loop's barrier
        B = 2 * args.batch  # images per step per GPU (the pairs workload extracts 2 images per pair)
        model = SuperPoint({"nms_radius": 3, "max_keypoints": MAXK, "keypoint_threshold": 0.005, "remove_borders": 4,
                            "state_dict": superpoint_state_dict(0)}).eval().to(dev)  # fmt: skip
        img0, img1, _ = make_pair_batch(1234 + rank, B // 2, H, W, distinct=min(B // 2, 4))
        img = torch.cat([img0, img1], 0).to(dev)
        for _ in range(args.warmup):
            out = model.forward_batched(img)
    except Exception as e:  # noqa: BLE001
        if world == 1:
            raise
        failed = e
    if not ranks_agree(failed is None, world, dev):
        raise LegSkipped(f"set-up failed on a rank: {failed!r}")
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    backend.profile_enable(dev, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model.forward_batched(img)
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    dt = time.perf_counter() - t0
    conv_ms, conv_n = backend.profile_read(dev, "conv3x3")
    backend.profile_enable(dev, False)
    if dist_on(world):
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        split = args.precision == 1
        peak = PEAK_F16_MFMA_TF if split else PEAK_F32_MFMA_TF
        conv_gf = 45.65 - 0.35 + 2.83 + 2.83  # 3x3 layers on the MFMA conv kernel: encoder minus conv1a, convPa, convDa
        achieved = conv_gf * 1e9 * B * args.steps / (conv_ms * 1e-3) / 1e12 if conv_ms else 0.0
        line = {
            "metric": "images/sec @640x480 SuperPoint extractor", "value": world * B * args.steps / dt, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 via 3xf16 split MFMA, f32 accumulate" if split else "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: SuperPoint (max 2048 kpts, nms 3, thr 0.005) on synthetic 640x480 images resident in HBM",
                       "images_per_step_per_gpu": B, "mean_keypoints": float(out["num_keypoints"].float().mean()),
                       "weights": "seeded random (imcui_hip/synth_weights.py), real architecture"},
            "roofline": {"kernel": "conv3x3_split_kernel (implicit-GEMM 3x3 convolutions)" if split else "conv3x3_kernel", "bound": "mfma",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                         "executed_tflops": achieved * (3.0 if split else 1.0),
                         "executed_frac_of_sustained_peak": achieved * 3.0 / SUSTAINED_F16_MFMA_TF if split else None,
                         "conv_ms_per_step": conv_ms / args.steps, "launches_per_step": conv_n / args.steps,
                         "note": "achieved = algorithmic TFLOP of the 3x3 layers / summed conv kernel time (HIP events)"},
            "algorithmic_tflops_end_to_end": SP_GF_PER_IMAGE * 1e9 * B / (dt / args.steps) / 1e12,
        }  # fmt: skip
        if not args.no_parity:
            line["parity"] = parity_superpoint(model, img, [0, 1, B // 2, B // 2 + 1])
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline_superpoint(img[:4].cpu())
        return line
    return None


def bench_loftr(args, dev, rank, world):
    """configs[3]: LoFTR dense matcher (coarse 1/8 + fine) on synthetic pairs; pairs/s, weak scaling."""
    from imcui_hip import backend
    from imcui_hip.distributed import TableGather
    from imcui_hip.hloc.matchers.loftr import LoFTR
    from imcui_hip.synth import make_pair
    from imcui_hip.synth_weights import loftr_state_dict  # seeded weights only

    failed = None
    try:  # set-up and warm-up: a failure on one rank must not leave the others in the timed loop's barrier
        eloftr = args.workload == "eloftr"  # matcher zoo entry `eloftr` (configs/matchers.py:288-306: 640x480, 2000 matches kept)
        Hh, Ww = args.size if args.size else ((480, 640) if eloftr else (1024, 1024))
        B = args.batch
        if eloftr:
            from imcui_hip.hloc.matchers.eloftr import ELoFTR
            from imcui_hip.synth_weights import eloftr_state_dict

            sd = eloftr_state_dict(0)
            model = ELoFTR({"match_threshold": 0.2, "max_keypoints": 2000, "state_dict": sd}).eval().to(dev)
        else:
            sd = loftr_state_dict(0)
            # (lab: IMCUI_BENCH_LOFTR_THR lowers the coarse threshold to raise the match count of the synthetic pair -- the fine level's cost
            # per match, profiles/r06_lab_loftr_fine*.txt; the leg itself always runs the zoo's 0.2)
            model = LoFTR({"match_threshold": float(os.environ.get("IMCUI_BENCH_LOFTR_THR", "0.2")), "max_keypoints": 2000, "state_dict": sd}).eval().to(dev)
        base, _, _ = make_pair(77 + rank, Hh + 16, Ww + 16, n_blobs=Hh * Ww // 150)
        img0 = base[..., 0:Hh, 0:Ww].contiguous().repeat(B, 1, 1, 1).to(dev)
        img1 = base[..., 8 : Hh + 8, 16 : Ww + 16].contiguous().repeat(B, 1, 1, 1).to(dev)
        cap = B * (Hh // 8) * (Ww // 8)
        gather = TableGather(world, cap + 1, 6, torch.float32, dev, force=FORCE_DIST)

        def step():
            out = model.forward_batched(img0, img1)
            if dist_on(world):  # fixed-capacity match table of this rank's pairs: rows (x0, y0, x1, y1, conf, pair), last row = count
                rows = torch.cat([out["keypoints0"], out["keypoints1"], out["confidence"][:, None], out["batch_indexes"].float()[:, None]], 1)
                gather(torch.cat([rows, out["num_matches"].float().expand(1, 6)], 0))
            return out

        for _ in range(args.warmup):
            out = step()
    except Exception as e:  # noqa: BLE001
        if world == 1:
            raise
        failed = e
    if not ranks_agree(failed is None, world, dev):
        raise LegSkipped(f"set-up failed on a rank: {failed!r}")
    gather.finish()
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    backend.profile_enable(dev, True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    gather.finish()
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    dt = time.perf_counter() - t0
    gemm_ms, gemm_n = backend.profile_read(dev, "gemm")
    conv_ms, conv_n = backend.profile_read(dev, "conv3x3")  # the 3x3 stride-1 convolutions run on the patch-staging kernel
    gemm_ms, gemm_n = gemm_ms + conv_ms, gemm_n + conv_n
    backend.profile_enable(dev, False)
    if dist_on(world):
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        import glob

        # HBM bytes of the GEMM-class kernels per launch from the committed PMC passes of the same workload (1024^2 only)
        traffic, traffic_source = None, None
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_eloftr.json" if eloftr else "r*_pmc_traffic_loftr.json")))
        if cands and (Hh, Ww) == ((480, 640) if eloftr else (1024, 1024)) and gemm_n:
            with open(cands[-1]) as fh:
                tj = json.load(fh)
            traffic = tj["traffic_bytes_per_step_gemm_kernels"] * B / tj["pairs_per_step"] / (gemm_n / args.steps)
            traffic_source = f"NOT measured in this run: {os.path.relpath(cands[-1], ROOT)} (committed rocprofv3 --pmc passes of this workload), per launch of the matrix class"
        # algorithmic work: 2.55 TF / pair at 1024^2, scaled by area (coarse sim by area^2)
        area = Hh * Ww / (1024.0 * 1024.0)
        tf_pair = (2.03 + 0.35 + 0.03) * area + 0.14 * area * area
        if eloftr:  # per 640x480 pair: 2 x (53.8 backbone + 35.5 fine fusion) + 30.2 transformer MLPs GMAC, 5.9 GMAC coarse sim
            a2 = Hh * Ww / (640.0 * 480.0)
            tf_pair = 0.4176 * a2 + 0.0118 * a2 * a2
        fine_stage = None
        if not eloftr:
            # The last FPN stage (layer1_outconv + layer1_outconv2 at 1/2 resolution, 0.626 TF of the 2.55 TF per 1024^2 pair) is evaluated on the 5x5
            # windows of the matches when that is cheaper (option loftr_fine_sparse, csrc/loftr.hip): the roofline line then counts the work
            # that was EXECUTED (98.5 MFLOP per match instead of the dense maps), not the nominal dense figure.
            mode, nmatch_all = model._impl.last_fine_mode(dev)
            dense_tf = 2 * (Hh // 2) * (Ww // 2) * 2.0 * (128 * 196 + 9 * 196 * 196 + 9 * 196 * 128) / 1e12
            win_tf = max(nmatch_all, 0) / B * 2 * 2.0 * (81 * 128 * 196 + 49 * 9 * 196 * 196 + 25 * 9 * 196 * 128) / 1e12
            fine_stage = {"mode": "windows of the matches" if mode == 1 else "dense maps", "matches_per_step": nmatch_all, "option_loftr_fine_sparse": backend.get_option(dev, "loftr_fine_sparse"),
                          "dense_tflop_per_pair": dense_tf, "executed_tflop_per_pair": win_tf if mode == 1 else dense_tf,
                          "note": "data dependent: the synthetic pair of this leg yields few matches; the dense evaluation is what `--fine-dense` (option 0) times"}  # fmt: skip
            if mode == 1:
                tf_pair += win_tf - dense_tf
                # the same step with the dense maps (option 0), timed right here: what this leg reported until round 5 and what a pair with
                # thousands of matches costs (break-even ~3000 matches per 1024^2 pair: profiles/r06_lab_loftr_fine.txt)
                with backend.option(dev, loftr_fine_sparse=0):
                    step()
                    torch.cuda.synchronize()
                    td = time.perf_counter()
                    for _ in range(max(2, args.steps // 2)):
                        step()
                    gather.finish()
                    torch.cuda.synchronize()
                    td = (time.perf_counter() - td) / max(2, args.steps // 2)
                fine_stage["dense_maps_pairs_per_s"] = B / td
                fine_stage["dense_maps_ms_per_step"] = td * 1e3
            if (Hh, Ww) == (1024, 1024) and "IMCUI_BENCH_LOFTR_THR" not in os.environ:
                # The synthetic pair with seeded weights yields 17 matches per pair at the zoo's threshold; a real outdoor pair yields thousands, and
                # both the fine transformer and the window evaluation scale with that.  The same step with the coarse threshold lowered until the
                # synthetic pair gives ~2000 matches per pair (3e-5), default routing and dense maps: the load a user is more likely to see.
                m2 = LoFTR({"match_threshold": 3e-5, "max_keypoints": 2000, "state_dict": sd}).eval().to(dev)
                rec = {}
                for tag, opt in (("default_routing", backend.get_option(dev, "loftr_fine_sparse")), ("dense_maps", 0)):
                    with backend.option(dev, loftr_fine_sparse=opt):
                        o2 = m2.forward_batched(img0, img1)
                        torch.cuda.synchronize()
                        t2 = time.perf_counter()
                        for _ in range(3):
                            o2 = m2.forward_batched(img0, img1)
                        torch.cuda.synchronize()
                        rec[tag + "_pairs_per_s"] = B * 3 / (time.perf_counter() - t2)
                rec["matches_per_pair"] = int(o2["num_matches"][0]) / B
                rec["coarse_threshold"] = 3e-5
                fine_stage["at_realistic_match_count"] = rec
                del m2
        split = args.precision == 1
        line = {
            "metric": "image-pairs/sec EfficientLoFTR dense matcher" if eloftr else "image-pairs/sec LoFTR dense matcher", "value": world * B * args.steps / dt, "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 via 3xf16 split MFMA, f32 accumulate" if split else "f32", "data": "synthetic",
            "config": {"workload": (f"EfficientLoFTR (RepVGG 8-1 + 4 x aggregated self / cross attention + two-stage fine matching) on synthetic {Ww}x{Hh} pairs resident in HBM" if eloftr else
                                    f"configs[3]: LoFTR (ResNetFPN_8_2 + 8 coarse + 2 fine linear-attention layers) on synthetic {Ww}x{Hh} pairs resident in HBM"),
                       "pairs_per_step_per_gpu": B, "matches": int(out["num_matches"][0]), **({"fine_stage": fine_stage} if fine_stage else {}),
                       "weights": "seeded random (imcui_hip/synth_weights.py), " + ("EfficientLoFTR architecture (transformers port names)" if eloftr else "kornia LoFTR architecture")},
            "roofline": {"kernel": "conv3x3_split_kernel + gemm_split_kernel + lg_ffn_kernel (matrix class: 3x3 convolutions, strided / 1x1 convolutions as implicit GEMM, projections, fused MLPs)" if split else "gemm_kernel", "bound": "mfma",
                         "achieved": tf_pair * B * args.steps / (gemm_ms * 1e-3) if gemm_ms else 0.0, "peak": PEAK_F16_MFMA_TF if split else PEAK_F32_MFMA_TF,
                         "unit": "TFLOP/s", "frac": (tf_pair * B * args.steps / (gemm_ms * 1e-3) / (PEAK_F16_MFMA_TF if split else PEAK_F32_MFMA_TF)) if gemm_ms else 0.0,
                         "traffic": traffic, "traffic_source": traffic_source, "gemm_ms_per_step": gemm_ms / args.steps, "launches_per_step": gemm_n / args.steps,
                         "note": "achieved = algorithmic TFLOP of a pair / summed matrix-class kernel time (HIP events; conv3x3 + GEMM classes)"},
            "algorithmic_tflops_end_to_end": tf_pair * B / (dt / args.steps),
        }  # fmt: skip
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_dense(eloftr, Hh, Ww, sd, img0, img1)
            if not args.no_parity:
                # parity of pair 0: the plugin's `_forward` (wrapper semantics: top-k by confidence) against the oracle's dictionary.  Rows
                # are paired by their coarse cell (keypoints of the un-refined image are exact cell centres); near-threshold rows may
                # differ (round-off ties), so >= 99.5 % of the oracle's rows must be present, refined points within 5e-3 px, scores 1e-3
                ref = cpu_baseline_dense.last_ref
                got = model({"image0": img0[:1], "image1": img1[:1]})
                key = lambda k: {(int(round(float(x) * 4)), int(round(float(y) * 4))): i for i, (x, y) in enumerate(k.tolist())}
                gk, rk = key(got["keypoints1"].cpu()), key(ref["keypoints1"])
                common = [c for c in rk if c in gk]
                gi, ri = [gk[c] for c in common], [rk[c] for c in common]
                perr = (got["keypoints0"].cpu()[gi] - ref["keypoints0"][ri]).abs().max().item() if common else float("inf")
                serr = (got["scores"].cpu()[gi] - ref["scores"][ri]).abs().max().item() if common else float("inf")
                frac = len(common) / max(len(rk), 1)
                if not (frac >= 0.995 and perr < 5e-3 and serr < 1e-3):
                    raise AssertionError(f"bench parity (dense): {frac:.4f} of the oracle's matches found, refined-point error {perr:.2e} px, score error {serr:.2e}")
                line["parity"] = {"status": "ok", "checked": "pair 0 through the plugin's _forward vs the CPU oracle (run timed for cpu_baseline)", "oracle_matches": len(rk),
                                  "common_fraction": frac, "max_refined_point_error_px": perr, "max_score_error": serr}  # fmt: skip
        return line
    return None


_DUST3R_MODELS: dict = {}  # (kind, device) -> module, reused by consecutive legs of one process
_DUST3R_CPU: dict = {}     # (kind, H, W) -> (cpu_baseline record, oracle output of pair 0)


def dust3r_tflop_per_pair(cfg: dict, H: int, W: int) -> dict:
    """Algorithmic TFLOP of one symmetrised image pair (2 images encoded once, 2 directed pairs decoded, 4 DPT heads)."""
    E, D, ne, nd = cfg["enc_dim"], cfg["dec_dim"], cfg["enc_depth"], cfg["dec_depth"]
    T = (H // 16) * (W // 16)
    enc = 2 * (T * 2 * 768 * E + ne * (T * 2 * 12 * E * E + 4 * T * T * E))
    dec = 2 * (T * 2 * E * D) + 2 * 2 * nd * (T * 2 * 16 * D * D + 2 * 4 * T * T * D)
    px = lambda lvl: T * 4.0 ** lvl  # cells of the 1/16 grid scaled by 4^lvl: -1 = 1/32, 0 = 1/16, 1 = 1/8, 2 = 1/4, 3 = 1/2, 4 = full resolution
    head = (T * 2 * (E * 96 + 96 * 1536 + D * 192 + 192 * 768 + D * 384 + D * 768) + px(-1) * 18 * 768 * 768  # reassemble
            + 18 * 256 * (px(2) * 96 + px(1) * 192 + px(0) * 384 + px(-1) * 768)                                 # layer_rn
            + 18 * 256 * 256 * (2 * px(-1) + 4 * px(0) + 4 * px(1) + 4 * px(2))                                  # residual units
            + 2 * 256 * 256 * (px(0) + px(1) + px(2) + px(3))                                                    # out_conv
            + 18 * 256 * 128 * px(3) + 18 * 128 * 128 * px(4) + 2 * 128 * 4 * px(4))                             # head
    return {"encoder": enc / 1e12, "decoder": dec / 1e12, "heads": 4 * head / 1e12, "total": (enc + dec + 4 * head) / 1e12}


def bench_dust3r(args, dev, rank, world):
    """configs[4]: DUSt3R ViT-L pair encoder + DPT regression head on 512x512 pairs; symmetrised image pairs/s (what one call of
    imcui/hloc/matchers/duster.py processes: the directed pairs (1, 0) and (0, 1)), weak scaling, pairs sharded over the ranks,
    no collective (the outputs are dense point maps consumed by the host-side aligner of the rank that owns the pair)."""
    from imcui_hip import backend
    from imcui_hip.hloc.matchers.duster import Duster
    from imcui_hip.synth import make_pair
    from imcui_hip.synth_weights import DUST3R_CFG, dust3r_state_dict  # seeded weights only

    failed = None
    try:  # set-up and warm-up: a failure on one rank must not leave the others in the timed loop's barrier
        Hh, Ww = args.size if args.size else (512, 512)
        B = args.batch
        if args.nn_arith == "auto":
            args.nn_arith = "split" if args.precision == 1 else "fp32"
        mast = args.workload == "mast3r"  # the same network with the 'catmlp+dpt' head + the reciprocal descriptor matching of mast3r.py:68-75
        cfg = {**DUST3R_CFG, "desc_dim": 24 if mast else 0}
        # generating and packing the 578 M seeded parameters takes ~30 s of host time: the packed buffer (a pure function of the seed, the
        # same on every rank) is kept in the temp directory -- the first rank to take the file lock writes it, the others load it -- and
        # the module built from it is kept for the next leg of the same process (dust3r_512 -> dust3r_512_fp16)
        import fcntl
        import tempfile

        kind = "mast3r" if mast else "dust3r"
        sd = None
        model = _DUST3R_MODELS.get((kind, str(dev)))
        if model is None:
            cache = os.path.join(tempfile.gettempdir(), f"imcui_hip_{kind}_seed0_v{backend.lib_version()}.pt")
            with open(cache + ".lock", "w") as lk:
                fcntl.flock(lk, fcntl.LOCK_EX)
                try:
                    if not os.path.exists(cache):
                        sd = dust3r_state_dict(0, cfg)
                        packed, _ = backend.pack_dust3r(sd)
                        try:
                            torch.save(packed, cache + f".tmp{os.getpid()}")
                            os.replace(cache + f".tmp{os.getpid()}", cache)
                        except OSError:
                            pass
                    else:
                        packed = torch.load(cache)
                finally:
                    fcntl.flock(lk, fcntl.LOCK_UN)
            if mast:
                from imcui_hip.hloc.matchers.mast3r import Mast3r

                model = Mast3r({"packed": (packed, cfg)}).eval().to(dev)
            else:
                model = Duster({"packed": (packed, cfg)}).eval().to(dev)
            del packed
            _DUST3R_MODELS.clear()  # one resident copy (2.3 GB of planes) at a time
            _DUST3R_MODELS[(kind, str(dev))] = model
        if mast:
            from imcui_hip.hloc.matchers.mast3r import fast_reciprocal_nns
        base, _, _ = make_pair(91 + rank, Hh + 16, Ww + 16, n_blobs=Hh * Ww // 150)
        g = torch.Generator().manual_seed(5 + rank)
        i0 = torch.cat((base[..., 0:Hh, 0:Ww], base[..., 4 : Hh + 4, 2 : Ww + 2] * 0.8 + 0.1, torch.rand(1, 1, Hh, Ww, generator=g)), 1)
        i1 = torch.cat((base[..., 8 : Hh + 8, 16 : Ww + 16], base[..., 12 : Hh + 12, 6 : Ww + 6] * 0.8 + 0.1, torch.rand(1, 1, Hh, Ww, generator=g)), 1)
        images = torch.cat((i0, i1), 0).repeat(B, 1, 1, 1).contiguous().to(dev)  # [2B,3,H,W]: images 2b, 2b+1 form pair b
        pairs = torch.tensor([[2 * b + 1 - a, 2 * b + a] for b in range(B) for a in (0, 1)], dtype=torch.int32, device=dev)  # (1, 0), (0, 1) per image pair

        model.conf["arithmetic"] = args.arith

        nmatch = [0]

        def step():
            out = model.forward_pairs(images, pairs)
            if mast:  # per image pair: descriptors of its second directed pair (mast3r.py:61-64), reciprocal matching, 2000 kept
                n = 0
                for b in range(B):
                    k0, k1 = fast_reciprocal_nns(out["desc"][0][2 * b + 1], out["desc"][1][2 * b + 1], subsample=2, split=args.nn_arith == "split")
                    if len(k0) > 2000:
                        keep = torch.linspace(0, len(k0) - 1, 2000, device=dev).round().long()
                        k0, k1 = k0[keep], k1[keep]
                    n += len(k0)
                nmatch[0] = n
            return out

        for _ in range(args.warmup):
            out = step()
    except Exception as e:  # noqa: BLE001
        if world == 1:
            raise
        failed = e
    if not ranks_agree(failed is None, world, dev):
        raise LegSkipped(f"set-up failed on a rank: {failed!r}")
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    backend.profile_enable(dev, True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    dt = time.perf_counter() - t0
    cls_ms = {k: backend.profile_read(dev, k) for k in ("gemm", "conv3x3", "attention")}
    backend.profile_enable(dev, False)
    net_ms = None
    if mast:  # the network alone, same batch (the matching step = the rest of a step)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            model.forward_pairs(images, pairs)
        torch.cuda.synchronize()
        net_ms = (time.perf_counter() - t1) / 3 * 1e3
    if dist_on(world):
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        tf = dust3r_tflop_per_pair(cfg, Hh, Ww)
        mat_ms = sum(v[0] for v in cls_ms.values())
        mat_n = sum(v[1] for v in cls_ms.values())
        ach = tf["total"] * B * args.steps / (mat_ms * 1e-3) if mat_ms else 0.0
        # HBM bytes of the matrix-class kernels per launch from the committed PMC passes of the same workload (512x512 only)
        import glob

        traffic, traffic_source = None, None
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_dust3r.json")))
        if cands and (Hh, Ww) == (512, 512) and mat_n and args.arith == "fp32":
            with open(cands[-1]) as fh:
                tj = json.load(fh)
            traffic = tj["traffic_bytes_per_step_gemm_kernels"] * B / tj["pairs_per_step"] / (mat_n / args.steps)
            traffic_source = f"NOT measured in this run: {os.path.relpath(cands[-1], ROOT)} (committed rocprofv3 --pmc passes of this workload), per launch of the matrix class"
        line = {
            "metric": ("image-pairs/sec MASt3R pair network + reciprocal matching @512x512" if mast else "image-pairs/sec DUSt3R pair network @512x512"), "value": world * B * args.steps / dt, "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 via 3xf16 split MFMA, f32 accumulate (the reference config names bf16; this path keeps fp32-grade results)" if args.arith == "fp32" else
                      "f16 operands (one MFMA product per element pair, 11-bit mantissa), f32 accumulate -- GEMMs, convolutions AND attention (attn_split_kernel<.., 4>: hi planes, probabilities rounded to f16); bf16-class, NOT a parity mode (the reference config names bf16)"),
            "data": "synthetic",
            "config": {"workload": ("MASt3R = the same network with the catmlp+dpt head, then reciprocal descriptor matching; " if mast else "") + f"configs[4]: DUSt3R ViT-L/16 encoder (24 x 1024) + 2 x 12 x 768 cross-attention decoder + DPT point-map heads on synthetic {Ww}x{Hh} "
                                   "pairs resident in HBM; one pair = the symmetrised call of duster.py (directed pairs (1,0) and (0,1), each image encoded once)",
                       "pairs_per_step_per_gpu": B, "weights": "seeded random (imcui_hip/synth_weights.py), AsymmetricCroCo3DStereo architecture, 578 M parameters",
                       "mean_confidence": float(out["conf"].mean()),
                       **({"head": "catmlp+dpt, 24-d descriptors; matching: fast_reciprocal_NNs(subsample 2, dot, 10 rounds) on the device, 2000 matches kept",
                           "matcher_arithmetic": args.nn_arith, "matches_per_pair": nmatch[0] / B, "network_ms_per_step": net_ms, "matching_ms_per_step": dt / args.steps * 1e3 - net_ms} if mast else {})},
            "roofline": {"kernel": "gemm_split_kernel + conv3x3_split_kernel + attn_split_kernel (matrix class)", "bound": "mfma", "achieved": ach,
                         "peak": PEAK_F16_MFMA_TF, "unit": "TFLOP/s", "frac": ach / PEAK_F16_MFMA_TF, "traffic": traffic, "traffic_source": traffic_source,
                         "class_ms_per_step": {k: v[0] / args.steps for k, v in cls_ms.items()}, "launches_per_step": mat_n / args.steps,
                         "algorithmic_tflop_per_pair": tf,
                         "note": "achieved = algorithmic TFLOP of a pair / summed matrix-class kernel time (HIP events); the split mode executes 3 f16 MFMAs per product"},
            "arithmetic": args.arith,
            "algorithmic_tflops_end_to_end": tf["total"] * B / (dt / args.steps),
        }  # fmt: skip
        if world == 1 and not args.no_cpu_baseline:
            ncpu = os.cpu_count() or 1
            ck = ("mast3r" if mast else "dust3r", Hh, Ww)
            if ck not in _DUST3R_CPU:  # (the fp16 leg of the same process re-uses the oracle run of the split leg: same pair, same weights)
                from oracle.dust3r import DUSt3ROracle, MASt3ROracle

                torch.set_num_threads(min(ncpu, 32))
                ora = (MASt3ROracle if mast else DUSt3ROracle)(sd if sd is not None else dust3r_state_dict(0, cfg), cfg)
                t0 = time.perf_counter()
                ref = ora.inference_symmetrized(i0, i1)
                el = time.perf_counter() - t0
                del ora
                rec = {"value": 1.0 / el, "unit": "pairs/s", "cores": torch.get_num_threads(), "host_cpus": ncpu, "kind": "port",
                       "sample": f"ONE synthetic {Ww}x{Hh} pair, no warm-up, fp32, the oracle's restatement of duster.py:66-73 (two forward passes, "
                                 f"both images encoded in each, as upstream's inference does{'; the NETWORK only -- the reciprocal matching (TFLOPs of dot products per round on the CPU) is not in the sample' if mast else ''}), torch {torch.__version__} CPU, "
                                 f"{torch.get_num_threads()} of {ncpu} host CPUs"}  # fmt: skip
                _DUST3R_CPU[ck] = (rec, {k: {kk: vv for kk, vv in ref[k].items() if kk in ("pts3d", "pts3d_in_other_view", "conf")} for k in ("pred1", "pred2")})
            rec, ref = _DUST3R_CPU[ck]
            if not args.no_parity:
                # parity of pair 0 with the oracle run timed for cpu_baseline (batch entries in make_pairs' order: (image1, image0), (image0, image1)).
                # fp32 (3 x f16 split) = the parity mode: 1e-4 of the scene scale (measured 1.1e-5); fp16 (one product, bf16-class, NOT a parity mode): the 5e-3
                # bar of tests/test_gpu_dust3r.py, which also anchors it below a bf16-autocast run of the oracle
                bar = 1e-4 if args.arith == "fp32" else 5e-3
                got = model.forward_pairs(images[:2], [[1, 0], [0, 1]])
                worst = 0.0
                for v, key in ((0, "pred1"), (1, "pred2")):
                    rp = ref[key]["pts3d" if v == 0 else "pts3d_in_other_view"]
                    err = (got["pts3d"][v].cpu() - rp).abs().max().item() / rp.abs().max().item()
                    cerr = ((got["conf"][v].cpu() - ref[key]["conf"]).abs() / ref[key]["conf"]).max().item()
                    worst = max(worst, err, cerr)
                if not worst < bar:
                    raise AssertionError(f"bench parity (dust3r, {args.arith}): point maps / confidences differ from the oracle by {worst:.2e} (bar {bar:.0e})")
                line["parity"] = {"status": "ok", "checked": f"pair 0 vs the CPU oracle run timed for cpu_baseline: point maps within {bar:.0e} of the scene scale, confidences within {bar:.0e} relative"
                                                            + ("" if args.arith == "fp32" else " (single-product arithmetic: bf16-class bar, not the fp32 parity mode)"),
                                  "max_error": worst, "bar": bar}  # fmt: skip
            line["cpu_baseline"] = rec
        return line
    return None


def bench_superglue(args, dev, rank, world):
    """SuperPoint + SuperGlue (matcher zoo entry `superglue`, imcui/hloc/configs/matchers.py:10-24: 50 Sinkhorn rounds) on
    640x480 pairs; pairs/s, weak scaling (pairs are independent; one all-gather of the match tables per step)."""
    from imcui_hip import backend
    from imcui_hip.distributed import TableGather
    from imcui_hip.pipeline import SuperPointSuperGluePipeline, match_table
    from imcui_hip.synth import make_pair_batch
    from imcui_hip.synth_weights import superglue_state_dict, superpoint_state_dict  # seeded weights only

    B = args.batch
    spc = {"nms_radius": 3, "max_keypoints": MAXK, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)}
    pipe = SuperPointSuperGluePipeline(spc, {"sinkhorn_iterations": args.sinkhorn, "match_threshold": 0.2, "state_dict": superglue_state_dict(0)}).eval().to(dev)
    img0, img1, _ = make_pair_batch(1234 + rank, B, H, W, distinct=min(B, 4))
    img0, img1 = img0.to(dev), img1.to(dev)

    gather = TableGather(world, B, 3 + 2 * MAXK, torch.int32, dev, force=FORCE_DIST)

    def step():
        out = pipe(img0, img1)
        if dist_on(world):  # the one exchange step of the path: all-gather of the match tables
            gather(match_table(out))
        return out

    def timed(steps):
        torch.cuda.synchronize()
        if dist_on(world):
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        gather.finish()
        torch.cuda.synchronize()
        if dist_on(world):
            dist.barrier()
        return time.perf_counter() - t0, out

    for _ in range(args.warmup):
        step()
    backend.profile_enable(dev, True)
    dt, out = timed(args.steps)
    attn_ms, attn_n = backend.profile_read(dev, "attention")
    conv_ms, _ = backend.profile_read(dev, "conv3x3")
    gemm_ms, _ = backend.profile_read(dev, "gemm")
    backend.profile_enable(dev, False)
    # the optimal transport alone: the same step without Sinkhorn rounds (outside the timed region)
    pipe.matcher.conf["runtime_match_threshold"] = True  # the plugin freezes its conf at _init like the reference; this leg re-reads it
    pipe.matcher.conf["sinkhorn_iterations"] = 0
    pipe(img0, img1)
    dt0, _ = timed(max(2, args.steps // 2))
    dt0 *= args.steps / max(2, args.steps // 2)
    if dist_on(world):
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        nk0, nk1 = out["num_keypoints0"].float().mean().item(), out["num_keypoints1"].float().mean().item()
        split = args.precision == 1
        peak = PEAK_F16_MFMA_TF if split else PEAK_F32_MFMA_TF
        attn_flops = B * 8.59e9 * (nk0 * nk1 / (MAXK * MAXK))  # one launch: 2 images x 4 heads x (QK^T + PV) at 2048 x 2048 x 64
        achieved = attn_flops / (attn_ms / max(attn_n, 1) * 1e-3) / 1e12 if attn_n else 0.0
        sk_ms = max(dt - dt0, 0.0) / args.steps * 1e3
        sk_bytes = args.sinkhorn * 4.0 * nk0 * nk1 * B  # the fused round reads the score matrix once (row pass + column statistics)
        line = {
            "metric": "image-pairs/sec @640x480 SuperPoint+SuperGlue", "value": world * B * args.steps / dt, "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 via 3xf16 split MFMA, f32 accumulate" if split else "f32", "data": "synthetic",
            "config": {"workload": "SuperPoint(max 2048 kpts, nms 3)+SuperGlue(18 layers, Sinkhorn) on synthetic 640x480 pairs resident in HBM",
                       "pairs_per_step_per_gpu": B, "sinkhorn_iterations": args.sinkhorn, "mean_keypoints": [nk0, nk1],
                       "mean_matches": float((out["matches0"] > -1).sum(1).float().mean()),
                       "weights": "seeded random (imcui_hip/synth_weights.py), real architecture"},
            "roofline": {"kernel": "attn_split_kernel (3xf16 split MFMA flash attention)" if split else "attn_kernel", "bound": "mfma",
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                         "avg_launch_ms": attn_ms / max(attn_n, 1), "launches": attn_n},
            "sinkhorn": {"bound": "hbm", "ms_per_step": sk_ms, "algorithmic_bytes_per_step": sk_bytes,
                         "achieved": sk_bytes / (sk_ms * 1e-3) / 1e9 if sk_ms > 0 else None, "peak": 8000.0, "unit": "GB/s",
                         "note": "step time minus the same step with 0 Sinkhorn rounds; rounds * 4 * n0 * n1 bytes per pair (one matrix read per round; "
                                 "the band partials add 12 %)"},
            "kernel_time_ms_per_step": {"attention": attn_ms / args.steps, "conv3x3": conv_ms / args.steps, "gemm": gemm_ms / args.steps},
        }  # fmt: skip
        if not args.no_parity and world == 1:
            # pair 0 of the batch: the matcher against oracle/superglue.py on the key-points / descriptors the HIP extractor produced
            # (the extractor itself is checked by the headline and the superpoint leg)
            try:
                from oracle.superglue import SuperGlueOracle
                from imcui_hip.synth_weights import superglue_state_dict as _sgsd

                n0, n1 = int(out["num_keypoints0"][0]), int(out["num_keypoints1"][0])
                t0 = time.perf_counter()
                ref = SuperGlueOracle(_sgsd(0), {"sinkhorn_iterations": args.sinkhorn, "match_threshold": 0.2})(
                    {"image0": img0[:1].cpu(), "image1": img1[:1].cpu(), "keypoints0": out["keypoints0"][0, :n0].cpu()[None], "keypoints1": out["keypoints1"][0, :n1].cpu()[None],
                     "scores0": out["scores0"][0, :n0].cpu()[None], "scores1": out["scores1"][0, :n1].cpu()[None],
                     "descriptors0": out["descriptors0"][0, :n0].cpu().t()[None], "descriptors1": out["descriptors1"][0, :n1].cpu().t()[None]})  # fmt: skip
                cpu_s = time.perf_counter() - t0
                mh, mr = out["matches0"][0, :n0].cpu().long(), ref["matches0"][0].long()
                same = mh == mr
                err = (out["matching_scores0"][0, :n0].cpu() - ref["matching_scores0"][0]).abs()[same].max().item() if same.any() else 0.0
                ok = int((~same).sum()) <= 2 and err < 1e-4
                line["parity"] = {"status": "ok" if ok else "MISMATCH", "checked": "pair 0: matcher vs oracle/superglue.py on the HIP extractor's key-points and descriptors",
                                  "matches": int((mr > -1).sum()), "differing_rows": int((~same).sum()), "max_score_error": err, "oracle_matcher_seconds": round(cpu_s, 2)}
            except Exception as e:  # noqa: BLE001 -- a parity record, not a crash
                line["parity"] = {"status": f"not run: {type(e).__name__}: {e}"}
        return line
    return None


def _pil_luma(blob: bytes):
    """PIL's (libjpeg's) gray decode of a JPEG file = its luma plane: what cv2.imread(IMREAD_GRAYSCALE) returns."""
    import io

    import numpy as np
    from PIL import Image

    im = Image.open(io.BytesIO(blob))
    im.draft("L", im.size)
    return np.array(im)


def ensure_built() -> None:
    """Build libimcui_hip.so when it is missing or older than its sources (no-op otherwise).  `imcui_hip.build.build` takes an exclusive
    file lock, so the N ranks of one launch on a clean checkout compile once and the others wait."""
    from imcui_hip import build as b

    b.build(force=False)


def ranks_agree(ok: bool, world: int, dev) -> bool:
    """True when EVERY rank reports ok.  The legs call it after set-up and warm-up, before their barrier-bracketed timed loop, so a
    rank that failed (out of memory, a missing file) makes all ranks skip the leg together instead of leaving the others in a barrier."""
    if not dist_on(world):
        return ok
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


# the BASELINE.json configs other than the headline, run after it in the same process by the default invocation and attached to the one
# JSON line as "workloads": {name: line}; (workload, overrides of the command-line arguments)
LEGS = {
    # reference-default LightGlue (depth 0.95 / width 0.99, imcui/hloc/matchers/lightglue.py:15-25): every pair of the batch its own scene
    "splg_adaptive": ("splg", {"batch": 64, "adaptive": True, "no_cpu_baseline": True}),
    # the reference's own call pattern through the plugin seam: one image / one pair per `_forward`, results brought to the host
    "splg_b1_seam": ("seam", {"batch": 1}),
    "nn": ("nn", {"batch": 64}),
    "superpoint": ("superpoint", {"batch": 64}),
    "loftr_1024": ("loftr", {"batch": 16, "size": None}),     # pairs per step: 4 / 8 / 16 = 101.7 / 105.5 / 107.9 pairs/s on one box (round 4)
    "dust3r_512": ("dust3r", {"batch": 32, "arith": "fp32", "size": None}),  # 8 / 16 / 32 = 86.4 / 90.4 / 92.4
    "dust3r_512_fp16": ("dust3r", {"batch": 32, "arith": "fp16", "size": None}),
    # SURVEY section 8 f-rows: the other matchers of the zoo that run on this backend
    "eloftr_640x480": ("eloftr", {"batch": 32, "size": None}),  # (8 pairs = 600 workgroups of the fused MLP = 2.3 rounds of the chip: 498 pairs/s; 16: 561; 32: 593)
    "mast3r_512": ("mast3r", {"batch": 16, "arith": "fp32", "size": None}),
    "superglue": ("superglue", {"batch": 64}),
}


def legs_enabled(args) -> bool:
    """The default invocation (`python bench.py --gpus N --steps K --warmup W`, what the driver runs) carries the legs; any switch that
    turns the run into a profiler pass or an A/B leg (--no-parity, --no-cpu-baseline, --adaptive, --graph, --h2d, --precision 0, --batch,
    another --workload, --gpus > 1) does not, and --no-legs / --legs select explicitly."""
    if args.no_legs:
        return False
    if args.legs is not None:
        return True
    # N > 1: the scaling runs measure the headline; the legs stay off unless asked for (`--legs all`): a leg that failed on ONE rank
    # would leave the others in a barrier and take the headline line down with it, and none of this has met an 8-GPU node yet
    return (args.workload == "splg" and not (args.no_parity or args.no_cpu_baseline or args.adaptive or args.graph or args.h2d)
            and args.precision == 1 and args.batch_given is None and args.gpus == 1)  # fmt: skip


def run_legs(args, dev, rank, world) -> dict:
    """Short legs of configs[0], [1], [3], [4] (and [4] in its bf16-class arithmetic) under the same clock discipline as the headline:
    W' = min(W, 3) warm-up steps, K' = min(K, 10) timed steps, barrier + synchronize on both sides, MAX over ranks.  Each returns the line
    `--workload <name>` prints on its own -- value, unit, ms_per_step, steps, config, dtype, roofline (dominant-kernel class time from
    the library's HIP-event hooks inside the timed loop), cpu_baseline and parity (world 1 only) -- and a failure of one leg is recorded
    in its entry instead of taking the headline down."""
    import argparse
    import gc
    import traceback

    names = list(LEGS) if args.legs in (None, "all") else [n for n in args.legs.split(",") if n]
    out = {}
    for name in names:
        if name not in LEGS:
            out[name] = {"status": "unknown leg", "known": list(LEGS)}
            continue
        workload, over = LEGS[name]
        a = argparse.Namespace(**{**vars(args), **over, "workload": workload, "steps": min(args.steps, 10), "warmup": min(args.warmup, 3), "nn_arith": "auto"})
        gc.collect()
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        fn = {"loftr": bench_loftr, "eloftr": bench_loftr, "dust3r": bench_dust3r, "mast3r": bench_dust3r, "superpoint": bench_superpoint, "nn": bench_nn,
              "splg": bench_splg, "seam": bench_seam, "superglue": bench_superglue}[workload]
        try:
            line = fn(a, dev, rank, world)
        except LegSkipped as e:
            line = {"status": "skipped", "reason": str(e)}
        except Exception as e:  # noqa: BLE001 -- the headline line must survive a broken leg; the entry says what happened
            if dist_on(world):
                raise  # ranks may be out of step with each other: no way to continue the collectives safely
            line = {"status": "failed", "error": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()[-1500:]}
        if line is not None:
            line.setdefault("status", "ok")
            line["leg_wall_s"] = round(time.perf_counter() - t0, 1)
        out[name] = line
        torch.cuda.synchronize()
    return out


class LegSkipped(RuntimeError):
    """Raised on every rank together (after `ranks_agree`) when some rank could not set a leg up."""


def rank_env(args):
    """(rank, local_rank, world) from the launcher's environment; the rank count must be what --gpus asked for."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    return rank, local_rank, world


def self_launch(args) -> None:
    """`python bench.py --gpus N` with no launcher around it: re-execute under torch.distributed.run with one rank per
    GPU (exactly the command line the driver uses for N > 1), forwarding every argument.  Rank 0 of the child job
    prints the JSON line."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]  # fmt: skip
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    raise SystemExit(subprocess.call(cmd, env=env))


def launchcheck(args) -> None:
    """CPU check of the launch / sharding / timing scaffold (tests/test_distributed_cpu.py): the same self-launch,
    rank environment, barrier-bracketed timing, MAX-over-ranks and match-table all-gather as the GPU workloads, on the
    gloo backend with a synthetic match table instead of HIP kernels.  Not a benchmark."""
    from imcui_hip.distributed import TableGather

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    rank, _, world = rank_env(args)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        assert dist.get_world_size() == args.gpus
    B, K = 4, 64
    gather = TableGather(world, B, 3 + 2 * K, torch.int32, "cpu")
    table = torch.full((B, 3 + 2 * K), rank, dtype=torch.int32)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = gather(table)
    gather.finish()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        assert got.shape[0] == world * B and all(int(got[r * B, 0]) == r for r in range(world))
    if rank == 0:
        print(json.dumps({"metric": "launchcheck (no HIP work)", "value": world * B * args.steps / dt, "unit": "pairs/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "config": {"workload": "launchcheck", "parallelism": f"gloo x{world}"}}), flush=True)  # fmt: skip
    if world > 1:
        dist.destroy_process_group()


ATTENTION_SOURCES = ("image-matching-webui_amd/csrc/attention.hip", "image-matching-webui_amd/csrc/attention.h", "image-matching-webui_amd/csrc/gemm_wreg.hip")  # what decides the attention kernel's HBM traffic (kernel, launch geometry, the plane layout its producer writes)


def sources_sha16(paths=ATTENTION_SOURCES) -> dict:
    """sha256 prefixes of the kernel sources: tools/summarize_profiles.py records them beside the PMC traffic it writes, bench.py compares them
    with the tree that runs (the GPU box has no .git, so a commit id cannot be checked there; file contents can)."""
    import hashlib

    out = {}
    for rel in paths:
        try:
            out[rel] = hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()[:16]
        except OSError:
            out[rel] = None
    return out


def _cross_variant(dev) -> int:
    """The attention variant LightGlue's cross blocks run (csrc/lightglue.hip): option attn_variant_cross, whose default -2 means 7 (two-product P.V)
    while attn_variant is the default kernel 8 and "follow attn_variant" otherwise; -1 = follow attn_variant."""
    from imcui_hip import backend

    cv, av = backend.get_option(dev, "attn_variant_cross"), backend.get_option(dev, "attn_variant")
    return (7 if av == 8 else -1) if cv == -2 else cv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="pairs per step per GPU (default 64; loftr: 16, eloftr: 32, dust3r: 32, mast3r: 16)")
    ap.add_argument("--adaptive", action="store_true", help="reference defaults depth 0.95 / width 0.99 (data dependent work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of one pair after the timed region (profiler passes)")
    ap.add_argument("--graph", action="store_true", help="splg: replay the step from a captured HIP graph (small-batch latency)")
    ap.add_argument("--decode-threads", type=int, default=32, help="--h2d jpeg: host threads of the Huffman stage")
    ap.add_argument("--h2d", nargs="?", const="raw", default=None, choices=["raw", "jpeg"],
                    help="splg: `--h2d` / `--h2d raw`: the uint8 images of every step are uploaded from pinned host memory inside the timed "
                                                       "region (the PCIe-inclusive rate quoted in DESIGN.md; never the headline `value`)")
    ap.add_argument("--workload", default="splg", choices=["splg", "nn", "loftr", "eloftr", "dust3r", "mast3r", "superpoint", "superglue", "launchcheck"],
                    help="splg = BASELINE metric (SuperPoint+LightGlue 640x480); nn = configs[0] mutual-NN matcher on 5000 x 128-d descriptors; loftr = configs[3] LoFTR dense matcher; "
                         "superpoint = configs[1] extractor only (images/s); superglue = SuperPoint+SuperGlue pairs; eloftr = EfficientLoFTR 640x480; "
                         "dust3r = configs[4] DUSt3R pair network 512x512; mast3r = the same network with the descriptor head + reciprocal matching")
    ap.add_argument("--sinkhorn", type=int, default=50, help="superglue: Sinkhorn rounds (zoo conf `superglue` = 50, `superglue-fast` = 5)")
    ap.add_argument("--size", type=int, nargs=2, default=None, metavar=("H", "W"), help="loftr image size (default 1024 1024)")
    ap.add_argument("--nn-arith", default="auto", choices=["auto", "fp32", "split"],
                    help="mast3r: arithmetic of the nearest-neighbour searches (auto = follows --precision: split by default; fp32 = exact-f32 MFMA)")
    ap.add_argument("--arith", default="fp32", choices=["fp32", "fp16"],
                    help="dust3r: fp32 = 3 x f16 split products (default, the parity mode), fp16 = one f16 product per element pair (bf16-class)")
    ap.add_argument("--precision", type=int, default=1, choices=[0, 1],
                    help="0 = exact f32 MFMA, 1 = 3 x f16 split MFMA with f32 accumulate (default, parity-tested)")
    ap.add_argument("--legs", default=None, help="comma-separated legs to run after the workload and attach as \"workloads\" (" + ", ".join(LEGS) + "; `all`); default: all of "
                                                 "them on the plain headline invocation, none otherwise")
    ap.add_argument("--no-legs", action="store_true", help="headline line only")
    ap.add_argument("--fine-dense", action="store_true", help="loftr: the last FPN stage as dense maps whatever the match count (option loftr_fine_sparse = 0; A/B)")
    args = ap.parse_args()
    args.batch_given = args.batch
    if args.batch is None:
        args.batch = 16 if args.workload == "loftr" else 32 if args.workload == "eloftr" else 32 if args.workload == "dust3r" else 16 if args.workload == "mast3r" else 64  # pairs per step and GPU (64: +2.5 % over 32, same kernels; dust3r 8 / 16 / 32: 86.5 / 90.5 / 92.7 pairs/s)

    ensure_built()  # a clean checkout / an N-rank launch builds libimcui_hip.so once, under a file lock
    if args.workload == "launchcheck":
        return launchcheck(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    rank, local_rank, world = rank_env(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on(world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        if rank == 0:
            print(f"[bench] RCCL process group up: {world} rank(s), backend {dist.get_backend()}", file=sys.stderr, flush=True)

    from imcui_hip import backend

    backend.set_precision(dev, args.precision)
    if args.fine_dense:
        backend.set_option(dev, "loftr_fine_sparse", 0)
    fn = {"splg": bench_splg, "loftr": bench_loftr, "eloftr": bench_loftr, "dust3r": bench_dust3r, "mast3r": bench_dust3r, "superpoint": bench_superpoint,
          "superglue": bench_superglue, "nn": bench_nn}[args.workload]  # fmt: skip
    line = fn(args, dev, rank, world)
    if legs_enabled(args):
        legs = run_legs(args, dev, rank, world)
        if line is not None:
            line["workloads"] = legs
    if rank == 0:
        # The driver keeps an 8 KB tail of stdout: the ONE JSON line is the compact record (every leg's value, unit, ms per step, roofline
        # fraction, parity and CPU figure survive in it); the full record -- per-leg configs, samples, per-pair parity -- goes to stderr
        # first (`python bench.py 2> detail.log`; profiles/ keeps both).
        print("[bench detail] " + json.dumps(line), file=sys.stderr, flush=True)
        print(json.dumps(compact_line(line)), flush=True)
    if dist_on(world):
        dist.destroy_process_group()


def _short(v, n=160):
    return v if not isinstance(v, str) or len(v) <= n else v[: n - 3] + "..."


def compact_line(line: dict) -> dict:
    """The record the driver must be able to hold (< 8 KB): the contract's keys in full, `roofline` / `cpu_baseline` / `parity` reduced to their
    numbers, and one short entry per leg under "legs" (the verbose per-leg lines are printed on stderr)."""
    if line is None:
        return line

    def roof(r):
        if not isinstance(r, dict):
            return r
        keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches")
        out = {k: (_short(r[k], 80) if k == "kernel" else r[k]) for k in keep if k in r}
        if r.get("traffic_source"):
            out["traffic_source"] = "committed PMC passes under profiles/ (not this run)"
            out["traffic_head"], out["traffic_stale"] = r.get("traffic_head"), r.get("traffic_stale")  # commit of the passes; kernel sources changed since ([] = none)
        return out

    def cpu(c):
        if not isinstance(c, dict):
            return c
        return {k: (_short(c[k], 120) if k == "sample" else c[k]) for k in ("value", "unit", "cores", "kind", "sample") if k in c}

    def par(q):
        if not isinstance(q, dict):
            return q
        skip = ("per_pair", "checked")
        return {k: (_short(v, 100) if isinstance(v, str) else v) for k, v in q.items() if k not in skip and not isinstance(v, (list, dict))}

    def leg(v):
        if not isinstance(v, dict):
            return v
        if v.get("status", "ok") != "ok":
            return {"status": v.get("status"), "why": _short(v.get("reason") or v.get("error") or "", 160)}
        r = v.get("roofline") or {}
        c = v.get("cpu_baseline") or {}
        q = v.get("parity") or {}
        e = {"value": v.get("value"), "unit": v.get("unit"), "ms_per_step": v.get("ms_per_step"), "per_step": (v.get("config") or {}).get("pairs_per_step_per_gpu") or (v.get("config") or {}).get("images_per_step_per_gpu"),
             "dtype": _short(v.get("dtype", ""), 48), "bound": r.get("bound"), "frac": r.get("frac"), "achieved": r.get("achieved"), "roof_unit": r.get("unit"),
             "parity": q.get("status"), "parity_max_err": next((q[k] for k in ("max_score_error", "max_error", "max_rel_error", "max_keypoint_error_px") if k in q), None),
             "cpu": c.get("value"), "cpu_cores": c.get("cores")}
        for k in ("seam", "mean_stop_layer"):
            if k in v:
                e[k] = v[k]
        if (v.get("config") or {}).get("mean_stop_layer") is not None:
            e["mean_stop_layer"] = v["config"]["mean_stop_layer"]
        fs = (v.get("config") or {}).get("fine_stage")
        if isinstance(fs, dict):  # LoFTR: how the last FPN stage ran (data dependent) and the dense-map figure measured beside it
            e["fine_stage"] = {k: fs[k] for k in ("mode", "matches_per_step", "dense_maps_pairs_per_s", "at_realistic_match_count") if k in fs}
        return {k: x for k, x in e.items() if x is not None}

    out = {}
    for k, v in line.items():
        if k == "workloads":
            out["legs"] = {n: leg(x) for n, x in v.items()}
        elif k == "roofline":
            out[k] = roof(v)
        elif k == "cpu_baseline":
            out[k] = cpu(v)
        elif k == "parity":
            out[k] = par(v)
        elif k == "config" and isinstance(v, dict):
            out[k] = {kk: _short(vv, 200) for kk, vv in v.items()}
        else:
            out[k] = v
    out["detail"] = "full record (per-leg configs, samples, per-pair parity): the `[bench detail]` line on stderr"
    return out


def bench_seam(args, dev, rank, world):
    """The reference's own call pattern (missing 2): the UI, the API and `match_from_paths` call ONE image per extractor
    `_forward` and ONE pair per matcher `_forward`, and bring the results to the host after each (imcui/hloc/extract_features.py:106-170
    `extract`, imcui/hloc/match_features.py:204-240 `match_images`: `pred = model(input_dict)`, then `v.cpu()` of every output;
    imcui/ui/utils.py:832-1095, imcui/api/core.py:108-128).  This leg times exactly that through the plugin seam -- `SuperPoint(...)(
    {"image": img})` twice, the tensors re-assembled the way `match_images` does, `LightGlue(...)(data)`, `.cpu()` -- one pair after the
    other, eager launches, in the fixed-work conf of the headline and in the reference's default adaptive conf.  What a drop-in user of
    the seam sees per pair; the batched drivers (the headline) are what `imcui_hip/hloc/match_features.py` adds on top."""
    from imcui_hip.hloc.extractors.superpoint import SuperPoint
    from imcui_hip.hloc.matchers.lightglue import LightGlue
    from imcui_hip.synth import make_pair_batch
    from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict  # seeded weights only

    if dist_on(world):
        raise LegSkipped("the seam leg is a single-GPU latency figure")
    spc = {"nms_radius": 3, "max_keypoints": MAXK, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)}
    exts = {False: SuperPoint(dict(spc)).eval().to(dev), True: SuperPoint({**spc, "hip_graph": True}).eval().to(dev)}
    ext = exts[False]
    npairs = 8
    img0, img1, _ = make_pair_batch(4321, npairs, H, W, distinct=npairs)
    img0, img1 = img0.to(dev), img1.to(dev)

    def extract(img):  # extract_features.extract: pred = model({"image": image}); the UI keeps the tensors, hloc's main() takes them to the host
        pred = ext({"image": img})
        return {**pred, "image": img}

    def match_images(model, feat0, feat1):  # match_features.match_images, up to the `.cpu()` of the outputs
        desc0, desc1 = feat0["descriptors"][0], feat1["descriptors"][0]
        data = {"image0": feat0["image"], "keypoints0": feat0["keypoints"][0][None], "scores0": feat0["scores"][0].unsqueeze(0), "descriptors0": desc0.unsqueeze(0),
                "image1": feat1["image"], "keypoints1": feat1["keypoints"][0][None], "scores1": feat1["scores"][0].unsqueeze(0), "descriptors1": desc1.unsqueeze(0)}  # fmt: skip
        pred = model(data)
        pred = {k: v.cpu().detach()[0] if isinstance(v, torch.Tensor) else v for k, v in pred.items()}
        kpts0, kpts1 = feat0["keypoints"][0].cpu().numpy(), feat1["keypoints"][0].cpu().numpy()
        valid = pred["matches0"] > -1
        return kpts0[valid.numpy()], kpts1[pred["matches0"][valid].numpy()], pred["matching_scores0"][valid]

    res = {}
    with torch.no_grad():
        # eager launches (the drop-in default) and the plugins' opt-in `hip_graph` conf (the same calls replayed from HIP graphs captured per
        # image shape / key-point capacity: outputs identical, tests/test_gpu_plugin_graph.py)
        for tag, (dc, wc), graphed in (("fixed_work", (-1.0, -1.0), False), ("reference_default_adaptive", (0.95, 0.99), False),
                                       ("fixed_work_hip_graph", (-1.0, -1.0), True), ("reference_default_adaptive_hip_graph", (0.95, 0.99), True)):  # fmt: skip
            ext = exts[graphed]
            matcher = LightGlue({"depth_confidence": dc, "width_confidence": wc, "match_threshold": 0.1, "state_dict": lightglue_state_dict(0), "hip_graph": graphed}).eval().to(dev)

            def one(i):
                f0, f1 = extract(img0[i : i + 1]), extract(img1[i : i + 1])
                return match_images(matcher, f0, f1)

            for i in range(min(args.warmup, 3) + 1):
                one(i % npairs)
            torch.cuda.synchronize()
            n = max(args.steps, 10) * 2
            t0 = time.perf_counter()
            nm = 0
            for i in range(n):
                nm += len(one(i % npairs)[0])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[tag] = {"ms_per_pair": dt / n * 1e3, "pairs_per_s": n / dt, "pairs_timed": n, "mean_matches": nm / n}
    fw = res["fixed_work"]
    return {"metric": "image-pairs/sec @640x480 SuperPoint+LightGlue through the plugin seam, one pair per call (the reference's call pattern)",
            "value": fw["pairs_per_s"], "unit": "pairs/s", "n_gpus": 1, "steps": fw["pairs_timed"], "warmup": min(args.warmup, 3) + 1, "ms_per_step": fw["ms_per_pair"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 via 3xf16 split MFMA, f32 accumulate", "data": "synthetic",
            "config": {"workload": "configs[2] at the reference's call granularity: SuperPoint._forward x 2 + LightGlue._forward + .cpu() of every output per pair, eager launches "
                                   "(value) and the plugins' opt-in hip_graph conf (seam.*_hip_graph), 8 distinct synthetic 640x480 scenes in turn", "pairs_per_step_per_gpu": 1},
            "roofline": {"bound": "launch/latency", "achieved": None, "peak": None, "unit": None, "frac": None,
                         "note": "one pair does not fill the chip: ~250 launches of a few microseconds each and three host round trips; no roofline fraction is claimed"},
            "seam": {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in res.items()}}  # fmt: skip


def bench_splg(args, dev, rank, world):
    """configs[2], the BASELINE metric: SuperPoint + LightGlue on 640x480 pair batches (module docstring)."""
    from imcui_hip import backend
    from imcui_hip.distributed import TableGather
    from imcui_hip.pipeline import SuperPointLightGluePipeline, match_table
    from imcui_hip.synth import make_pair_batch
    from imcui_hip.synth_weights import lightglue_state_dict, superpoint_state_dict  # seeded weights only

    failed = None
    try:  # set-up and warm-up: a failure on one rank must not leave the others in the timed loop's barrier
        B = args.batch
        dc, wc = (0.95, 0.99) if args.adaptive else (-1.0, -1.0)
        pipe = SuperPointLightGluePipeline(
            {"nms_radius": 3, "max_keypoints": MAXK, "keypoint_threshold": 0.005, "remove_borders": 4, "state_dict": superpoint_state_dict(0)},
            {"depth_confidence": dc, "width_confidence": wc, "match_threshold": 0.1, "state_dict": lightglue_state_dict(0)},
        ).eval().to(dev)
        torch.manual_seed(1234 + rank)
        # fixed-work run: the cost of a pair does not depend on its content, 8 generated pairs are tiled over the batch (host prep time);
        # --adaptive: the work IS the content, every pair of the batch is its own scene
        distinct = B if args.adaptive else min(B, 8)
        img0, img1, _ = make_pair_batch(1234 + rank, B, H, W, distinct=distinct)
        img0, img1 = img0.to(dev), img1.to(dev)
        gather = TableGather(world, B, 3 + 2 * MAXK, torch.int32, dev, force=FORCE_DIST)

        run = pipe
        if args.graph:
            from imcui_hip.pipeline import GraphedPipeline

            run = GraphedPipeline(pipe, img0, img1)

        if args.h2d == "jpeg":
            # what a caller that holds JPEG FILES pays: per step the 2 B files are entropy-decoded on host threads,
            # the luma coefficients travel over PCIe, and the pixels are reconstructed on the device (csrc/jpeg.hip: bit-exact libjpeg
            # arithmetic), one step ahead of the compute stream on a side stream driven by a feeder thread
            import io
            from concurrent.futures import ThreadPoolExecutor

            from PIL import Image

            from imcui_hip.hloc.utils.jpeg import JpegDecoder

            def to_jpeg(im):  # the synthetic scene as a colour photograph would be stored: 3 components, 4:2:0, quality 90
                g = (im[0] * 255.0).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
                buf = io.BytesIO()
                Image.fromarray(g).convert("RGB").save(buf, "JPEG", quality=90, subsampling="4:2:0")
                return buf.getvalue()

            blobs = [to_jpeg(im) for im in img0] + [to_jpeg(im) for im in img1]
            jdec = JpegDecoder(dev, threads=args.decode_threads)
            side = torch.cuda.Stream(device=dev)
            feeder = ThreadPoolExecutor(max_workers=1)
            fstate = {"next": None}

            def prepare():
                torch.cuda.set_device(dev)
                with torch.cuda.stream(side):
                    imgs = jdec.decode_batch(blobs, True)
                    batch = torch.stack(imgs)  # [2B,H,W] uint8 = the luma planes (what cv2.IMREAD_GRAYSCALE returns)
                    ev = torch.cuda.Event()
                    ev.record(side)
                return batch, ev

            host = None
            fstate["next"] = feeder.submit(prepare)
        elif args.h2d:
            # what a caller that holds decoded images in host memory pays: 2 B uint8 images per step over PCIe on a side stream, one step
            # ahead of the compute stream (two device buffers), then u8 -> f32 / 255 on the device
            host = [(im * 255.0).round().clamp(0, 255).to(torch.uint8).cpu().pin_memory() for im in (img0, img1)]
            copy_stream = torch.cuda.Stream(device=dev)
            slots = [[torch.empty_like(h, device=dev) for h in host] for _ in range(2)]
            ready = [torch.cuda.Event() for _ in range(2)]
            free = [torch.cuda.Event() for _ in range(2)]
            state = {"i": 0}

            def upload(k):
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(free[k])  # the step that read this slot has finished
                    for d, h in zip(slots[k], host):
                        d.copy_(h, non_blocking=True)
                    ready[k].record(copy_stream)

            for k in range(2):
                free[k].record(torch.cuda.current_stream(dev))
            upload(0)

        def step():
            if args.h2d == "jpeg":
                batch, ev = fstate["next"].result()
                fstate["next"] = feeder.submit(prepare)  # the next step's files are decoded while this step computes
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                batch.record_stream(cur)
                f = batch.float().div_(255.0)[:, None]
                out = run(f[:B], f[B:])
            elif args.h2d:
                k = state["i"] & 1
                state["i"] += 1
                upload(k ^ 1)  # next step's images travel while this step computes
                torch.cuda.current_stream(dev).wait_event(ready[k])
                a, b = (d.float() / 255.0 for d in slots[k])
                out = run(a, b)
                free[k].record(torch.cuda.current_stream(dev))
            else:
                out = run(img0, img1)
            if dist_on(world):
                gather(match_table(out))
            return out

        for _ in range(args.warmup):
            out = step()
    except Exception as e:  # noqa: BLE001
        if world == 1:
            raise
        failed = e
    if not ranks_agree(failed is None, world, dev):
        raise LegSkipped(f"set-up failed on a rank: {failed!r}")
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    backend.profile_enable(dev, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    gather.finish()
    torch.cuda.synchronize()
    if dist_on(world):
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if args.h2d == "jpeg":  # the feeder is one step ahead: drain it before the parity check uses the device
        fstate["next"].result()
        torch.cuda.synchronize()
        feeder.shutdown()
        jdec.close()
    attn_ms, attn_n = backend.profile_read(dev, "attention")
    conv_ms, conv_n = backend.profile_read(dev, "conv3x3")
    gemm_ms, gemm_n = backend.profile_read(dev, "gemm")
    backend.profile_enable(dev, False)
    if dist_on(world):
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        nk0 = out["num_keypoints0"].float().mean().item()
        nk1 = out["num_keypoints1"].float().mean().item()
        stop = out["stop"].float().mean().item()
        pairs = world * B * args.steps
        ms_step = dt / args.steps * 1e3
        # algorithmic flops of one attention launch: B pairs x 15.03 GF / 2 launches per layer
        attn_flops = B * ATTN_GF_PER_LAYER_PAIR * 1e9 / 2 * (nk0 * nk1 / (MAXK * MAXK))
        achieved = attn_flops / (attn_ms / max(attn_n, 1) * 1e-3) / 1e12 if attn_n else 0.0
        if args.adaptive:
            # early stopping and pruning skip work inside the launches (dead tiles exit): the full-size algorithmic flops above are NOT
            # what the kernel did, so no roofline fraction is claimed for this operating point
            achieved = None
        split = args.precision == 1
        peak = PEAK_F16_MFMA_TF if split else PEAK_F32_MFMA_TF
        # executed matrix flops: both cross directions recompute QK^T (8.59 + 8.59 vs 15.03 GF) and the
        # split mode issues 3 f16 MFMAs per product
        executed = achieved * (17.18 / 15.03) * (3.0 if split else 1.0) if achieved is not None else None
        # HBM-side bytes per attention launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate runs, FETCH doubled per MI355X_MICROARCH.md); scales with the batch
        traffic, traffic_source, traffic_head, traffic_stale = None, None, None, None
        import glob

        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_attention_traffic.json")))  # newest round's PMC passes
        tpath = cands[-1] if cands else ""
        if split and tpath:
            with open(tpath) as fh:
                tj = json.load(fh)
            traffic = tj["traffic_bytes_per_launch"] * B / tj["batch_pairs"]
            traffic_source = (f"NOT measured in this run: {os.path.relpath(tpath, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at "
                              f"{tj['batch_pairs']} pairs per step, committed), scaled to {B} pairs per step")
            # the commit the counters were taken at + whether the kernel's sources changed since
            traffic_head = str(tj.get("head", ""))
            rec = tj.get("sources_sha16")
            traffic_stale = None if not rec else sorted(k for k, v in sources_sha16(tuple(rec)).items() if v != rec[k])
            if traffic_stale:
                print(f"[bench] WARNING: roofline.traffic was collected at {traffic_head} and {traffic_stale} changed since: re-run the PMC passes "
                      "(tools/collect_profiles.sh, tools/summarize_profiles.py)", file=sys.stderr)
        line = {
            "metric": "image-pairs/sec @640x480 SuperPoint+LightGlue",
            "value": pairs / dt,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 via 3xf16 split MFMA, f32 accumulate" + ("" if _cross_variant(dev) < 0 else
                      "; P.V of LightGlue's cross blocks in two products (audited per block: layer error <= 7.1e-6, score error <= 4.7e-5, profiles/r05_lab_attention_mix.txt)")) if split else "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[2]: SuperPoint(max 2048 kpts, nms 3)+LightGlue(9 layers) on synthetic 640x480 pairs "
                            + ("decoded from JPEG files inside the timed region: Huffman on host threads, coefficients over PCIe, pixels reconstructed on the device (NOT the headline value)" if args.h2d == "jpeg" else
                               "uploaded as uint8 from pinned host memory inside the timed region (PCIe-inclusive; NOT the headline value)" if args.h2d else "resident in HBM"),
                "pairs_per_step_per_gpu": B, "global_pairs_per_step": world * B, "parallelism": f"pairs sharded over {world} rank(s) (RCCL world size {dist.get_world_size() if dist_on(world) else 1}), async all-gather of match tables",
                "lightglue_adaptive": bool(args.adaptive), "hip_graph": bool(args.graph), "h2d_inside_timed_region": args.h2d or False, **({"decode_threads": args.decode_threads, "jpeg_bytes_per_image": sum(len(b) for b in blobs) / len(blobs)} if args.h2d == "jpeg" else {}), "mean_keypoints": [nk0, nk1], "mean_stop_layer": stop,
                "weights": "seeded random (imcui_hip/synth_weights.py), real architecture",
            },
            "roofline": {
                "kernel": "attn_split_kernel (3xf16 split MFMA flash attention)" if split else "attn_kernel (f32 MFMA flash attention)",
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak if achieved is not None else None, "traffic": traffic, "traffic_source": traffic_source,
                "traffic_head": traffic_head, "traffic_stale": traffic_stale,  # commit of the PMC passes; attention sources changed since ([] = none, None = not recorded)
                "executed_tflops": executed, "executed_frac": executed / peak if executed is not None else None,
                "executed_frac_of_sustained_peak": (executed / SUSTAINED_F16_MFMA_TF) if split and executed is not None else None,
                **({"note": "adaptive depth / width: the work per launch is data dependent, no roofline fraction is claimed"} if args.adaptive else {}),
                "avg_launch_ms": attn_ms / max(attn_n, 1), "launches": attn_n, "algorithmic_gflop_per_launch": attn_flops / 1e9,
            },
            "kernel_time_ms_per_step": {"attention": attn_ms / args.steps, "conv3x3": conv_ms / args.steps, "gemm": gemm_ms / args.steps},
            "algorithmic_tflops_end_to_end": (2 * SP_GF_PER_IMAGE + 9 * LG_GF_PER_LAYER_PAIR + 2.7) * 1e9 * B / (ms_step * 1e-3) / 1e12,
        }  # fmt: skip
        if not args.no_parity:
            # (with --h2d the device saw the images quantised to uint8: the oracle gets the same values)
            if args.h2d == "jpeg":  # the device saw PIL-identical luma planes of the files: the oracle gets PIL's own decode
                dec8 = torch.stack([torch.from_numpy(_pil_luma(b)) for b in blobs[:4] + blobs[B : B + 4]])
                pa, pb = (dec8[:4].float() / 255.0)[:, None].to(dev), (dec8[4:].float() / 255.0)[:, None].to(dev)  # (the four pairs the parity check reads)
            else:
                pa, pb = ((h.float() / 255.0).to(dev) for h in host) if args.h2d else (img0, img1)
            nd = len(pa) if args.h2d == "jpeg" else (B if args.adaptive else min(B, 8))  # distinct scenes of the batch
            line["parity"] = parity_splg(pipe, pa, pb, dc, wc, which=sorted({0, 1 % B, 2 % B, 3 % B}), keypoints_only=[i for i in range(4, min(nd, 16))])
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        return line
    return None


if __name__ == "__main__":
    main()
