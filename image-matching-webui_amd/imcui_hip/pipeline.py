"""Batched SuperPoint + LightGlue pair pipeline (the unit of the throughput metric).

One call = B image pairs: SuperPoint on the 2B images, LightGlue on the B pairs, all on the
current stream with fixed-stride device tensors in between (no host synchronisation).  This is
what `bench.py` times and what a batched hloc driver would call; the reference itself only ever
runs batch 1 (imcui/hloc/match_features.py:172-174).
"""
from __future__ import annotations

import torch

from .hloc.extractors.superpoint import SuperPoint
from .hloc.matchers.lightglue import LightGlue


class SuperPointLightGluePipeline(torch.nn.Module):
    def __init__(self, sp_conf: dict, lg_conf: dict):
        super().__init__()
        self.extractor = SuperPoint(sp_conf)
        self.matcher = LightGlue(lg_conf)

    @torch.no_grad()
    def forward(self, image0: torch.Tensor, image1: torch.Tensor) -> dict:
        """image0/image1 [B,1,H,W] in [0,1] -> fixed-stride match table (device tensors)."""
        B = image0.shape[0]
        same = image0.shape == image1.shape
        if same:
            f = self.extractor.forward_batched(torch.cat([image0, image1], 0))
            k0, k1 = f["keypoints"][:B], f["keypoints"][B:]
            d0, d1 = f["descriptors"][:B], f["descriptors"][B:]
            n0, n1 = f["num_keypoints"][:B], f["num_keypoints"][B:]
            s0, s1 = f["scores"][:B], f["scores"][B:]
        else:
            f0 = self.extractor.forward_batched(image0)
            f1 = self.extractor.forward_batched(image1)
            k0, k1, d0, d1 = f0["keypoints"], f1["keypoints"], f0["descriptors"], f1["descriptors"]
            n0, n1, s0, s1 = f0["num_keypoints"], f1["num_keypoints"], f0["scores"], f1["scores"]
        size0 = (image0.shape[-1], image0.shape[-2])
        size1 = (image1.shape[-1], image1.shape[-2])
        m = self.matcher.forward_batched(k0, k1, d0, d1, n0, n1, size0, size1)
        return {
            "keypoints0": k0, "keypoints1": k1, "scores0": s0, "scores1": s1, "descriptors0": d0, "descriptors1": d1,
            "num_keypoints0": n0, "num_keypoints1": n1, **m,
        }  # fmt: skip


class SuperPointSuperGluePipeline(torch.nn.Module):
    """SuperPoint -> SuperGlue on B pairs (the `superpoint_max` + `superglue` entry of the matcher zoo,
    imcui/hloc/configs/matchers.py:10-24): images in, fixed-stride match table out, no host synchronisation."""

    def __init__(self, sp_conf: dict, sg_conf: dict):
        super().__init__()
        from .hloc.matchers.superglue import SuperGlue

        self.extractor = SuperPoint(sp_conf)
        self.matcher = SuperGlue(sg_conf)

    @torch.no_grad()
    def forward(self, image0: torch.Tensor, image1: torch.Tensor) -> dict:
        B = image0.shape[0]
        if image0.shape == image1.shape:
            f = self.extractor.forward_batched(torch.cat([image0, image1], 0))
            f0 = {k: v[:B] for k, v in f.items() if k in ("keypoints", "descriptors", "num_keypoints", "scores")}
            f1 = {k: v[B:] for k, v in f.items() if k in ("keypoints", "descriptors", "num_keypoints", "scores")}
        else:
            f0 = self.extractor.forward_batched(image0)
            f1 = self.extractor.forward_batched(image1)
        size0 = (image0.shape[-1], image0.shape[-2])
        size1 = (image1.shape[-1], image1.shape[-2])
        m = self.matcher.forward_batched(f0["keypoints"], f1["keypoints"], f0["scores"], f1["scores"], f0["descriptors"],
                                         f1["descriptors"], f0["num_keypoints"], f1["num_keypoints"], size0, size1)  # fmt: skip
        return {
            "keypoints0": f0["keypoints"], "keypoints1": f1["keypoints"], "scores0": f0["scores"], "scores1": f1["scores"],
            "descriptors0": f0["descriptors"], "descriptors1": f1["descriptors"], "num_keypoints0": f0["num_keypoints"],
            "num_keypoints1": f1["num_keypoints"], **m,
        }  # fmt: skip


class GraphedPipeline:
    """One forward of a pipeline captured in a HIP graph and replayed (fixed image shapes and batch size).

    The batched path never synchronises with the host -- key-point counts, early stopping and point pruning live on
    the device and every kernel is launched for the worst case -- so the whole extract + match step (~250 launches)
    is capturable.  This is for the latency-bound small-batch case (the UI / API match one pair at a time); at the
    bench batch the launches are already hidden.  The returned tensors are the graph's static outputs: consume or
    copy them before the next call."""

    def __init__(self, pipe: torch.nn.Module, image0: torch.Tensor, image1: torch.Tensor, warmup: int = 3):
        self.pipe = pipe
        self.image0, self.image1 = image0.clone(), image1.clone()
        side = torch.cuda.Stream(device=image0.device)
        side.wait_stream(torch.cuda.current_stream(image0.device))
        with torch.cuda.stream(side):  # warm-up off the capture: workspaces reach their final size
            for _ in range(warmup):
                pipe(self.image0, self.image1)
        torch.cuda.current_stream(image0.device).wait_stream(side)
        torch.cuda.synchronize(image0.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = pipe(self.image0, self.image1)

    @torch.no_grad()
    def __call__(self, image0: torch.Tensor, image1: torch.Tensor) -> dict:
        if image0.shape != self.image0.shape or image1.shape != self.image1.shape:
            raise ValueError(f"graph was captured for {tuple(self.image0.shape)} / {tuple(self.image1.shape)}")
        self.image0.copy_(image0)
        self.image1.copy_(image1)
        self.graph.replay()
        return self.out


class GraphedCall:
    """A function of fixed-shape device tensors captured in a HIP graph: `GraphedCall(fn, *example_inputs)` warms `fn` up, captures ONE call on
    static copies of the inputs and replays it on every `__call__` (inputs are copied into the static buffers first).  Used by the plugins'
    opt-in `conf["hip_graph"]` (one image / one pair per `_forward`: the reference's call pattern is launch-bound, ~250 launches of a few
    microseconds).  `fn` must not synchronise with the host and must return a dict of tensors; the returned tensors are the graph's static
    outputs -- the caller slices and CLONES what it keeps."""

    def __init__(self, fn, *inputs: torch.Tensor, warmup: int = 2):
        dev = inputs[0].device
        self.fn = fn
        self.static_in = [t.clone() for t in inputs]
        # The graph's OWN stream, for the warm-up and for the capture: the C ABI's scratch is one buffer per (device, stream)
        # (`backend._Workspace`) and a buffer handed out during a capture is pinned, so every graph pins a workspace of its own, sized by its
        # own warm-up -- a later graph of a larger capacity (LightGlue 2048 key-points after 1024, a bigger SuperPoint image) no longer asks
        # a pinned buffer to grow (torch.cuda.graph's default capture stream is shared by all graphs of the process), and the warm-up leaves no
        # second buffer behind under a throw-away stream's key.  ADVICE round 5.
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(self.stream):  # warm-up off the capture: workspaces reach their final size
            for _ in range(warmup):
                fn(*self.static_in)
        torch.cuda.current_stream(dev).wait_stream(self.stream)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.out = fn(*self.static_in)

    @torch.no_grad()
    def __call__(self, *inputs: torch.Tensor) -> dict:
        for s, t in zip(self.static_in, inputs):
            s.copy_(t)
        self.graph.replay()
        return self.out


def match_table(out: dict) -> torch.Tensor:
    """Fixed-stride per-pair record for the multi-GPU all-gather:
    int32 [B, 3 + 2*K]: n0, n1, stop, matches0[K], bit-cast matching_scores0[K]."""
    B, K = out["matches0"].shape
    stop = out["stop"] if "stop" in out else torch.zeros_like(out["num_keypoints0"])  # SuperGlue has no early exit
    head = torch.stack([out["num_keypoints0"], out["num_keypoints1"], stop], 1).to(torch.int32)
    return torch.cat([head, out["matches0"].to(torch.int32), out["matching_scores0"].contiguous().view(torch.int32)], 1)
