"""EfficientLoFTR dense matcher plugin on the MI355X HIP backend.

Drop-in for imcui/hloc/matchers/eloftr.py: module name `eloftr`, same `default_conf` (:25-34) and `required_inputs`
(:35); `_forward` keeps the wrapper's semantics -- image0 <-> image1 are swapped before the net ("we refine kpts in
image0", :69-78), the top-k matches by confidence are kept with `argsort(descending)[:k]` (:90-97), key names are swapped
back (:100) and `mconf` becomes `scores`.  The model itself (the 'full' EfficientLoFTR, fp32, after `reparameter()`,
:56-61) runs in libimcui_hip (imcui_hip_eloftr_forward).

Weights: conf["state_dict"] / conf["weights_path"], in either naming:
  * the UPSTREAM names of `eloftr_outdoor.ckpt`, the file the reference downloads and loads with
    `torch.load(...)["state_dict"]` -> `ELoFTR_(config).load_state_dict` -> `reparameter()` (eloftr.py:56-61):
    `matcher.backbone.layer{0..3}[.{block}].{rbr_dense,rbr_1x1}.{conv,bn}`, `.rbr_identity`, `matcher.loftr_coarse.layers.{0..7}.
    {aggregate,norm1,q_proj,k_proj,v_proj,merge,mlp.0,mlp.2,norm2}`, `matcher.fine_preprocess.layer{3,2,1}_outconv[2.{0,1,3}]`
    (with or without the Lightning `matcher.` prefix).  `upstream_to_port_names` maps them onto the port's names; the map is
    total and one-to-one on the 447 tensors of the model (checked at load: every port key produced exactly once, no source key
    left over, shapes confirmed by the packer against `imcui_hip_eloftr_layer_shape`), and `port_to_upstream_names` is its
    inverse (round-trip test: tests/test_host_cpu.py).  Both trees use the interleaved rotary pairs (upstream
    `x[..., ::2], x[..., 1::2]` = the port's Cohere-style `rotate_half`), so no projection rows are permuted;
  * the names of the maintained port (`transformers.EfficientLoFTRForKeypointMatching.state_dict()`).

Image pairs whose two images differ in size are accepted (the batch path of `match_dense.py` produces them).

`precision` (eloftr.py:32-33,43-47,63-64): "fp32" = the parity arithmetic (3 x f16 split products, fp32-grade); "fp16" (the
reference's `self.net.half()`) and "mp" (its autocast flag) both select ONE f16 product per element pair, f32 accumulate, in the
backbone and fine-fusion convolutions -- 11-bit operands where the reference's half run carries 11 and an autocast-bf16 run 8 --
while the transformer, the similarity, the matching and the fine stages stay in the split arithmetic
(`imcui_hip_eloftr_forward_ex`, arith 1; tests/test_gpu_eloftr.py compares it with the fp32 oracle).
`model_type: "opt"` (eloftr.py:38-41: upstream's `opt_default_cfg`, a different coarse-matching rule without the dual soft-max)
is refused: its semantics cannot be confirmed offline (the EfficientLoFTR sources are an un-vendored submodule).
"""
from __future__ import annotations

import re

import torch

from ... import backend
from ..utils.base_model import BaseModel
from ..utils.weights import resolve_state_dict

_BB = "efficientloftr.backbone.stages."
_TR = "efficientloftr.local_feature_transformer.layers."
_RF = "refinement_layer."
# upstream module tree -> port module tree (prefix rewrites; the tensor suffix .weight / .bias / .running_* is kept)
_LAYER_PARTS = {"aggregate": "aggregation.q_aggregation", "norm1": "aggregation.norm", "q_proj": "attention.q_proj", "k_proj": "attention.k_proj",
                "v_proj": "attention.v_proj", "merge": "attention.o_proj", "mlp.0": "mlp.fc1", "mlp.2": "mlp.fc2", "norm2": "mlp.layer_norm"}
_RBR = {"rbr_dense.conv": "conv1.conv", "rbr_dense.bn": "conv1.norm", "rbr_1x1.conv": "conv2.conv", "rbr_1x1.bn": "conv2.norm", "rbr_identity": "identity"}
_FINE = {"layer3_outconv": "out_conv", "layer2_outconv": "out_conv_layers.0.out_conv1", "layer2_outconv2.0": "out_conv_layers.0.out_conv2",
         "layer2_outconv2.1": "out_conv_layers.0.batch_norm", "layer2_outconv2.3": "out_conv_layers.0.out_conv3",
         "layer1_outconv": "out_conv_layers.1.out_conv1", "layer1_outconv2.0": "out_conv_layers.1.out_conv2",
         "layer1_outconv2.1": "out_conv_layers.1.batch_norm", "layer1_outconv2.3": "out_conv_layers.1.out_conv3"}  # fmt: skip


def _upstream_key(k: str) -> str | None:
    """One upstream parameter name -> the port's name (None: a key the network does not own, e.g. a loss buffer)."""
    k = k[len("matcher."):] if k.startswith("matcher.") else k
    m = re.fullmatch(r"backbone\.layer(\d+)\.(?:(\d+)\.)?(rbr_dense\.conv|rbr_dense\.bn|rbr_1x1\.conv|rbr_1x1\.bn|rbr_identity)\.(.+)", k)
    if m:
        stage, block, part, leaf = m.groups()
        if (stage == "0") != (block is None):
            return None  # layer0 is a single block, layers 1-3 are sequences
        return f"{_BB}{stage}.blocks.{block or 0}.{_RBR[part]}.{leaf}"
    m = re.fullmatch(r"loftr_coarse\.layers\.(\d+)\.(aggregate|norm1|q_proj|k_proj|v_proj|merge|mlp\.0|mlp\.2|norm2)\.(.+)", k)
    if m:
        i, part, leaf = int(m.group(1)), m.group(2), m.group(3)
        return f"{_TR}{i // 2}.{'self_attention' if i % 2 == 0 else 'cross_attention'}.{_LAYER_PARTS[part]}.{leaf}"  # layer_names = [self, cross] x 4
    m = re.fullmatch(r"fine_preprocess\.(layer3_outconv|layer[12]_outconv2\.[013]|layer[12]_outconv)\.(.+)", k)
    if m:
        return f"{_RF}{_FINE[m.group(1)]}.{m.group(2)}"
    return None


def upstream_to_port_names(sd: dict) -> dict:
    """State dict of upstream `EfficientLoFTR/src/loftr.LoFTR` (what `eloftr_outdoor.ckpt["state_dict"]` holds) under the
    parameter names the HIP packer reads.  Raises on any tensor of the three sub-networks it cannot place and on collisions."""
    out, unknown = {}, []
    for k, v in sd.items():
        nk = _upstream_key(k)
        if nk is None:
            kk = k[len("matcher."):] if k.startswith("matcher.") else k
            if kk.startswith(("backbone.", "loftr_coarse.", "fine_preprocess.")):
                unknown.append(k)
            continue  # anything else (position-encoding buffers, training-only heads) carries no inference weights
        if nk in out:
            raise KeyError(f"EfficientLoFTR checkpoint: {k!r} maps onto {nk!r} twice")
        out[nk] = v
    if unknown:
        raise KeyError(f"EfficientLoFTR checkpoint: unrecognised tensors of the network, e.g. {unknown[:3]}")
    return out


def port_to_upstream_names(sd: dict, prefix: str = "matcher.") -> dict:
    """Inverse of `upstream_to_port_names` (used by the round-trip test and to export weights for the reference wrapper)."""
    inv_rbr = {v: k for k, v in _RBR.items()}
    inv_parts = {v: k for k, v in _LAYER_PARTS.items()}
    inv_fine = {v: k for k, v in _FINE.items()}
    out = {}
    for k, v in sd.items():
        m = re.fullmatch(re.escape(_BB) + r"(\d+)\.blocks\.(\d+)\.(conv1\.conv|conv1\.norm|conv2\.conv|conv2\.norm|identity)\.(.+)", k)
        if m:
            stage, block, part, leaf = m.groups()
            mid = "" if stage == "0" else f"{block}."
            out[f"{prefix}backbone.layer{stage}.{mid}{inv_rbr[part]}.{leaf}"] = v
            continue
        m = re.fullmatch(re.escape(_TR) + r"(\d+)\.(self_attention|cross_attention)\.(.+)\.(weight|bias)", k)
        if m:
            i = 2 * int(m.group(1)) + (m.group(2) == "cross_attention")
            out[f"{prefix}loftr_coarse.layers.{i}.{inv_parts[m.group(3)]}.{m.group(4)}"] = v
            continue
        m = re.fullmatch(re.escape(_RF) + r"(out_conv_layers\.\d\.(?:out_conv[123]|batch_norm)|out_conv)\.(.+)", k)
        if m:
            out[f"{prefix}fine_preprocess.{inv_fine[m.group(1)]}.{m.group(2)}"] = v
            continue
        raise KeyError(f"not a parameter of the EfficientLoFTR port: {k!r}")
    return out


def to_port_names(sd: dict) -> dict:
    """Accept either naming; refuse anything else loudly (a silent mis-assignment would produce plausible but wrong matches)."""
    if any(k.startswith("efficientloftr.") for k in sd) and any(k.startswith("refinement_layer.") for k in sd):
        return sd
    if any(re.match(r"(matcher\.)?backbone\.layer0\.rbr_dense\.conv\.weight$", k) for k in sd):
        return upstream_to_port_names(sd)
    raise KeyError(
        "EfficientLoFTR weights must use the upstream names of eloftr_outdoor.ckpt (matcher.backbone.layer0.rbr_dense...) or those of "
        f"transformers.EfficientLoFTRForKeypointMatching (got keys like {next(iter(sd))!r})"
    )


class ELoFTR(BaseModel):
    default_conf = {
        "model_name": "eloftr_outdoor.ckpt",
        "match_threshold": 0.2,
        "max_keypoints": -1,
        "model_type": "full",
        "precision": "fp32",
    }
    # The reference writes match_threshold into upstream's config when the model is BUILT (`cfg["match_coarse"]["thr"] = conf["match_threshold"]`,
    # imcui/hloc/matchers/eloftr.py:51-52), and picks the arithmetic there too (:43-47,63-64): the UI's later mutation of a cached model's
    # conf (imcui/ui/utils.py:921-922) reaches neither.  False (default) = exactly that; True = re-read conf["match_threshold"] on every
    # call (what the slider intends).  Not a reference key: read with conf.get("runtime_match_threshold", False), default_conf stays the reference's.
    required_inputs = ["image0", "image1"]

    def _init(self, conf):
        if conf.get("model_type", "full") != "full":
            raise NotImplementedError("the HIP EfficientLoFTR computes the 'full' model (model_type 'opt' changes the coarse matching rule; not restated)")
        if conf.get("precision", "fp32") not in ("fp32", "fp16", "mp"):
            raise ValueError(f"precision {conf.get('precision')!r}: one of 'fp32', 'mp', 'fp16' (imcui/hloc/matchers/eloftr.py:32-33)")
        self._match_threshold = float(conf["match_threshold"])  # frozen here, like `cfg["match_coarse"]["thr"]` (see default_conf)
        self._arith = 0 if conf.get("precision", "fp32") == "fp32" else 1  # `_default_cfg["mp"]` / `.half()` are applied once, in `_init`
        sd = resolve_state_dict(conf, "eloftr")
        if "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
        sd = to_port_names(sd)
        self.conf.pop("state_dict", None)
        self.register_buffer("packed", backend.pack_eloftr(sd), persistent=False)
        self._impl = backend.ELoFTRHIP()

    def forward_batched(self, image0: torch.Tensor, image1: torch.Tensor, debug_windows: bool = False) -> dict:
        """Upstream forward(image0, image1) on a batch: fixed-capacity outputs, no host sync."""
        return self._impl.forward(self.packed, image0, image1, self.match_threshold(), debug_windows, self._arith)

    def match_threshold(self) -> float:
        """The coarse-matching threshold of this call: the value the model was built with, or conf's current one under the opt-in."""
        c = self.conf
        return float(c["match_threshold"]) if c.get("runtime_match_threshold", False) else self._match_threshold

    def forward_pairs(self, image0: torch.Tensor, image1: torch.Tensor) -> list:
        """`_forward` on B pairs at once (the batched dense driver): the per-pair dictionaries the wrapper would return for
        `{"image0": image0[b:b+1], "image1": image1[b:b+1]}` -- images exchanged before the net, per-pair top-k by confidence,
        key names exchanged back.  One device-to-host read (the match count)."""
        out = self.forward_batched(image1, image0)
        n = int(out["num_matches"][0])
        bidx = out["batch_indexes"][:n]
        kp0, kp1, conf = out["keypoints0"][:n], out["keypoints1"][:n], out["confidence"][:n]
        top_k = self.conf["max_keypoints"]
        res = []
        for b in range(image0.shape[0]):
            sel = (bidx == b).nonzero()[:, 0]
            k0, k1, sc = kp0[sel], kp1[sel], conf[sel]
            if top_k is not None and len(sc) > top_k:
                keep = torch.argsort(sc, descending=True)[:top_k]
                k0, k1, sc = k0[keep], k1[keep], sc[keep]
            res.append({"keypoints0": k1, "keypoints1": k0, "scores": sc})
        return res

    @staticmethod
    def _refuse_masks(data):
        """The reference renames `mask0` / `mask1` and hands them to the net (eloftr.py:70-78); upstream uses them on padded training
        batches (coarse attention and `sim.masked_fill_(~mask, -inf)`).  No caller of the reference produces them for a
        single pair (`match_dense.ImagePairDataset`, `match_images`): the HIP path has no masked kernels and refuses the keys
        instead of silently ignoring them."""
        for k in ("mask0", "mask1"):
            if data.get(k) is not None:
                raise NotImplementedError(f"{k}: padded-batch masks are not supported by the HIP dense matchers (pass un-padded images)")

    def _forward(self, data):
        self._refuse_masks(data)
        out = self.forward_batched(data["image1"], data["image0"])  # the reference refines key-points in image0
        n = int(out["num_matches"][0])
        kp0, kp1, scores = out["keypoints0"][:n], out["keypoints1"][:n], out["confidence"][:n]
        top_k = self.conf["max_keypoints"]
        if top_k is not None and len(scores) > top_k:
            keep = torch.argsort(scores, descending=True)[:top_k]
            kp0, kp1, scores = kp0[keep], kp1[keep], scores[keep]
        return {"keypoints0": kp1, "keypoints1": kp0, "scores": scores}
