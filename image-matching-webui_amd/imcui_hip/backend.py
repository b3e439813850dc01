"""Thin Python host layer over the C ABI: handles, weight packing, workspaces, launches.

PyTorch-ROCm tensors in, tensors out.  Everything is enqueued on
``torch.cuda.current_stream()``; nothing here synchronises except where the reference's
ragged list outputs force a device->host read of the key-point counts.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from .lib_loader import ImcuiHipError, load_library

SP_ORDER = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convPb", "convDa", "convDb"]

_handles: dict[int, "Handle"] = {}
_hlock = threading.Lock()


class Handle:
    def __init__(self, device_index: int):
        self.lib = load_library()
        self.device_index = device_index
        h = C.c_void_p()
        rc = self.lib.imcui_hip_create(device_index, C.byref(h))
        if rc != 0 or not h:
            raise ImcuiHipError(f"imcui_hip_create(device={device_index}) failed with {rc} (no usable HIP device?)")
        self.h = h

    def check(self, rc: int, what: str = ""):
        if rc != 0:
            msg = self.lib.imcui_hip_last_error(self.h)
            raise ImcuiHipError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def get_handle(device: torch.device) -> Handle:
    if device.type != "cuda":
        raise ImcuiHipError(f"the HIP backend needs a ROCm device tensor, got device '{device}' (no CPU fallback)")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with _hlock:
        if idx not in _handles:
            _handles[idx] = Handle(idx)
        return _handles[idx]


def _stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: torch.Tensor | None) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _as_f32_host(t) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        t = t.detach().to("cpu", torch.float32).contiguous().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


class _Workspace:
    """Caller-provided scratch of the C ABI: one grow-only byte buffer per (device, stream).

    Kernels of one stream execute in order, so a buffer is only ever shared by launches that are ordered
    anyway; two streams (two Gradio worker threads, a side stream) get separate buffers and cannot overwrite
    each other's scratch while kernels are in flight.  A buffer that was handed out during a HIP-graph capture
    is baked into that graph: it is pinned, and a later request that would have to re-allocate it raises
    instead of freeing memory the graph still replays on."""

    def __init__(self):
        self._bufs: dict[tuple[int, int], torch.Tensor] = {}
        self._pinned: set[tuple[int, int]] = set()

    def get(self, nbytes: int, device: torch.device) -> torch.Tensor:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, torch.cuda.current_stream(device).cuda_stream)
        capturing = torch.cuda.is_current_stream_capturing()
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            if key in self._pinned:
                raise ImcuiHipError(
                    f"workspace of stream {key[1]:#x} is referenced by a captured HIP graph ({buf.numel()} B) and cannot grow to "
                    f"{int(nbytes)} B: capture the graph after a warm-up at the largest shape, or run the larger call on another stream"
                )
            buf = None
            self._bufs.pop(key, None)
            buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        if capturing:
            self._pinned.add(key)
        return buf


# ------------------------------------------------------------------ SuperPoint
def pack_superpoint(state_dict: dict) -> torch.Tensor:
    """Upstream SuperPoint state dict -> packed float32 buffer (host)."""
    lib = load_library()
    ws = [_as_f32_host(state_dict[f"{n}.weight"]) for n in SP_ORDER]
    bs = [_as_f32_host(state_dict[f"{n}.bias"]) for n in SP_ORDER]
    expect = {"conv1a": (64, 1, 3, 3), "convPb": (65, 256, 1, 1), "convDb": (256, 256, 1, 1)}
    for n, shp in expect.items():
        if tuple(state_dict[f"{n}.weight"].shape) != shp:
            raise ImcuiHipError(f"unexpected shape for {n}.weight: {tuple(state_dict[f'{n}.weight'].shape)}")
    packed = np.zeros(lib.imcui_hip_superpoint_packed_floats(), dtype=np.float32)
    wp = (C.c_void_p * 12)(*[w.ctypes.data for w in ws])
    bp = (C.c_void_p * 12)(*[b.ctypes.data for b in bs])
    rc = lib.imcui_hip_superpoint_pack_weights(wp, bp, packed.ctypes.data)
    if rc != 0:
        raise ImcuiHipError(f"imcui_hip_superpoint_pack_weights failed ({rc})")
    return torch.from_numpy(packed)


SP_TOPK_MAX = 16384  # on-chip top-k sorter of sp_topk_kernel (csrc/superpoint.hip)


class SuperPointHIP:
    def __init__(self):
        self._ws = _Workspace()
        self._lock = threading.Lock()

    def forward(self, packed: torch.Tensor, image: torch.Tensor, conf: dict, want_score_map: bool = False, kcap: int | None = None):
        """image [B,1,H,W] float32 on the GPU.  Returns dict of fixed-stride tensors + counts.

        Never synchronises: `status` [1] int32 is the selection status word on the device (0 = fine; bit 1 =
        `kcap` was too small, which only max_keypoints = -1 with exactly tied scores can cause) -- callers that
        read `num_keypoints` on the host anyway (the ragged plugin path) read it in the same copy."""
        hd = get_handle(image.device)
        if packed.device != image.device:
            raise ImcuiHipError("packed weights and image live on different devices")
        lib = hd.lib
        image = image.contiguous().float()
        B, Cc, H, W = image.shape
        if Cc != 1:
            raise ImcuiHipError(f"SuperPoint expects a 1-channel image, got {Cc}")
        nms = int(conf["nms_radius"])
        maxk = int(conf["max_keypoints"])
        if maxk > SP_TOPK_MAX and H * W > SP_TOPK_MAX:
            raise ImcuiHipError(f"max_keypoints={maxk} exceeds the on-chip top-k sorter ({SP_TOPK_MAX}); use -1 to keep every key-point")
        if kcap is None:
            kcap = lib.imcui_hip_superpoint_max_keypoints_bound(H, W, nms) if maxk < 0 else max(1, min(maxk, H * W))
        dev = image.device
        kpts = torch.empty((B, kcap, 2), dtype=torch.float32, device=dev)
        scores = torch.empty((B, kcap), dtype=torch.float32, device=dev)
        desc = torch.empty((B, kcap, 256), dtype=torch.float32, device=dev)
        nk = torch.empty((B,), dtype=torch.int32, device=dev)
        status = torch.empty((1,), dtype=torch.int32, device=dev)
        smap = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_score_map else None
        with self._lock:
            ws = self._ws.get(lib.imcui_hip_superpoint_workspace_bytes(B, H, W, nms), dev)
            with torch.cuda.device(dev):
                rc = lib.imcui_hip_superpoint_forward(
                    hd.h, _ptr(packed), _ptr(image), B, H, W, nms, float(conf["keypoint_threshold"]),
                    int(conf["remove_borders"]), maxk, int(bool(conf.get("fix_sampling", False))), kcap,
                    _ptr(kpts), _ptr(scores), _ptr(desc), _ptr(nk), _ptr(status), _ptr(smap), _ptr(ws), ws.numel(), _stream_ptr(),
                )  # fmt: skip
                hd.check(rc, "imcui_hip_superpoint_forward")
        out = {"keypoints": kpts, "scores": scores, "descriptors": desc, "num_keypoints": nk, "status": status}
        if want_score_map:
            out["score_map"] = smap
        return out


# ------------------------------------------------------------------ LightGlue
def lightglue_tensor_names() -> list[str]:
    lib = load_library()
    return [lib.imcui_hip_lightglue_tensor_name(i).decode() for i in range(lib.imcui_hip_lightglue_num_tensors())]


def _rename_old_lightglue_keys(sd: dict) -> dict:
    """Old checkpoints name blocks `self_attn.{i}...` / `cross_attn.{i}...` (renamed on load upstream)."""
    out = {}
    for k, v in sd.items():
        for i in range(9):
            for blk in ("self_attn", "cross_attn"):
                old = f"{blk}.{i}"
                if k.startswith(old + "."):
                    k = f"transformers.{i}.{blk}" + k[len(old):]
        out[k] = v
    return out


def lightglue_variant(state_dict: dict) -> tuple[int, bool]:
    """(input_dim, add_scale_ori) of a LightGlue state dict: upstream's `features` table in the weights themselves --
    `input_proj` exists when the descriptors are not 256-d (disk / aliked / sift: 128), `posenc.Wr` has 4 input columns
    when key-point scale and orientation are encoded (sift, doghardnet)."""
    sd = _rename_old_lightglue_keys(state_dict)
    input_dim = int(sd["input_proj.weight"].shape[1]) if "input_proj.weight" in sd else 256
    return input_dim, int(sd["posenc.Wr.weight"].shape[1]) == 4


def pack_lightglue(state_dict: dict) -> torch.Tensor:
    lib = load_library()
    sd = _rename_old_lightglue_keys(state_dict)
    names = lightglue_tensor_names()
    arrs = []
    for n in names:
        if n not in sd:
            raise ImcuiHipError(f"LightGlue state dict lacks '{n}'")
        arrs.append(_as_f32_host(sd[n]))
    input_dim, scale_ori = lightglue_variant(sd)
    if arrs[0].shape not in ((32, 2), (32, 4)) or arrs[1].shape != (768, 256) or input_dim % 32 or input_dim > 256:
        raise ImcuiHipError("only the 256-d / 4-head / 9-layer LightGlue (descriptor input 32..256, optional scale / orientation) is supported")
    packed = np.zeros(lib.imcui_hip_lightglue_packed_floats(), dtype=np.float32)
    wr = arrs[0]
    arrs[0] = np.ascontiguousarray(wr[:, :2])  # the base packer takes the (x, y) columns; pack_input stores all of them
    tp = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    rc = lib.imcui_hip_lightglue_pack_weights(tp, packed.ctypes.data)
    if rc != 0:
        raise ImcuiHipError(f"imcui_hip_lightglue_pack_weights failed ({rc})")
    wi = _as_f32_host(sd["input_proj.weight"]) if input_dim != 256 else None
    bi = _as_f32_host(sd["input_proj.bias"]) if input_dim != 256 else None
    rc = lib.imcui_hip_lightglue_pack_input(wr.ctypes.data, wr.shape[1], None if wi is None else wi.ctypes.data,
                                            None if bi is None else bi.ctypes.data, input_dim, packed.ctypes.data)  # fmt: skip
    if rc != 0:
        raise ImcuiHipError(f"imcui_hip_lightglue_pack_input failed ({rc})")
    return torch.from_numpy(packed)


class LightGlueHIP:
    def __init__(self):
        self._ws = _Workspace()
        self._lock = threading.Lock()

    def forward(self, packed, kpts0, kpts1, desc0, desc1, n0, n1, size0, size1, depth_confidence, width_confidence,
                filter_threshold, pruning_threshold: int = -1, layer_dump: bool = False, scales_oris=None):  # fmt: skip
        """kptsX [B,ncap,2], descX [B,ncap,D] (row per point; D = the weights' input_dim), nX [B] int32 on the GPU;
        sizeX = (W, H); scales_oris = (scales0, oris0, scales1, oris1) [B,ncap] for add_scale_ori weights.
        pruning_threshold: upstream pruning_keypoint_thresholds[device] (-1 = the CPU path: always prune).
        layer_dump (parity tests): also return `_layers` [9, 2B, R, 256], the token states after every layer."""
        dev = kpts0.device
        hd = get_handle(dev)
        lib = hd.lib
        B, ncap0 = kpts0.shape[0], kpts0.shape[1]
        ncap1 = kpts1.shape[1]
        ncap = max(ncap0, ncap1, 1)

        def pad(t, n):
            if t.shape[1] == n:
                return t.contiguous().float()
            shape = list(t.shape)
            shape[1] = n
            o = torch.zeros(shape, dtype=torch.float32, device=dev)
            o[:, : t.shape[1]] = t
            return o

        kpts0, kpts1, desc0, desc1 = pad(kpts0, ncap), pad(kpts1, ncap), pad(desc0, ncap), pad(desc1, ncap)
        input_dim = desc0.shape[2]
        so = [None] * 4 if scales_oris is None else [pad(t, ncap) for t in scales_oris]
        n0 = n0.to(device=dev, dtype=torch.int32).contiguous()
        n1 = n1.to(device=dev, dtype=torch.int32).contiguous()
        m0 = torch.empty((B, ncap), dtype=torch.int32, device=dev)
        m1 = torch.empty((B, ncap), dtype=torch.int32, device=dev)
        s0 = torch.empty((B, ncap), dtype=torch.float32, device=dev)
        s1 = torch.empty((B, ncap), dtype=torch.float32, device=dev)
        stop = torch.empty((B,), dtype=torch.int32, device=dev)
        p0 = torch.empty((B, ncap), dtype=torch.int32, device=dev)
        p1 = torch.empty((B, ncap), dtype=torch.int32, device=dev)
        dump = None
        with self._lock:
            ws = self._ws.get(lib.imcui_hip_lightglue_workspace_bytes(B, ncap), dev)
            with torch.cuda.device(dev):
                if layer_dump:
                    R = (ncap + 127) // 128 * 128
                    dump = torch.zeros((9, 2 * B, R, 256), dtype=torch.float32, device=dev)
                    hd.check(lib.imcui_hip_lightglue_set_layer_dump(hd.h, _ptr(dump), dump.numel()), "set_layer_dump")
                try:
                    rc = lib.imcui_hip_lightglue_forward(
                        hd.h, _ptr(packed), B, ncap, input_dim, _ptr(kpts0), _ptr(kpts1), _ptr(desc0), _ptr(desc1),
                        _ptr(so[0]), _ptr(so[1]), _ptr(so[2]), _ptr(so[3]), _ptr(n0), _ptr(n1),
                        float(size0[0]), float(size0[1]), float(size1[0]), float(size1[1]),
                        float(depth_confidence), float(width_confidence), int(pruning_threshold), float(filter_threshold),
                        _ptr(m0), _ptr(m1), _ptr(s0), _ptr(s1), _ptr(stop), _ptr(p0), _ptr(p1), _ptr(ws), ws.numel(), _stream_ptr(),
                    )  # fmt: skip
                finally:
                    if layer_dump:
                        lib.imcui_hip_lightglue_set_layer_dump(hd.h, None, 0)
                hd.check(rc, "imcui_hip_lightglue_forward")
        out = {
            "matches0": m0[:, :ncap0], "matches1": m1[:, :ncap1], "matching_scores0": s0[:, :ncap0],
            "matching_scores1": s1[:, :ncap1], "stop": stop, "prune0": p0[:, :ncap0], "prune1": p1[:, :ncap1],
        }  # fmt: skip
        if dump is not None:
            out["_layers"] = dump
        return out


# ------------------------------------------------------------------ SuperGlue
def superglue_tensor_names() -> list[str]:
    lib = load_library()
    return [lib.imcui_hip_superglue_tensor_name(i).decode() for i in range(lib.imcui_hip_superglue_num_tensors())]


def pack_superglue(state_dict: dict) -> torch.Tensor:
    """Upstream SuperGlue state dict (Conv1d weights [out,in,1]) -> packed float32 buffer (host)."""
    lib = load_library()
    names = superglue_tensor_names()
    arrs = []
    for n in names:
        if n not in state_dict:
            raise ImcuiHipError(f"SuperGlue state dict lacks '{n}'")
        arrs.append(_as_f32_host(state_dict[n]).reshape(-1) if n == "bin_score" else _as_f32_host(state_dict[n]))
    if arrs[0].size != 32 * 3 or arrs[names.index("gnn.layers.17.mlp.3.weight")].size != 256 * 512:
        raise ImcuiHipError("only the 256-d / 4-head / 18-layer SuperGlue with the [32,64,128,256] key-point encoder is supported")
    packed = np.zeros(lib.imcui_hip_superglue_packed_floats(), dtype=np.float32)
    tp = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    rc = lib.imcui_hip_superglue_pack_weights(tp, packed.ctypes.data)
    if rc != 0:
        raise ImcuiHipError(f"imcui_hip_superglue_pack_weights failed ({rc})")
    return torch.from_numpy(packed)


class SuperGlueHIP:
    def __init__(self):
        self._ws = _Workspace()
        self._lock = threading.Lock()

    def forward(self, packed, kpts0, kpts1, scores0, scores1, desc0, desc1, n0, n1, size0, size1, sinkhorn_iterations,
                match_threshold):  # fmt: skip
        """kptsX [B,ncap,2], scoresX [B,ncap], descX [B,ncap,256] (row per point), nX [B] int32 on the GPU; sizeX = (W, H)."""
        dev = kpts0.device
        hd = get_handle(dev)
        lib = hd.lib
        B, ncap0 = kpts0.shape[0], kpts0.shape[1]
        ncap1 = kpts1.shape[1]
        ncap = max(ncap0, ncap1, 1)

        def pad(t, n):
            if t.shape[1] == n:
                return t.contiguous().float()
            shape = list(t.shape)
            shape[1] = n
            o = torch.zeros(shape, dtype=torch.float32, device=dev)
            o[:, : t.shape[1]] = t
            return o

        kpts0, kpts1, desc0, desc1 = pad(kpts0, ncap), pad(kpts1, ncap), pad(desc0, ncap), pad(desc1, ncap)
        scores0, scores1 = pad(scores0, ncap), pad(scores1, ncap)
        n0 = n0.to(device=dev, dtype=torch.int32).contiguous()
        n1 = n1.to(device=dev, dtype=torch.int32).contiguous()
        m0 = torch.empty((B, ncap), dtype=torch.int32, device=dev)
        m1 = torch.empty((B, ncap), dtype=torch.int32, device=dev)
        s0 = torch.empty((B, ncap), dtype=torch.float32, device=dev)
        s1 = torch.empty((B, ncap), dtype=torch.float32, device=dev)
        with self._lock:
            ws = self._ws.get(lib.imcui_hip_superglue_workspace_bytes(B, ncap), dev)
            with torch.cuda.device(dev):
                rc = lib.imcui_hip_superglue_forward(
                    hd.h, _ptr(packed), B, ncap, _ptr(kpts0), _ptr(kpts1), _ptr(scores0), _ptr(scores1), _ptr(desc0), _ptr(desc1),
                    _ptr(n0), _ptr(n1), float(size0[0]), float(size0[1]), float(size1[0]), float(size1[1]),
                    int(sinkhorn_iterations), float(match_threshold),
                    _ptr(m0), _ptr(m1), _ptr(s0), _ptr(s1), _ptr(ws), ws.numel(), _stream_ptr(),
                )  # fmt: skip
                hd.check(rc, "imcui_hip_superglue_forward")
        return {
            "matches0": m0[:, :ncap0], "matches1": m1[:, :ncap1], "matching_scores0": s0[:, :ncap0],
            "matching_scores1": s1[:, :ncap1],
        }  # fmt: skip


# ------------------------------------------------------------------ LoFTR
def _fold_bn(w: torch.Tensor, sd: dict, bn: str | None, eps: float = 1e-5):
    """conv (no bias) followed by eval-mode BatchNorm -> (w', b')."""
    cout = w.shape[0]
    if bn is None:
        return w, torch.zeros(cout)
    g = sd[bn + ".weight"].float() / torch.sqrt(sd[bn + ".running_var"].float() + eps)
    return w * g.view(-1, 1, 1, 1), sd[bn + ".bias"].float() - sd[bn + ".running_mean"].float() * g


def _conv_gemm_layout(w: torch.Tensor, b: torch.Tensor, n_pad: int, cin_pad: int):
    """OIHW -> [Cout_pad][tap][Cin_pad] (K order of the implicit im2col), zero padded."""
    cout, cin, kh, kw = w.shape
    out = torch.zeros(n_pad, kh * kw, cin_pad)
    out[:cout, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    bo = torch.zeros(n_pad)
    bo[:cout] = b
    return out.reshape(n_pad, kh * kw * cin_pad).contiguous(), bo


def pack_loftr(state_dict: dict) -> torch.Tensor:
    """kornia LoFTR state dict -> packed float32 buffer (host): BN folding, GEMM layouts, 196 -> 256 padding."""
    lib = load_library()
    sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if isinstance(v, torch.Tensor)}
    bb = "backbone."
    w1, b1 = _fold_bn(sd[bb + "conv1.weight"], sd, bb + "bn1")
    conv1_w = w1[:, 0].reshape(128, 49).t().contiguous().numpy()  # [tap][cout]
    conv1_b = b1.contiguous().numpy()
    P = lambda c: 256 if c == 196 else c  # noqa: E731
    convs = [  # (weight key, bn key or None) in the order of the C layer table
        ("layer1.0.conv1", "layer1.0.bn1"), ("layer1.0.conv2", "layer1.0.bn2"), ("layer1.1.conv1", "layer1.1.bn1"), ("layer1.1.conv2", "layer1.1.bn2"),
        ("layer2.0.conv1", "layer2.0.bn1"), ("layer2.0.conv2", "layer2.0.bn2"), ("layer2.0.downsample.0", "layer2.0.downsample.1"),
        ("layer2.1.conv1", "layer2.1.bn1"), ("layer2.1.conv2", "layer2.1.bn2"),
        ("layer3.0.conv1", "layer3.0.bn1"), ("layer3.0.conv2", "layer3.0.bn2"), ("layer3.0.downsample.0", "layer3.0.downsample.1"),
        ("layer3.1.conv1", "layer3.1.bn1"), ("layer3.1.conv2", "layer3.1.bn2"),
        ("layer3_outconv", None), ("layer2_outconv", None), ("layer2_outconv2.0", "layer2_outconv2.1"), ("layer2_outconv2.3", None),
        ("layer1_outconv", None), ("layer1_outconv2.0", "layer1_outconv2.1"), ("layer1_outconv2.3", None),
    ]  # fmt: skip
    ws, bs = [], []
    for wk, bk in convs:
        w, b = _fold_bn(sd[bb + wk + ".weight"], sd, bb + bk if bk else None)
        wg, bg = _conv_gemm_layout(w, b, P(w.shape[0]), P(w.shape[1]))
        ws.append(wg)
        bs.append(bg)
    lin_names = ["q_proj", "k_proj", "v_proj", "merge", "mlp.0", "mlp.2"]
    for i in range(8):
        for n in lin_names:
            ws.append(sd[f"loftr_coarse.layers.{i}.{n}.weight"].contiguous())
            bs.append(None)
    for n in ("down_proj", "merge_feat"):
        ws.append(sd[f"fine_preprocess.{n}.weight"].contiguous())
        bs.append(sd[f"fine_preprocess.{n}.bias"].contiguous())
    for i in range(2):
        for n in lin_names:
            ws.append(sd[f"loftr_fine.layers.{i}.{n}.weight"].contiguous())
            bs.append(None)
    norms = []
    for pre, nl in (("loftr_coarse", 8), ("loftr_fine", 2)):
        for i in range(nl):
            for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias"):
                norms.append(sd[f"{pre}.layers.{i}.{n}"].contiguous())
    nl = lib.imcui_hip_loftr_num_layers()
    assert len(ws) == nl and len(norms) == lib.imcui_hip_loftr_num_norms(), (len(ws), nl)
    N, K = C.c_int(), C.c_int()
    w_np, b_np = [], []
    for i, (w, b) in enumerate(zip(ws, bs)):
        lib.imcui_hip_loftr_layer_shape(i, C.byref(N), C.byref(K))
        if tuple(w.shape) != (N.value, K.value):
            raise ImcuiHipError(f"LoFTR layer {i}: expected {(N.value, K.value)}, got {tuple(w.shape)}")
        w_np.append(_as_f32_host(w))
        b_np.append(None if b is None else _as_f32_host(b))
    n_np = [_as_f32_host(n) for n in norms]
    packed = np.zeros(lib.imcui_hip_loftr_packed_floats(), dtype=np.float32)
    wp = (C.c_void_p * nl)(*[a.ctypes.data for a in w_np])
    bp = (C.c_void_p * nl)(*[(0 if a is None else a.ctypes.data) for a in b_np])
    npp = (C.c_void_p * len(n_np))(*[a.ctypes.data for a in n_np])
    c1w, c1b = np.ascontiguousarray(conv1_w, dtype=np.float32), np.ascontiguousarray(conv1_b, dtype=np.float32)
    rc = lib.imcui_hip_loftr_pack_weights(c1w.ctypes.data, c1b.ctypes.data, wp, bp, npp, packed.ctypes.data)
    if rc != 0:
        raise ImcuiHipError(f"imcui_hip_loftr_pack_weights failed ({rc})")
    return torch.from_numpy(packed)


class LoFTRHIP:
    def __init__(self):
        self._ws = _Workspace()
        self._lock = threading.Lock()
        self.last_ws = None

    def forward(self, packed, image0, image1, match_threshold, temp_bug_fix=False):
        """kornia LoFTR.forward on image0 [B,1,H0,W0] / image1 [B,1,H1,W1] (sizes may differ between the two sides);
        fixed-capacity outputs + device match count."""
        dev = image0.device
        hd = get_handle(dev)
        lib = hd.lib
        image0, image1 = image0.contiguous().float(), image1.contiguous().float()
        B, Cc, H0, W0 = image0.shape
        B1, C1, H1, W1 = image1.shape
        if Cc != 1 or C1 != 1 or B1 != B:
            raise ImcuiHipError("LoFTR expects two batches of 1-channel images of equal batch size")
        cap = B * (H0 // 8) * (W0 // 8)
        kp0 = torch.empty((cap, 2), dtype=torch.float32, device=dev)
        kp1 = torch.empty((cap, 2), dtype=torch.float32, device=dev)
        conf = torch.empty((cap,), dtype=torch.float32, device=dev)
        bidx = torch.empty((cap,), dtype=torch.int32, device=dev)
        nm = torch.zeros((1,), dtype=torch.int32, device=dev)
        with self._lock:
            ws = self._ws.get(lib.imcui_hip_loftr_workspace_bytes(B, H0, W0, H1, W1), dev)
            self.last_ws = ws
            self.last_dims = (B, H0, W0, H1, W1)
            with torch.cuda.device(dev):
                rc = lib.imcui_hip_loftr_forward(
                    hd.h, _ptr(packed), _ptr(image0), _ptr(image1), B, H0, W0, H1, W1, float(match_threshold), int(bool(temp_bug_fix)),
                    _ptr(kp0), _ptr(kp1), _ptr(conf), _ptr(bidx), _ptr(nm), _ptr(ws), ws.numel(), _stream_ptr(),
                )  # fmt: skip
                hd.check(rc, "imcui_hip_loftr_forward")
        return {"keypoints0": kp0, "keypoints1": kp1, "confidence": conf, "batch_indexes": bidx, "num_matches": nm}

    @staticmethod
    def last_fine_mode(dev) -> tuple:
        """(mode, matches) of the last forward on this device's handle (imcui_hip_loftr_last_fine_mode): mode 0 = the last FPN stage as dense
        maps, 1 = on the 5x5 windows of the matches; matches = the count read back for that decision (-1: none)."""
        hd = get_handle(dev)
        m = C.c_int(-1)
        return int(hd.lib.imcui_hip_loftr_last_fine_mode(hd.h, C.byref(m))), m.value

    def debug_buffer(self, which: int, shape) -> torch.Tensor:
        """Workspace buffer `which` of the last forward (imcui_hip_loftr_debug_offset) viewed as float32 `shape`."""
        lib = load_library()
        off = lib.imcui_hip_loftr_debug_offset(which, *self.last_dims)
        n = int(np.prod(shape))
        return self.last_ws[off : off + 4 * n].view(torch.float32).view(*shape)


# ------------------------------------------------------------------ EfficientLoFTR
def _repvgg_reparam(sd: dict, p: str, eps: float = 1e-5):
    """One RepVGG block (3x3 + BN, 1x1 + BN, optional identity BN) -> a single 3x3 convolution (w, b): the
    `reparameter()` step of the reference wrapper (eloftr.py:61), done in float64."""
    def branch(wk, bn):
        w = sd[wk].double()
        g = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + eps)
        return w * g.view(-1, 1, 1, 1), sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * g

    w3, b3 = branch(p + ".conv1.conv.weight", p + ".conv1.norm")
    w1, b1 = branch(p + ".conv2.conv.weight", p + ".conv2.norm")
    w = w3.clone()
    w[:, :, 1:2, 1:2] += w1
    b = b3 + b1
    if p + ".identity.weight" in sd:
        c = w.shape[0]
        g = sd[p + ".identity.weight"].double() / torch.sqrt(sd[p + ".identity.running_var"].double() + eps)
        w[torch.arange(c), torch.arange(c), 1, 1] += g
        b = b + sd[p + ".identity.bias"].double() - sd[p + ".identity.running_mean"].double() * g
    return w.float(), b.float()


def pack_eloftr(state_dict: dict) -> torch.Tensor:
    """EfficientLoFTR state dict (names of `transformers.EfficientLoFTRForKeypointMatching`) -> packed float32 buffer
    (host): RepVGG re-parameterisation, BatchNorm folding, GEMM layouts."""
    lib = load_library()
    sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if isinstance(v, torch.Tensor)}
    bb = "efficientloftr.backbone.stages."
    w0, b0 = _repvgg_reparam(sd, bb + "0.blocks.0")
    conv0_w = np.ascontiguousarray(w0[:, 0].reshape(64, 9).t().numpy(), dtype=np.float32)  # [tap][cout]
    conv0_b = np.ascontiguousarray(b0.numpy(), dtype=np.float32)
    ws, bs = [], []
    for s, nb in ((1, 2), (2, 4), (3, 14)):
        for b in range(nb):
            w, bias = _repvgg_reparam(sd, f"{bb}{s}.blocks.{b}")
            wg, bg = _conv_gemm_layout(w, bias, w.shape[0], w.shape[1])
            ws.append(wg)
            bs.append(bg)
    tr = "efficientloftr.local_feature_transformer.layers."
    dws, norms = [], []
    for layer in range(4):
        for kind in ("self_attention", "cross_attention"):
            p = f"{tr}{layer}.{kind}."
            for n in ("attention.q_proj", "attention.k_proj", "attention.v_proj", "attention.o_proj", "mlp.fc1", "mlp.fc2"):
                ws.append(sd[p + n + ".weight"].contiguous())
                bs.append(None)
            dws.append(sd[p + "aggregation.q_aggregation.weight"].reshape(256, 16).contiguous())
            norms += [sd[p + "aggregation.norm.weight"], sd[p + "aggregation.norm.bias"], sd[p + "mlp.layer_norm.weight"], sd[p + "mlp.layer_norm.bias"]]
    rf = "refinement_layer."
    w = sd[rf + "out_conv.weight"] / 16.0  # coarse features enter the fusion divided by sqrt(256)
    wg, bg = _conv_gemm_layout(w, torch.zeros(w.shape[0]), w.shape[0], w.shape[1])
    ws.append(wg)
    bs.append(bg)
    for i in range(2):
        p = f"{rf}out_conv_layers.{i}."
        for wk, bn in ((p + "out_conv1", None), (p + "out_conv2", p + "batch_norm"), (p + "out_conv3", None)):
            w, bias = _fold_bn(sd[wk + ".weight"], sd, bn)
            wg, bg = _conv_gemm_layout(w, bias, w.shape[0], w.shape[1])
            ws.append(wg)
            bs.append(bg)
    nl = lib.imcui_hip_eloftr_num_layers()
    assert len(ws) == nl, (len(ws), nl)
    N, K = C.c_int(), C.c_int()
    w_np, b_np = [], []
    for i, (w, b) in enumerate(zip(ws, bs)):
        lib.imcui_hip_eloftr_layer_shape(i, C.byref(N), C.byref(K))
        if tuple(w.shape) != (N.value, K.value):
            raise ImcuiHipError(f"EfficientLoFTR layer {i}: expected {(N.value, K.value)}, got {tuple(w.shape)}")
        w_np.append(_as_f32_host(w))
        b_np.append(None if b is None else _as_f32_host(b))
    d_np = [_as_f32_host(d) for d in dws]
    n_np = [_as_f32_host(n) for n in norms]
    inv_freq = _as_f32_host(1.0 / (10000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128)))  # the port's rope init, float32
    packed = np.zeros(lib.imcui_hip_eloftr_packed_floats(), dtype=np.float32)
    wp = (C.c_void_p * nl)(*[a.ctypes.data for a in w_np])
    bp = (C.c_void_p * nl)(*[(0 if a is None else a.ctypes.data) for a in b_np])
    dp = (C.c_void_p * 8)(*[a.ctypes.data for a in d_np])
    npp = (C.c_void_p * len(n_np))(*[a.ctypes.data for a in n_np])
    rc = lib.imcui_hip_eloftr_pack_weights(conv0_w.ctypes.data, conv0_b.ctypes.data, wp, bp, dp, npp, inv_freq.ctypes.data, packed.ctypes.data)
    if rc != 0:
        raise ImcuiHipError(f"imcui_hip_eloftr_pack_weights failed ({rc})")
    return torch.from_numpy(packed)


class ELoFTRHIP:
    def __init__(self):
        self._ws = _Workspace()
        self._lock = threading.Lock()
        self.last_ws = None

    def forward(self, packed, image0, image1, match_threshold, debug_windows=False, arith=0):
        """Upstream EfficientLoFTR forward on image0 [B,1,H0,W0] / image1 [B,1,H1,W1] (multiples of 32; the two sizes may
        differ); fixed-capacity outputs + device match count.  arith 1 = the wrapper's precision "fp16" / "mp" (one f16
        product per element pair in the convolutions)."""
        dev = image0.device
        hd = get_handle(dev)
        lib = hd.lib
        image0, image1 = image0.contiguous().float(), image1.contiguous().float()
        B, Cc, H0, W0 = image0.shape
        B1, C1, H1, W1 = image1.shape
        if Cc != 1 or C1 != 1 or B1 != B:
            raise ImcuiHipError("EfficientLoFTR expects two batches of 1-channel images of equal batch size")
        cap = B * (H0 // 8) * (W0 // 8)
        kp0 = torch.empty((cap, 2), dtype=torch.float32, device=dev)
        kp1 = torch.empty((cap, 2), dtype=torch.float32, device=dev)
        conf = torch.empty((cap,), dtype=torch.float32, device=dev)
        bidx = torch.empty((cap,), dtype=torch.int32, device=dev)
        nm = torch.zeros((1,), dtype=torch.int32, device=dev)
        with self._lock:
            ws = self._ws.get(lib.imcui_hip_eloftr_workspace_bytes(B, H0, W0, H1, W1, int(bool(debug_windows))), dev)
            self.last_ws = ws
            self.last_dims = (B, H0, W0, H1, W1)
            with torch.cuda.device(dev):
                rc = lib.imcui_hip_eloftr_forward_ex(
                    hd.h, _ptr(packed), _ptr(image0), _ptr(image1), B, H0, W0, H1, W1, float(match_threshold), int(arith), _ptr(kp0), _ptr(kp1),
                    _ptr(conf), _ptr(bidx), _ptr(nm), int(bool(debug_windows)), _ptr(ws), ws.numel(), _stream_ptr(),
                )  # fmt: skip
                hd.check(rc, "imcui_hip_eloftr_forward")
        return {"keypoints0": kp0, "keypoints1": kp1, "confidence": conf, "batch_indexes": bidx, "num_matches": nm}

    def debug_buffer(self, which: int, shape) -> torch.Tensor:
        """Workspace buffer `which` of the last forward (imcui_hip_eloftr_debug_offset) viewed as float32 `shape`."""
        lib = load_library()
        off = lib.imcui_hip_eloftr_debug_offset(which, *self.last_dims)
        n = int(np.prod(shape))
        return self.last_ws[off : off + 4 * n].view(torch.float32).view(*shape)


DUST3R_CFG = {"enc_dim": 1024, "enc_depth": 24, "dec_dim": 768, "dec_depth": 12, "desc_dim": 0}  # DUSt3R_ViTLarge_BaseDecoder_512_dpt


def _dust3r_c5(cfg: dict) -> tuple:
    return (cfg["enc_dim"], cfg["enc_depth"], cfg["dec_dim"], cfg["dec_depth"], cfg.get("desc_dim", 0))


def dust3r_cfg_of(state_dict: dict) -> dict:
    """Architecture of an `AsymmetricCroCo3DStereo` state dict (dims from the tensors, depths from the block indices)."""
    def depth(prefix):
        return 1 + max(int(k[len(prefix) :].split(".")[0]) for k in state_dict if k.startswith(prefix))

    return {
        "enc_dim": state_dict["patch_embed.proj.weight"].shape[0],
        "enc_depth": depth("enc_blocks."),
        "dec_dim": state_dict["decoder_embed.weight"].shape[0],
        "dec_depth": depth("dec_blocks."),
        # MASt3R: head_local_features.fc2 emits (desc_dim + 1) x 16 x 16 values per token
        "desc_dim": (state_dict["downstream_head1.head_local_features.fc2.weight"].shape[0] // 256 - 1) if "downstream_head1.head_local_features.fc2.weight" in state_dict else 0,
    }


def _fold_layernorm(w: torch.Tensor, b, gamma: torch.Tensor, beta: torch.Tensor):
    """(W, b) of a linear layer that reads LayerNorm(x; gamma, beta) -> (W * gamma, b + W beta): the layer then reads the plain
    normalisation of x.  float64 for the bias sum, one f32 rounding per weight."""
    w64 = w.double()
    b64 = (b.double() if b is not None else torch.zeros(w.shape[0], dtype=torch.float64)) + w64 @ beta.double()
    return (w64 * gamma.double()[None, :]).float(), b64.float()


def dust3r_matrices(state_dict: dict):
    """The matrices [N, K], biases and f32 vectors of a DUSt3R / MASt3R state dict in the order of the C layer table (what
    `pack_dust3r` hands to imcui_hip_dust3r_pack_weights) -> (cfg, matrices, biases, vectors).

    The affine part of every LayerNorm whose ONLY consumer is a linear layer is folded into that layer (round 3):
    `LN(x) W^T + b = xhat (W * gamma)^T + (b + W beta)` with `xhat = (x - mean) / sqrt(var + eps)`, so the device normalises without
    gamma / beta -- one pass serves `norm1` and the other side's `norm_y` of a decoder block (same input, different affine parts) and
    both decoder sides at once.  Folded: encoder norm1 -> attn.qkv, norm2 -> mlp.fc1; decoder norm1 -> attn.qkv, norm2 ->
    cross_attn.projq, norm_y -> cross_attn.projk / projv, norm3 -> mlp.fc1.  enc_norm and dec_norm feed the DPT heads as they are and stay
    LayerNorms.  The folded vectors are still handed over (layout unchanged); the device ignores them."""
    sd = {k: v.detach().float().cpu() for k, v in state_dict.items() if isinstance(v, torch.Tensor)}
    cfg = dust3r_cfg_of(sd)
    ws, bs, vecs = [], [], []

    def lin(name, ln=None):
        w = sd[name + ".weight"].reshape(sd[name + ".weight"].shape[0], -1)
        b = sd.get(name + ".bias")
        if ln is not None:
            w, b = _fold_layernorm(w, b, sd[ln + ".weight"], sd[ln + ".bias"])
        ws.append(w.contiguous())
        bs.append(b)

    def conv(name):  # OIHW -> [Cout][tap][Cin]
        w = sd[name + ".weight"]
        ws.append(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous())
        bs.append(sd.get(name + ".bias"))

    def deconv(name):  # [Cin][Cout][s][s] -> [(dy, dx, cout)][cin], bias tiled over (dy, dx)
        w = sd[name + ".weight"]
        s = w.shape[2]
        ws.append(w.permute(2, 3, 1, 0).reshape(s * s * w.shape[1], w.shape[0]).contiguous())
        bs.append(sd[name + ".bias"].repeat(s * s).contiguous())

    def norm(name):
        vecs.extend([sd[name + ".weight"], sd[name + ".bias"]])

    lin("patch_embed.proj")
    for i in range(cfg["enc_depth"]):
        p = f"enc_blocks.{i}."
        lin(p + "attn.qkv", p + "norm1")
        lin(p + "attn.proj")
        lin(p + "mlp.fc1", p + "norm2")
        lin(p + "mlp.fc2")
        norm(p + "norm1")
        norm(p + "norm2")
    norm("enc_norm")
    lin("decoder_embed")
    for blocks in ("dec_blocks", "dec_blocks2"):
        for i in range(cfg["dec_depth"]):
            p = f"{blocks}.{i}."
            lin(p + "attn.qkv", p + "norm1")
            lin(p + "attn.proj")
            lin(p + "cross_attn.projq", p + "norm2")
            # keys / values are projected from the OTHER stream's tokens, normalised by THIS block's norm_y
            wkv, bkv = _fold_layernorm(torch.cat((sd[p + "cross_attn.projk.weight"], sd[p + "cross_attn.projv.weight"]), 0),
                                       torch.cat((sd[p + "cross_attn.projk.bias"], sd[p + "cross_attn.projv.bias"]), 0),
                                       sd[p + "norm_y.weight"], sd[p + "norm_y.bias"])  # fmt: skip
            ws.append(wkv.contiguous())
            bs.append(bkv.contiguous())
            lin(p + "cross_attn.proj")
            lin(p + "mlp.fc1", p + "norm3")
            lin(p + "mlp.fc2")
            for n in ("norm1", "norm2", "norm_y", "norm3"):
                norm(p + n)
    norm("dec_norm")
    for hd in (1, 2):
        p = f"downstream_head{hd}.dpt."
        lin(p + "act_postprocess.0.0")
        deconv(p + "act_postprocess.0.1")
        lin(p + "act_postprocess.1.0")
        deconv(p + "act_postprocess.1.1")
        lin(p + "act_postprocess.2.0")
        lin(p + "act_postprocess.3.0")
        conv(p + "act_postprocess.3.1")
        for k in range(4):
            conv(f"{p}scratch.layer_rn.{k}")
        for r in (4, 3, 2, 1):
            for u in (1, 2):
                conv(f"{p}scratch.refinenet{r}.resConfUnit{u}.conv1")
                conv(f"{p}scratch.refinenet{r}.resConfUnit{u}.conv2")
            lin(f"{p}scratch.refinenet{r}.out_conv")
        conv(p + "head.0")
        conv(p + "head.2")
        if cfg["desc_dim"] > 0:
            lin(f"downstream_head{hd}.head_local_features.fc1")
            lin(f"downstream_head{hd}.head_local_features.fc2")
    for hd in (1, 2):
        p = f"downstream_head{hd}.dpt.head.4"
        vecs.extend([sd[p + ".weight"].reshape(4, 128).contiguous(), sd[p + ".bias"]])
    vecs.append(1.0 / (100.0 ** (torch.arange(0, 32, 2, dtype=torch.float32) / 32)))  # RoPE2D(freq=100), D = 32 per axis
    return cfg, ws, bs, vecs


def pack_dust3r(state_dict: dict) -> tuple[torch.Tensor, dict]:
    """`AsymmetricCroCo3DStereo` state dict (head_type 'dpt', upstream names; imcui/hloc/matchers/duster.py:37) -> (packed
    float32 host buffer, cfg).  Layer order and layouts: include/imcui_hip.h, the DUSt3R section."""
    lib = load_library()
    cfg, ws, bs, vecs = dust3r_matrices(state_dict)
    c4 = _dust3r_c5(cfg)
    nl = lib.imcui_hip_dust3r_num_layers(*c4)
    if nl == 0:
        raise ImcuiHipError(f"DUSt3R configuration {cfg} is not supported (dims multiples of 64 up to 1024, dec_depth a multiple of 4)")
    nv = lib.imcui_hip_dust3r_num_vectors(*c4)
    assert len(ws) == nl and len(vecs) == nv, (len(ws), nl, len(vecs), nv)
    N, K = C.c_int(), C.c_int()
    w_np, b_np, v_np = [], [], []
    for i, (w, b) in enumerate(zip(ws, bs)):
        lib.imcui_hip_dust3r_layer_shape(*c4, i, C.byref(N), C.byref(K))
        if tuple(w.shape) != (N.value, K.value):
            raise ImcuiHipError(f"DUSt3R layer {i}: expected {(N.value, K.value)}, got {tuple(w.shape)}")
        w_np.append(_as_f32_host(w))
        b_np.append(None if b is None else _as_f32_host(b))
    for i, v in enumerate(vecs):
        if v.numel() != lib.imcui_hip_dust3r_vector_len(*c4, i):
            raise ImcuiHipError(f"DUSt3R vector {i}: expected {lib.imcui_hip_dust3r_vector_len(*c4, i)} values, got {v.numel()}")
        v_np.append(_as_f32_host(v))
    packed = np.zeros(lib.imcui_hip_dust3r_packed_floats(*c4), dtype=np.float32)
    wp = (C.c_void_p * nl)(*[a.ctypes.data for a in w_np])
    bp = (C.c_void_p * nl)(*[(0 if a is None else a.ctypes.data) for a in b_np])
    vp = (C.c_void_p * nv)(*[a.ctypes.data for a in v_np])
    rc = lib.imcui_hip_dust3r_pack_weights(*c4, wp, bp, vp, packed.ctypes.data)
    if rc != 0:
        raise ImcuiHipError(f"imcui_hip_dust3r_pack_weights failed ({rc})")
    return torch.from_numpy(packed), cfg


def check_dust3r_packed(packed: torch.Tensor, cfg: dict) -> None:
    """Raise unless `packed` is a buffer THIS library's packer wrote for `cfg` (size and format trailer; include/imcui_hip.h, the DUSt3R
    section).  Called once when a module is built from conf["packed"] -- the documented way to skip the ~20 s of packing -- so that a blob
    cached from an older layout is refused instead of running with its LayerNorm affine parts dropped."""
    lib = load_library()
    c4 = _dust3r_c5(cfg)
    want = lib.imcui_hip_dust3r_packed_floats(*c4)
    if packed.dtype != torch.float32 or packed.numel() != want:
        raise ImcuiHipError(f"DUSt3R packed buffer: {packed.numel()} {packed.dtype} values, format {lib.imcui_hip_dust3r_format_version()} of this library needs "
                            f"{want} float32 -- a blob of another layout / configuration; pack the state dict again (backend.pack_dust3r)")
    if packed.device.type == "cpu":
        rc = lib.imcui_hip_dust3r_check_packed(*c4, packed.contiguous().data_ptr(), want)
    else:  # device-resident: bring the 64-word trailer (the last words of the buffer) to the host and compare it here
        tail = packed.reshape(-1)[-64:].detach().cpu().contiguous().view(torch.int32)[:9].tolist()
        fmt = lib.imcui_hip_dust3r_format_version()
        expect = [0x494D4455, fmt, want & 0xFFFFFFFF, want >> 32, *c4]
        rc = 0 if [v & 0xFFFFFFFF for v in tail] == [v & 0xFFFFFFFF for v in expect] else -1
    if rc != 0:
        raise ImcuiHipError(f"DUSt3R packed buffer: no format-{lib.imcui_hip_dust3r_format_version()} trailer for configuration {cfg} -- not written by this library's "
                            "imcui_hip_dust3r_pack_weights; pack the state dict again (backend.pack_dust3r)")


class DUSt3RHIP:
    def __init__(self):
        self._ws = _Workspace()
        self._lock = threading.Lock()
        self.last_dump = None

    def forward(self, packed, cfg, images, pairs, dump=False, arith=0):
        """images [NI,3,H,W] in [0,1] (H, W multiples of 16), pairs [P,2] int (view-1 image, view-2 image) ->
        {"pts3d": [2,P,H,W,3], "conf": [2,P,H,W]} (view 1 in its own frame, view 2 in view 1's frame); a MASt3R network
        (cfg["desc_dim"] > 0) also returns "desc" [2,P,H,W,desc_dim] and "desc_conf" [2,P,H,W].
        arith: 0 = 3 x f16 split products (fp32-grade), 1 = one f16 product per element pair (bf16-class)."""
        dev = images.device
        hd = get_handle(dev)
        lib = hd.lib
        if lib.imcui_hip_get_precision(hd.h) != 1:  # imcui_hip_dust3r_forward exists in the 3 x f16 split arithmetic only
            raise ImcuiHipError("DUSt3R / MASt3R run in the library's default arithmetic (imcui_hip_set_precision(h, 1)): the exact-f32 matrix "
                                "mode has no ViT kernels (for an exact-f32 nearest-neighbour search set conf['matcher_arithmetic'] = 'fp32')")
        images = images.contiguous().float()
        NI, Cc, H, W = images.shape
        if Cc != 3:
            raise ImcuiHipError("DUSt3R expects 3-channel images")
        pairs = torch.as_tensor(pairs, dtype=torch.int32).reshape(-1, 2)
        if int(pairs.min()) < 0 or int(pairs.max()) >= NI:
            raise ImcuiHipError(f"DUSt3R pair table refers to images outside [0, {NI})")
        pairs = pairs.to(dev).contiguous()
        P = pairs.shape[0]
        c4 = _dust3r_c5(cfg)
        dd = c4[4]
        pts = torch.empty((2, P, H, W, 3), dtype=torch.float32, device=dev)
        conf = torch.empty((2, P, H, W), dtype=torch.float32, device=dev)
        desc = torch.empty((2, P, H, W, dd), dtype=torch.float32, device=dev) if dd else None
        dconf = torch.empty((2, P, H, W), dtype=torch.float32, device=dev) if dd else None
        nd = lib.imcui_hip_dust3r_dump_floats(*c4, NI, P, H, W) if dump else 0
        dbuf = torch.zeros((nd,), dtype=torch.float32, device=dev) if dump else None
        with self._lock:
            nbytes = lib.imcui_hip_dust3r_workspace_bytes(*c4, NI, P, H, W)
            if nbytes == 0:
                raise ImcuiHipError(f"DUSt3R: unsupported sizes ({NI} images of {W}x{H}, {P} pairs; multiples of 16)")
            ws = self._ws.get(nbytes, dev)
            with torch.cuda.device(dev):
                rc = lib.imcui_hip_dust3r_forward(hd.h, *c4, _ptr(packed), packed.numel(), _ptr(images), NI, H, W, _ptr(pairs), P, int(arith), _ptr(pts), _ptr(conf),
                                                  _ptr(desc), _ptr(dconf), _ptr(dbuf), nd, _ptr(ws), ws.numel(), _stream_ptr())  # fmt: skip
                hd.check(rc, "imcui_hip_dust3r_forward")
        self.last_dump = dbuf
        out = {"pts3d": pts, "conf": conf}
        if dd:
            out.update(desc=desc, desc_conf=dconf)
        return out

    def forward_sizes(self, packed, cfg, images, pairs, dump=False, arith=0):
        """The same network on images of several sizes (imcui_hip_dust3r_forward_sizes): `images` = list of [3,H_i,W_i] (or
        [1,3,H_i,W_i]) tensors in [0,1], every size a multiple of 16, at most 4 distinct sizes; pairs [P,2] ->
        {"pts3d": [view][pair] -> [H,W,3], "conf": [view][pair] -> [H,W]} (+ "desc" / "desc_conf" for a MASt3R network) as nested
        lists of views into ONE ragged device buffer per output: the map of (view v, pair p) has the size of image pairs[p][v]."""
        import ctypes as C

        imgs = [im.reshape(im.shape[-3:]).contiguous().float() for im in images]
        dev = imgs[0].device
        hd = get_handle(dev)
        lib = hd.lib
        if lib.imcui_hip_get_precision(hd.h) != 1:
            raise ImcuiHipError("DUSt3R / MASt3R run in the library's default arithmetic (imcui_hip_set_precision(h, 1))")
        NI = len(imgs)
        if any(im.shape[0] != 3 for im in imgs):
            raise ImcuiHipError("DUSt3R expects 3-channel images")
        sizes = [(int(im.shape[1]), int(im.shape[2])) for im in imgs]
        ptab = torch.as_tensor(pairs, dtype=torch.int32).reshape(-1, 2).cpu().contiguous()
        if int(ptab.min()) < 0 or int(ptab.max()) >= NI:
            raise ImcuiHipError(f"DUSt3R pair table refers to images outside [0, {NI})")
        P = ptab.shape[0]
        flat = torch.cat([im.reshape(-1) for im in imgs])
        pairs_dev = ptab.to(dev)
        c4 = _dust3r_c5(cfg)
        dd = c4[4]
        sz = (C.c_int * (2 * NI))(*[v for hw in sizes for v in hw])
        ph = (C.c_int * (2 * P))(*[int(v) for v in ptab.reshape(-1)])
        offs = (C.c_size_t * (2 * P + 1))()
        total = sum(sizes[int(ptab[p, v])][0] * sizes[int(ptab[p, v])][1] for v in range(2) for p in range(P))
        pts = torch.empty((total, 3), dtype=torch.float32, device=dev)
        conf = torch.empty((total,), dtype=torch.float32, device=dev)
        desc = torch.empty((total, dd), dtype=torch.float32, device=dev) if dd else None
        dconf = torch.empty((total,), dtype=torch.float32, device=dev) if dd else None
        # (images that all share one size take the one-size path inside the library, which also dumps the DPT head tensors)
        one_size = len(set(sizes)) == 1
        nd = (lib.imcui_hip_dust3r_dump_floats(*c4, NI, P, *sizes[0]) if one_size else lib.imcui_hip_dust3r_token_dump_floats(*c4, NI, sz, P)) if dump else 0
        dbuf = torch.zeros((nd,), dtype=torch.float32, device=dev) if dump else None
        with self._lock:
            nbytes = lib.imcui_hip_dust3r_workspace_bytes_sizes(*c4, NI, sz, P)
            if nbytes == 0:
                raise ImcuiHipError(f"DUSt3R: unsupported sizes {sizes} (multiples of 16, 32 .. 4096)")
            ws = self._ws.get(nbytes, dev)
            with torch.cuda.device(dev):
                rc = lib.imcui_hip_dust3r_forward_sizes(hd.h, *c4, _ptr(packed), packed.numel(), _ptr(flat), NI, sz, ph, _ptr(pairs_dev), P, int(arith), _ptr(pts), _ptr(conf),
                                                        _ptr(desc), _ptr(dconf), offs, _ptr(dbuf), nd, _ptr(ws), ws.numel(), _stream_ptr())  # fmt: skip
                hd.check(rc, "imcui_hip_dust3r_forward_sizes")
        assert offs[2 * P] == total
        self.last_dump = dbuf

        def maps(buf, tail):
            out = []
            for v in range(2):
                row = []
                for p in range(P):
                    Hh, Ww = sizes[int(ptab[p, v])]
                    o = offs[v * P + p]
                    row.append(buf[o : o + Hh * Ww].reshape((Hh, Ww) + tail))
                out.append(row)
            return out

        out = {"pts3d": maps(pts, (3,)), "conf": maps(conf, ())}
        if dd:
            out.update(desc=maps(desc, (dd,)), desc_conf=maps(dconf, ()))
        return out


def conv_gemm_f32(x_nhwc, w_oihw, bias, resid=None, stride=1, act=0):
    """Building block: NHWC conv through the implicit-im2col GEMM (k in {1,3}, Cin % 32 == 0)."""
    hd = get_handle(x_nhwc.device)
    B, H, W, Cin = x_nhwc.shape
    Cout, _, ks, _ = w_oihw.shape
    wg, bg = _conv_gemm_layout(w_oihw.float().cpu(), bias.float().cpu(), Cout, Cin)
    pad = ks // 2
    ho, wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    out = torch.empty((B, ho, wo, Cout), dtype=torch.float32, device=x_nhwc.device)
    wd, bd = wg.to(x_nhwc.device), bg.to(x_nhwc.device)
    x_nhwc = x_nhwc.contiguous().float()
    if resid is not None:
        resid = resid.contiguous().float()
    with torch.cuda.device(x_nhwc.device):
        hd.check(
            hd.lib.imcui_hip_conv_gemm_f32(hd.h, _ptr(x_nhwc), _ptr(wd), _ptr(bd), _ptr(resid), _ptr(out), B, H, W, Cin, Cout, ks, stride, act, _stream_ptr()),
            "conv_gemm",
        )
    return out


# ------------------------------------------------------------------ mutual NN
_nn_ws = _Workspace()
_nn_lock = threading.Lock()


def mutual_nn(desc0_nd: torch.Tensor, desc1_md: torch.Tensor, ratio_threshold=None, distance_threshold=None,
              do_mutual_check=True):  # fmt: skip
    """desc0 [B,N,D], desc1 [B,M,D] (row per descriptor) -> matches0 [B,N] int32, scores0 [B,N]."""
    dev = desc0_nd.device
    hd = get_handle(dev)
    lib = hd.lib
    desc0_nd, desc1_md = desc0_nd.contiguous().float(), desc1_md.contiguous().float()
    B, N, D = desc0_nd.shape
    M = desc1_md.shape[1]
    m0 = torch.empty((B, N), dtype=torch.int32, device=dev)
    s0 = torch.empty((B, N), dtype=torch.float32, device=dev)
    if D % 32:
        raise ImcuiHipError(f"descriptor dim {D} must be a multiple of 32")
    with _nn_lock:
        ws = _nn_ws.get(lib.imcui_hip_mutual_nn_workspace_bytes_d(hd.h, B, N, M, D), dev)  # (no similarity matrix for D = 64 / 128 / 256 or the split arithmetic)
        with torch.cuda.device(dev):
            rc = lib.imcui_hip_mutual_nn(
                hd.h, _ptr(desc0_nd), _ptr(desc1_md), B, N, M, D, float(ratio_threshold or 0.0),
                float(distance_threshold or 0.0), int(bool(do_mutual_check)), _ptr(m0), _ptr(s0), _ptr(ws), ws.numel(),
                _stream_ptr(),
            )  # fmt: skip
            hd.check(rc, "imcui_hip_mutual_nn")
    return m0, s0


def mutual_nn_dn(desc0_dn: torch.Tensor, desc1_dm: torch.Tensor, ratio_threshold=None, distance_threshold=None, do_mutual_check=True):
    """`mutual_nn` on the layout `NearestNeighbor._forward` receives: desc0 [B,D,N], desc1 [B,D,M] (column per descriptor); the transpose
    runs on the device inside the call (imcui_hip_mutual_nn_dn) -> matches0 [B,N] int32, scores0 [B,N]."""
    dev = desc0_dn.device
    hd = get_handle(dev)
    lib = hd.lib
    desc0_dn, desc1_dm = desc0_dn.contiguous().float(), desc1_dm.contiguous().float()
    B, D, N = desc0_dn.shape
    M = desc1_dm.shape[2]
    m0 = torch.empty((B, N), dtype=torch.int32, device=dev)
    s0 = torch.empty((B, N), dtype=torch.float32, device=dev)
    if D % 32:
        raise ImcuiHipError(f"descriptor dim {D} must be a multiple of 32")
    with _nn_lock:
        ws = _nn_ws.get(lib.imcui_hip_mutual_nn_dn_workspace_bytes_for(hd.h, B, N, M, D), dev)
        with torch.cuda.device(dev):
            rc = lib.imcui_hip_mutual_nn_dn(
                hd.h, _ptr(desc0_dn), _ptr(desc1_dm), B, N, M, D, float(ratio_threshold or 0.0), float(distance_threshold or 0.0),
                int(bool(do_mutual_check)), _ptr(m0), _ptr(s0), _ptr(ws), ws.numel(), _stream_ptr(),
            )  # fmt: skip
            hd.check(rc, "imcui_hip_mutual_nn_dn")
    return m0, s0


def nn_argmax(queries: torch.Tensor, db: torch.Tensor, return_best: bool = False, split: bool = False):
    """queries [Q,D], db [N,D] (D in 16 / 24 / 32) -> int64 [Q]: the FIRST arg-max over n of <queries[q], db[n]> (what
    `cdistMatcher(dist="dot").query` of upstream's mast3r/fast_nn.py returns; imcui/hloc/matchers/mast3r.py:68-75).
    split=True: the 3 x f16 split arithmetic (4 x fewer matrix cycles, fp32-grade values; near-ties may resolve differently)."""
    dev = queries.device
    hd = get_handle(dev)
    lib = hd.lib
    queries, db = queries.contiguous().float(), db.contiguous().float()
    Q, D = queries.shape
    N = db.shape[0]
    idx = torch.empty((Q,), dtype=torch.int32, device=dev)
    best = torch.empty((Q,), dtype=torch.float32, device=dev) if return_best else None
    if Q == 0:
        return (idx.long(), best) if return_best else idx.long()
    with _nn_lock:
        fn_ws, fn = ((lib.imcui_hip_nn_argmax_split_workspace_bytes, lib.imcui_hip_nn_argmax_split_f32) if split else
                     (lib.imcui_hip_nn_argmax_workspace_bytes, lib.imcui_hip_nn_argmax_f32))
        ws = _nn_ws.get(fn_ws(Q, N), dev)
        with torch.cuda.device(dev):
            rc = fn(hd.h, _ptr(queries), _ptr(db), Q, N, D, _ptr(idx), _ptr(best), _ptr(ws), ws.numel(), _stream_ptr())
            hd.check(rc, "imcui_hip_nn_argmax_f32")
    return (idx.long(), best) if return_best else idx.long()


def dual_softmax(desc0: torch.Tensor, desc1: torch.Tensor, threshold: float = 0.2, inv_temperature: float = 20.0, normalize: bool = True):
    """desc0 [B,C,N], desc1 [B,C,M] (channels-first, as the plugin receives them) -> matches0 [B,N] int32, scores0 [B,N]."""
    dev = desc0.device
    hd = get_handle(dev)
    lib = hd.lib
    desc0, desc1 = desc0.contiguous().float(), desc1.contiguous().float()
    B, C, N = desc0.shape
    M = desc1.shape[2]
    m0 = torch.empty((B, N), dtype=torch.int32, device=dev)
    s0 = torch.empty((B, N), dtype=torch.float32, device=dev)
    with _nn_lock:
        ws = _nn_ws.get(lib.imcui_hip_dual_softmax_workspace_bytes(B, C, N, M), dev)
        with torch.cuda.device(dev):
            rc = lib.imcui_hip_dual_softmax(hd.h, _ptr(desc0), _ptr(desc1), B, C, N, M, float(threshold), float(inv_temperature),
                                            int(bool(normalize)), _ptr(m0), _ptr(s0), _ptr(ws), ws.numel(), _stream_ptr())  # fmt: skip
            hd.check(rc, "imcui_hip_dual_softmax")
    return m0, s0


# ------------------------------------------------------------------ arithmetic mode
def set_precision(device: torch.device, mode: int):
    """0 = exact f32 MFMA, 1 = 3 x f16 split MFMA (default; ~fp32 accuracy at ~5x the matrix rate)."""
    hd = get_handle(device)
    hd.check(hd.lib.imcui_hip_set_precision(hd.h, int(mode)), "set_precision")


def get_precision(device: torch.device) -> int:
    hd = get_handle(device)
    return hd.lib.imcui_hip_get_precision(hd.h)


def lib_version() -> int:
    """imcui_hip_version() of the loaded library (400 since round 4: packed-buffer formats are tied to it)."""
    return int(load_library().imcui_hip_version())


def set_option(device: torch.device, name: str, value: int) -> int:
    """A/B switch of the kernel routing (imcui_hip_set_option: "gemm_wreg", "wreg_pipe", "attn_variant", "attn_variant_self", "attn_variant_cross", "attn_mix_layers", "simred", "ffn_tile", "wreg_tile", "conv_tall", "conv_narrow", "attn_split", "loftr_fine_sparse"); returns
    the previous value.  The IMCUI_* environment variables of the same names are only read when the handle is created."""
    hd = get_handle(device)
    old = C.c_int(0)
    hd.check(hd.lib.imcui_hip_get_option(hd.h, name.encode(), C.byref(old)), "get_option")
    hd.check(hd.lib.imcui_hip_set_option(hd.h, name.encode(), int(value)), "set_option")
    return old.value


def get_option(device: torch.device, name: str) -> int:
    """Current value of a routing switch (imcui_hip_get_option)."""
    hd = get_handle(device)
    val = C.c_int(0)
    hd.check(hd.lib.imcui_hip_get_option(hd.h, name.encode(), C.byref(val)), "get_option")
    return val.value


class option:
    """`with backend.option(dev, attn_variant=6): ...` -- set switches for a block of calls and restore them afterwards."""

    def __init__(self, device: torch.device, **kw):
        self.device, self.kw, self.old = device, kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = set_option(self.device, k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.old.items():
            set_option(self.device, k, v)


# ------------------------------------------------------------------ live kernel timing
KERNEL_CLASSES = {"attention": 0, "conv3x3": 1, "gemm": 2}


def profile_enable(device: torch.device, on: bool = True):
    hd = get_handle(device)
    hd.check(hd.lib.imcui_hip_profile_enable(hd.h, int(on)), "profile_enable")


def profile_read(device: torch.device, kernel_class: str):
    """(total kernel ms, launches) of one class since the last read; HIP events on the launch stream."""
    hd = get_handle(device)
    tot, cnt = C.c_double(0.0), C.c_int(0)
    hd.check(hd.lib.imcui_hip_profile_read(hd.h, KERNEL_CLASSES[kernel_class], C.byref(tot), C.byref(cnt)), "profile_read")
    return tot.value, cnt.value


# ------------------------------------------------------------------ building blocks (tests)
def linear_f32(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, relu: bool = False) -> torch.Tensor:
    hd = get_handle(a.device)
    a, w = a.contiguous().float(), w.contiguous().float()
    M, K = a.shape
    N = w.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        hd.check(hd.lib.imcui_hip_linear_f32(hd.h, _ptr(a), _ptr(w), _ptr(bias), _ptr(c), M, N, K, int(relu), _stream_ptr()), "linear")
    return c


def rgb_to_gray(rgb_u8: torch.Tensor) -> torch.Tensor:
    """Device-side `cv2.cvtColor(RGB2GRAY)` + `astype(float32) / 255` (extract_features.py:120-160) for decoded
    uint8 images [B,H,W,3] already on the device -> [B,1,H,W] float32 in [0,1]."""
    if rgb_u8.dtype != torch.uint8 or rgb_u8.dim() != 4 or rgb_u8.shape[-1] != 3:
        raise ImcuiHipError("rgb_to_gray expects uint8 [B,H,W,3]")
    hd = get_handle(rgb_u8.device)
    rgb_u8 = rgb_u8.contiguous()
    B, H, W, _ = rgb_u8.shape
    out = torch.empty((B, 1, H, W), dtype=torch.float32, device=rgb_u8.device)
    with torch.cuda.device(rgb_u8.device):
        hd.check(hd.lib.imcui_hip_rgb_to_gray_f32(hd.h, _ptr(rgb_u8), _ptr(out), B, H, W, _stream_ptr()), "rgb_to_gray")
    return out


_area_tables: dict = {}


def _area_table(ssize: int, dsize: int, device: torch.device):
    """OpenCV INTER_AREA decimation table ssize -> dsize as device tensors (start [dsize+1], index, weight); cached."""
    key = (ssize, dsize, str(device))
    if key not in _area_tables:
        lib = load_library()
        start = np.zeros(dsize + 1, dtype=np.int32)
        n = lib.imcui_hip_area_table(ssize, dsize, start.ctypes.data, None, None)
        if n <= 0:
            raise ImcuiHipError(f"area table {ssize} -> {dsize}: only shrinking resizes are supported")
        idx, wgt = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.float32)
        lib.imcui_hip_area_table(ssize, dsize, start.ctypes.data, idx.ctypes.data, wgt.ctypes.data)
        _area_tables[key] = tuple(torch.from_numpy(a).to(device) for a in (start, idx, wgt))
    return _area_tables[key]


def preprocess_area(img_u8: torch.Tensor, size) -> torch.Tensor:
    """Device-side `extract.preprocess` with a resize (extract_features.py:120-148): uint8 [B,H,W] / [B,H,W,1] gray or
    [B,H,W,3] RGB on the device -> gray (cv2 fixed point) -> float32 -> cv2.INTER_AREA to `size` = (w, h) -> / 255 ->
    float32 [B,1,h,w].  Shrinking resizes only."""
    if img_u8.dtype != torch.uint8 or img_u8.dim() not in (3, 4):
        raise ImcuiHipError("preprocess_area expects uint8 [B,H,W] or [B,H,W,C]")
    if img_u8.dim() == 3:
        img_u8 = img_u8[..., None]
    hd = get_handle(img_u8.device)
    img_u8 = img_u8.contiguous()
    B, H, W, Cc = img_u8.shape
    ow, oh = int(size[0]), int(size[1])
    out = torch.empty((B, 1, oh, ow), dtype=torch.float32, device=img_u8.device)
    tabs = [None] * 6
    if not (W % ow == 0 and H % oh == 0) and ow <= W and oh <= H:
        tabs = [*_area_table(W, ow, img_u8.device), *_area_table(H, oh, img_u8.device)]
    with torch.cuda.device(img_u8.device):
        hd.check(hd.lib.imcui_hip_preprocess_area_f32(hd.h, _ptr(img_u8), B, H, W, Cc, *[_ptr(t) for t in tabs], _ptr(out), oh, ow, _stream_ptr()),
                 "preprocess_area")
    return out


_linear_tables: dict = {}
_aa_tables: dict = {}


def linear_table_host(ssize: int, dsize: int, horizontal: bool):
    """cv2.INTER_LINEAR tap table (host numpy): (i0, i1, w1)."""
    lib = load_library()
    i0, i1, w1 = np.zeros(dsize, dtype=np.int32), np.zeros(dsize, dtype=np.int32), np.zeros(dsize, dtype=np.float32)
    if lib.imcui_hip_linear_table(ssize, dsize, int(horizontal), i0.ctypes.data, i1.ctypes.data, w1.ctypes.data) != 0:
        raise ImcuiHipError(f"linear table {ssize} -> {dsize}")
    return i0, i1, w1


def aa_table_host(in_size: int, out_size: int):
    """ATen anti-aliased bilinear tap table (host numpy): (first, count, weights [out_size, kmax])."""
    lib = load_library()
    kmax = lib.imcui_hip_aa_table(in_size, out_size, None, None, None, 0)
    if kmax <= 0:
        raise ImcuiHipError(f"aa table {in_size} -> {out_size}")
    first, count = np.zeros(out_size, dtype=np.int32), np.zeros(out_size, dtype=np.int32)
    w = np.zeros((out_size, kmax), dtype=np.float32)
    if lib.imcui_hip_aa_table(in_size, out_size, first.ctypes.data, count.ctypes.data, w.ctypes.data, kmax) != kmax:
        raise ImcuiHipError(f"aa table {in_size} -> {out_size}")
    return first, count, w


def preprocess_linear(img_u8: torch.Tensor, size) -> torch.Tensor:
    """Device-side `resize_image(..., "cv2_area")` when a side GROWS (extract_features.py:29-31: cv2.INTER_LINEAR): uint8
    [B,H,W] / [B,H,W,C] on the device -> gray (cv2 fixed point) -> float32 -> INTER_LINEAR to `size` = (w, h) -> / 255 ->
    float32 [B,1,h,w]."""
    if img_u8.dtype != torch.uint8 or img_u8.dim() not in (3, 4):
        raise ImcuiHipError("preprocess_linear expects uint8 [B,H,W] or [B,H,W,C]")
    if img_u8.dim() == 3:
        img_u8 = img_u8[..., None]
    hd = get_handle(img_u8.device)
    img_u8 = img_u8.contiguous()
    B, H, W, Cc = img_u8.shape
    ow, oh = int(size[0]), int(size[1])
    key = (W, ow, H, oh, str(img_u8.device))
    if key not in _linear_tables:
        _linear_tables[key] = tuple(torch.from_numpy(a).to(img_u8.device) for a in (*linear_table_host(W, ow, True), *linear_table_host(H, oh, False)))
    tabs = _linear_tables[key]
    out = torch.empty((B, 1, oh, ow), dtype=torch.float32, device=img_u8.device)
    with torch.cuda.device(img_u8.device):
        hd.check(hd.lib.imcui_hip_preprocess_linear_f32(hd.h, _ptr(img_u8), B, H, W, Cc, *[_ptr(t) for t in tabs], _ptr(out), oh, ow, _stream_ptr()),
                 "preprocess_linear")
    return out


def resize_aa(image: torch.Tensor, size_hw) -> torch.Tensor:
    """Device-side `torchvision.transforms.functional.resize(image, size, antialias=True)` of a float image [..., H, W]
    (extract_features.py:142-148, match_dense.py:182): ATen's anti-aliased bilinear arithmetic, bit for bit.  Like torchvision,
    returns the image unchanged when the size already matches."""
    oh, ow = int(size_hw[0]), int(size_hw[1])
    H, W = image.shape[-2:]
    if (oh, ow) == (H, W):
        return image
    hd = get_handle(image.device)
    src = image.contiguous().float()
    planes = src.numel() // (H * W)
    key = (W, ow, H, oh, str(image.device))
    if key not in _aa_tables:
        xt, yt = aa_table_host(W, ow), aa_table_host(H, oh)
        _aa_tables[key] = (tuple(torch.from_numpy(a).to(image.device) for a in xt), xt[2].shape[1], tuple(torch.from_numpy(a).to(image.device) for a in yt), yt[2].shape[1])
    xt, kx, yt, ky = _aa_tables[key]
    out = torch.empty((*src.shape[:-2], oh, ow), dtype=torch.float32, device=image.device)
    with torch.cuda.device(image.device):
        hd.check(hd.lib.imcui_hip_resize_aa_f32(hd.h, _ptr(src), planes, H, W, _ptr(xt[0]), _ptr(xt[1]), _ptr(xt[2]), kx, _ptr(yt[0]), _ptr(yt[1]),
                                                _ptr(yt[2]), ky, _ptr(out), oh, ow, _stream_ptr()), "resize_aa")
    return out


def pack_linear_split(w: torch.Tensor):
    """Host: nn.Linear weight [N, K] -> (hi, lo) uint16 fragment-major planes and the inverse scale 2^-e."""
    from .lib_loader import load_library

    lib = load_library()
    wh = _as_f32_host(w)
    N, K = wh.shape
    npad = (N + 31) // 32 * 32
    hi = np.zeros(npad * K, dtype=np.uint16)
    lo = np.zeros(npad * K, dtype=np.uint16)
    sc = lib.imcui_hip_linear_pack_split(wh.ctypes.data, N, K, hi.ctypes.data, lo.ctypes.data)
    if sc <= 0:
        raise ImcuiHipError("linear_pack_split failed (K must be a multiple of 16)")
    return hi, lo, float(sc)


def linear_split_f32(a: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, relu: bool = False) -> torch.Tensor:
    """Building block: a projection with PRE-SPLIT weights, i.e. the GEMM path every network layer takes in the split
    mode (fragment-major planes, 128- or 256-column tiles by N)."""
    hd = get_handle(a.device)
    a = a.contiguous().float()
    M, K = a.shape
    N = w.shape[0]
    hi, lo, sc = pack_linear_split(w)
    dh = torch.from_numpy(hi.view(np.int16)).to(a.device)
    dl = torch.from_numpy(lo.view(np.int16)).to(a.device)
    ds = torch.tensor([sc], dtype=torch.float32, device=a.device)
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        hd.check(hd.lib.imcui_hip_linear_split_f32(hd.h, _ptr(a), _ptr(dh), _ptr(dl), _ptr(ds), _ptr(bias), _ptr(c), M, N, K, int(relu), _stream_ptr()),
                 "linear_split")
    return c


def set_range_check(device, enable: bool = True) -> None:
    """Debugging aid: scan the f32 activation operand of every split-mode GEMM / convolution / FFN launch for values the f16
    split cannot carry (imcui_hip_set_range_check)."""
    hd = get_handle(torch.device(device))
    hd.check(hd.lib.imcui_hip_set_range_check(hd.h, int(bool(enable))), "set_range_check")


def range_status(device) -> int:
    """Accumulated range-check word since the last read (synchronises): bit 0 = |x| > 65504 seen, bit 1 = NaN / Inf seen."""
    hd = get_handle(torch.device(device))
    st = C.c_int(0)
    hd.check(hd.lib.imcui_hip_get_range_status(hd.h, C.byref(st)), "get_range_status")
    return int(st.value)


def qkv_split_f32(x: torch.Tensor, w: torch.Tensor, bias, cos, sin, cnt: torch.Tensor, rows_per_seq: int, alpha: float, cross: bool = False):
    """Building block: the attention-layout projection (imcui_hip_qkv_split_f32).  x [nseq * R, 256] (device), w [768 | 512, 256]
    (host, packed row order) -> int16 views of the planes: q, k [2, nseq, 4, R, 64], vt [2, nseq, 4, 64, R] (plane 0 = hi)."""
    hd = get_handle(x.device)
    x = x.contiguous().float()
    R = int(rows_per_seq)
    nseq = x.shape[0] // R
    hi, lo, sc = pack_linear_split(w)
    dev = x.device
    dh = torch.from_numpy(hi.view(np.int16)).to(dev)
    dl = torch.from_numpy(lo.view(np.int16)).to(dev)
    ds = torch.tensor([sc], dtype=torch.float32, device=dev)
    q = torch.zeros((2, nseq, 4, R, 64), dtype=torch.int16, device=dev)
    k = torch.zeros((2, nseq, 4, R, 64), dtype=torch.int16, device=dev)
    v = torch.zeros((2, nseq, 4, 64, R), dtype=torch.int16, device=dev)
    cnt = cnt.to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        hd.check(hd.lib.imcui_hip_qkv_split_f32(hd.h, _ptr(x), _ptr(dh), _ptr(dl), _ptr(ds), _ptr(bias), _ptr(cos), _ptr(sin), _ptr(cnt), nseq, R,
                                                float(alpha), int(cross), _ptr(q), _ptr(k), _ptr(v), _stream_ptr()), "qkv_split")
    return q, k, v


def pack_ffn_w2(w2: torch.Tensor):
    """Host: ffn.3 weight [256, 512] -> planes in the K order of the fused FFN kernel, and the inverse scale."""
    from .lib_loader import load_library

    lib = load_library()
    wh = _as_f32_host(w2)
    assert wh.shape == (256, 512)
    hi = np.zeros(256 * 512, dtype=np.uint16)
    lo = np.zeros(256 * 512, dtype=np.uint16)
    sc = lib.imcui_hip_ffn_pack_w2(wh.ctypes.data, hi.ctypes.data, lo.ctypes.data)
    if sc <= 0:
        raise ImcuiHipError("ffn_pack_w2 failed")
    return hi, lo, float(sc)


class FusedFFN:
    """Building block: x + W2 GELU(LN(W1 [x | ctx] + b1)) + b2 in ONE kernel (csrc/ffn.hip), weights packed once.
    gamma = beta = None: ReLU instead of LayerNorm + GELU (SuperGlue's MLP with the BatchNorm folded into W1, b1).
    act = 2 / 3: x + LN_256(W2 act(W1 [x | ctx] + b1) + b2) with LeakyReLU(0.01) / ReLU (gamma, beta [256]): the dense matchers."""

    def __init__(self, w1, b1, gamma, beta, w2, b2, device, act=None):
        self.act = act if act is not None else (1 if gamma is None else 0)
        if gamma is None:
            gamma = beta = torch.zeros(512)
        def up(a):
            return torch.from_numpy(a.view(np.int16)).to(device)

        h1, l1, s1 = pack_linear_split(w1)
        h2, l2, s2 = pack_ffn_w2(w2)
        self.w1h, self.w1l, self.w2h, self.w2l = up(h1), up(l1), up(h2), up(l2)
        self.s1 = torch.tensor([s1], dtype=torch.float32, device=device)
        self.s2 = torch.tensor([s2], dtype=torch.float32, device=device)
        self.b1, self.gamma, self.beta, self.b2 = (t.detach().float().contiguous().to(device) for t in (b1, gamma, beta, b2))

    def __call__(self, x: torch.Tensor, ctx: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        hd = get_handle(x.device)
        x, ctx = x.contiguous().float(), ctx.contiguous().float()
        M = x.shape[0]
        out = torch.empty_like(x) if out is None else out
        with torch.cuda.device(x.device):
            hd.check(
                hd.lib.imcui_hip_ffn_split_f32(hd.h, _ptr(x), _ptr(ctx), _ptr(self.w1h), _ptr(self.w1l), _ptr(self.s1), _ptr(self.b1), _ptr(self.gamma),
                                               _ptr(self.beta), _ptr(self.w2h), _ptr(self.w2l), _ptr(self.s2), _ptr(self.b2), _ptr(out), M, self.act, _stream_ptr()),
                "ffn_split",
            )
        return out


def conv3x3_f32(x_nhwc: torch.Tensor, w_oihw: torch.Tensor, bias: torch.Tensor, relu=True, pool=False) -> torch.Tensor:
    hd = get_handle(x_nhwc.device)
    lib = hd.lib
    B, H, W, Cin = x_nhwc.shape
    Cout = w_oihw.shape[0]
    wh = _as_f32_host(w_oihw)
    x_nhwc = x_nhwc.contiguous().float()
    bias = bias.contiguous().float().to(x_nhwc.device)
    out = torch.empty((B, H // 2, W // 2, Cout) if pool else (B, H, W, Cout), dtype=torch.float32, device=x_nhwc.device)
    if lib.imcui_hip_get_precision(hd.h) == 1:
        hi = np.zeros(Cout * Cin * 9, dtype=np.uint16)
        lo = np.zeros(Cout * Cin * 9, dtype=np.uint16)
        sc = lib.imcui_hip_conv3x3_pack_split(wh.ctypes.data, Cout, Cin, hi.ctypes.data, lo.ctypes.data)
        if sc == 0.0:
            raise ImcuiHipError("conv3x3_pack_split failed")
        dh = torch.from_numpy(hi.view(np.int16)).to(x_nhwc.device)
        dl = torch.from_numpy(lo.view(np.int16)).to(x_nhwc.device)
        ds = torch.tensor([sc], dtype=torch.float32, device=x_nhwc.device)
        with torch.cuda.device(x_nhwc.device):
            hd.check(
                lib.imcui_hip_conv3x3_split_f32(hd.h, _ptr(x_nhwc), _ptr(dh), _ptr(dl), _ptr(ds), _ptr(bias), _ptr(out), B, H, W, Cin, Cout, int(relu), int(pool), _stream_ptr()),
                "conv3x3_split",
            )
        return out
    packed = np.zeros(Cout * Cin * 9, dtype=np.float32)
    if lib.imcui_hip_conv3x3_pack(wh.ctypes.data, Cout, Cin, packed.ctypes.data) != 0:
        raise ImcuiHipError("conv3x3_pack failed")
    wp = torch.from_numpy(packed).to(x_nhwc.device)
    with torch.cuda.device(x_nhwc.device):
        hd.check(
            lib.imcui_hip_conv3x3_f32(hd.h, _ptr(x_nhwc), _ptr(wp), _ptr(bias), _ptr(out), B, H, W, Cin, Cout, int(relu), int(pool), _stream_ptr()),
            "conv3x3",
        )
    return out


def _split_planes(t: torch.Tensor) -> torch.Tensor:
    """f32 tensor -> [2, ...] f16 (hi, lo) planes, the operand format of the split attention kernel."""
    hi = t.half()
    lo = (t - hi.float()).half()
    return torch.stack([hi, lo], 0).contiguous()


def attention_f32(q, k, v, cnt, cross=False, log2_domain=False):
    """q,k,v [S,heads,rows,64] head-major (q pre-scaled); cnt [S] int32 -> [S*rows, heads*64].
    log2_domain: q is multiplied by log2(e) here and the kernel's base-2 soft-max path (the one the layers use) runs."""
    hd = get_handle(q.device)
    S, Hh, R, d = q.shape
    assert d == 64
    if log2_domain:
        q = q.float() * 1.4426950408889634
    if hd.lib.imcui_hip_get_precision(hd.h) == 1:
        # the split kernel consumes pre-split f16 planes of Q, K and V^T [S,heads,64,rows]
        q, k, v = _split_planes(q.float()), _split_planes(k.float()), _split_planes(v.float().transpose(2, 3).contiguous())
    else:
        q, k, v = q.contiguous().float(), k.contiguous().float(), v.contiguous().float()
    o = torch.zeros((S * R, Hh * 64), dtype=torch.float32, device=q.device)
    cnt = cnt.to(torch.int32).contiguous()
    with torch.cuda.device(q.device):
        hd.check(
            hd.lib.imcui_hip_attention_f32(hd.h, _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(cnt), S, Hh, R, int(cross), int(bool(log2_domain)), _stream_ptr()),
            "attention",
        )
    return o


def attention_mx_f32(q, k, v, cnt, cross=False):
    """`attention_f32(..., log2_domain=True)` in attention variant 9 (csrc/attention_mx.hip: P.V as one f16 product + two block-scaled fp6
    correction products); split arithmetic only."""
    hd = get_handle(q.device)
    S, Hh, R, d = q.shape
    assert d == 64
    q = _split_planes(q.float() * 1.4426950408889634)
    k, v = _split_planes(k.float()), _split_planes(v.float().transpose(2, 3).contiguous())
    o = torch.zeros((S * R, Hh * 64), dtype=torch.float32, device=q.device)
    cnt = cnt.to(torch.int32).contiguous()
    nbytes = hd.lib.imcui_hip_attention_mx_scratch_bytes(S, Hh, R)
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    with torch.cuda.device(q.device):
        hd.check(hd.lib.imcui_hip_attention_mx_f32(hd.h, _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(cnt), S, Hh, R, int(cross), _ptr(scratch), nbytes, _stream_ptr()), "attention (variant 9)")
    return o


def simple_nms(scores: torch.Tensor, radius: int) -> torch.Tensor:
    hd = get_handle(scores.device)
    scores = scores.contiguous().float()
    B, H, W = scores.shape
    out = torch.empty_like(scores)
    with torch.cuda.device(scores.device):
        hd.check(hd.lib.imcui_hip_simple_nms(hd.h, _ptr(scores), _ptr(out), B, H, W, int(radius), _stream_ptr()), "simple_nms")
    return out
