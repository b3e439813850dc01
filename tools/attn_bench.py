"""Micro-benchmark of the split attention kernels at the bench shape (64 sequences x 4 heads x 2048 rows, head_dim 64):
HIP-event time per launch and algorithmic / executed TFLOP/s.
    python tools/attn_bench.py [cross] [natural]     (default: the base-2 path the LightGlue / SuperGlue layers use)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "image-matching-webui_amd")]
import torch  # noqa: E402

from imcui_hip import backend  # noqa: E402

dev = torch.device("cuda:0")
S, Hh, R = 64, 4, 2048
cross = "cross" in sys.argv[1:]
l2d = "natural" not in sys.argv[1:]
g = torch.Generator().manual_seed(0)
q = (torch.randn(S, Hh, R, 64, generator=g) * 0.5).to(dev)
k = torch.randn(S, Hh, R, 64, generator=g).to(dev)
v = torch.randn(S, Hh, R, 64, generator=g).to(dev)
cnt = torch.full((S,), R, dtype=torch.int32, device=dev)
qs, ks, vs = backend._split_planes(q * (1.4426950408889634 if l2d else 1.0)), backend._split_planes(k), backend._split_planes(v.transpose(2, 3).contiguous())
hd = backend.get_handle(dev)
o = torch.zeros((S * R, Hh * 64), dtype=torch.float32, device=dev)


def run():
    hd.check(hd.lib.imcui_hip_attention_f32(hd.h, backend._ptr(qs), backend._ptr(ks), backend._ptr(vs), backend._ptr(o), backend._ptr(cnt), S, Hh, R, int(cross),
                                            int(l2d), backend._stream_ptr()), "attention")


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 30
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
flops = S * Hh * 2 * 2 * R * R * 64
kk = k[[1, 0]] if cross else k[:2]
vv = v[[1, 0]] if cross else v[:2]
att = torch.softmax(q[:2].double() @ kk.double().transpose(-1, -2), -1) @ vv.double()
err = (o.view(S, R, Hh, 64)[:2].permute(0, 2, 1, 3) - att.float()).abs().max().item()
print(f"base2={int(l2d)} cross={int(cross)}: {ms * 1e3:.1f} us / launch, {flops / ms / 1e9:.1f} algorithmic TFLOP/s, "
      f"{3 * flops / ms / 1e9:.1f} executed (3 MFMAs / product) = {3 * flops / ms / 1e9 / 2500:.3f} of the 2.5 PF f16 peak; max error vs f64 {err:.2e}")
