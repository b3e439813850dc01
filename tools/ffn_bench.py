"""Time the fused LightGlue FFN kernel alone at the bench shape (64 x 2048 tokens) -- GPU box only.

    python tools/ffn_bench.py [M] [iters]

Prints the average launch time (torch events on the current stream, which is the stream the kernel is launched on) and the
executed matrix rate; IMCUI_FFN_VARIANT=0/1 selects the K-loop variant of csrc/ffn.hip.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "image-matching-webui_amd"))
import torch

from imcui_hip import backend

M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
w1 = torch.randn(512, 512, generator=g) / 512 ** 0.5
w2 = torch.randn(256, 512, generator=g) / 512 ** 0.5
ffn = backend.FusedFFN(w1, torch.zeros(512), torch.ones(512), torch.zeros(512), w2, torch.zeros(256), dev)
x = torch.randn(M, 256, device=dev)
ctx = torch.randn(M, 256, device=dev)
out = torch.empty_like(x)
for _ in range(3):
    ffn(x, ctx, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ffn(x, ctx, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
flop = 2.0 * M * (512 * 512 + 512 * 256) * 3
print(f"ffn variant={os.environ.get('IMCUI_FFN_VARIANT', '1')} M={M}: {ms * 1e3:.1f} us/launch, executed {flop / ms / 1e9:.0f} TFLOP/s, "
      f"HBM (x, ctx in; x out) {3 * M * 1024 / ms / 1e9:.2f} TB/s")

# phase breakdown from the per-workgroup wall-clock stamps (100 MHz)
hd = backend.get_handle(dev)
nwg = M // 128
stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
hd.check(hd.lib.imcui_hip_ffn_set_debug(hd.h, stamps.data_ptr()), "ffn_set_debug")
ffn(x, ctx, out=out)
torch.cuda.synchronize()
hd.check(hd.lib.imcui_hip_ffn_set_debug(hd.h, None), "ffn_set_debug")
t = stamps.view(nwg, 8).cpu().double()[:, :6] * 0.01  # us
names = ["GEMM 1 (16 K tiles)", "bias + LayerNorm + GELU", "hand-over + GEMM 2 first half", "hand-over + GEMM 2 second half", "output tile + residual store"]
d = t[:, 1:] - t[:, :-1]
print("per-workgroup phase times, us (mean / median / max over %d workgroups):" % nwg)
for i, n in enumerate(names):
    print(f"  {n:34s} {d[:, i].mean():7.2f} {d[:, i].median():7.2f} {d[:, i].max():7.2f}")
tot = t[:, 5] - t[:, 0]
print(f"  {'workgroup total':34s} {tot.mean():7.2f} {tot.median():7.2f} {tot.max():7.2f}")
print(f"  launch span {(t[:, 5].max() - t[:, 0].min()):.1f} us; first-wave start spread {(t[:, 0].median() - t[:, 0].min()):.1f} us")
