"""Micro-benchmark of one fused conv launch (the dominant kernel of the S128 workload) for rocprofv3 PMC passes.

    python tools/conv_micro.py [--B 32 --cin 128 --cout 128 --hw 128 --k 3 --iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcedm_amd  # noqa: F401,E402
from mcedm_amd import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--cin", type=int, default=128)
ap.add_argument("--cout", type=int, default=128)
ap.add_argument("--hw", type=int, default=128)
ap.add_argument("--k", type=int, default=3)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--tile", type=int, nargs=3, default=None)
a = ap.parse_args()
torch.manual_seed(0)
x = torch.randn(a.B, a.cin, a.hw, a.hw, device="cuda")
w = torch.randn(a.cout, a.cin, a.k, a.k, device="cuda") / (a.cin * a.k * a.k) ** 0.5
b = torch.randn(a.cout, device="cuda") * 0.1
res = torch.randn(a.B, a.cout, a.hw, a.hw, device="cuda")
coef = torch.stack([torch.randn(a.B, a.cin) * 0.1, 1 + 0.1 * torch.randn(a.B, a.cin), 0.1 * torch.randn(a.B, a.cin),
                    torch.zeros(a.B, a.cin)], -1).cuda()
wpk, bpk = lib.op_pack_conv(w, b)
out = torch.empty(a.B, a.cout, a.hw, a.hw, device="cuda")
if a.tile:
    lib.set_conv_tile(*a.tile)
for _ in range(3):
    lib.op_conv(x, None, wpk, bpk, a.cout, a.k, coef=coef, act=1, res=res, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    lib.op_conv(x, None, wpk, bpk, a.cout, a.k, coef=coef, act=1, res=res, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
flops = 2.0 * a.B * a.hw * a.hw * a.cout * a.cin * a.k * a.k
byts = 4.0 * (a.B * a.hw * a.hw * (a.cin + 2 * a.cout) + a.cout * a.cin * a.k * a.k)
print(f"conv B={a.B} {a.cin}->{a.cout} {a.hw}x{a.hw} k={a.k}: {ms:.4f} ms  {flops / ms / 1e9:.1f} TFLOP/s  "
      f"{byts / ms / 1e6:.1f} GB/s algorithmic")
