"""Diagnostic: where a SMALL conv launch spends its time (in-kernel timestamps of every workgroup + event timing).

    python tools/conv_small_timeline.py B cin cout hw k [res] [gn]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcedm_amd  # noqa
from mcedm_amd import lib

B, cin, cout, hw, k = (int(v) for v in sys.argv[1:6])
use_res = len(sys.argv) > 6 and sys.argv[6] == "1"
x = torch.randn(B, cin, hw, hw, device="cuda")
w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
b = torch.randn(cout, device="cuda")
res = torch.randn(B, cout, hw, hw, device="cuda") if use_res else None
coef = torch.stack([torch.zeros(B, cin), torch.ones(B, cin), torch.zeros(B, cin), torch.zeros(B, cin)], -1).cuda()
wpk, bpk = lib.op_pack_conv(w, b)
out = torch.empty(B, cout, hw, hw, device="cuda")
run = lambda: lib.op_conv(x, None, wpk, bpk, cout, k, coef=coef, act=1, res=res, out=out)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
nb = 8192
dbg = torch.zeros(nb * 16, dtype=torch.int64, device="cuda")
l = lib._bind_ops()
l.mcedm_op_set_conv_debug.argtypes = [C.c_void_p]
l.mcedm_op_set_conv_debug(dbg.data_ptr())
for _ in range(5):
    run()
torch.cuda.synchronize()
l.mcedm_op_set_conv_debug(None)
d = dbg.cpu().numpy().reshape(nb, 16)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
st, pro, loop, end = [(d[:, i] - t0) / 100.0 for i in range(4)]
flops = 2.0 * B * hw * hw * cout * cin * k * k
print(f"B={B} {cin}->{cout} {hw}x{hw} k={k} res={int(use_res)}: {us:.1f} us/launch back-to-back ({flops / us / 1e6:.1f} TFLOP/s), "
      f"{len(d)} workgroups, in-kernel span {end.max():.1f} us | start spread {st.max():.1f} | prologue {np.mean(pro - st):.1f} "
      f"| K loop {np.mean(loop - pro):.1f} | epilogue {np.mean(end - loop):.1f}")
cyc = np.median(d[:, 6] - d[:, 5])
print(f"   K loop: {cyc:.0f} shader cycles (median), clock {np.median((d[:, 6] - d[:, 5]) / ((d[:, 2] - d[:, 1]) * 10e-9) / 1e9):.2f} GHz")
if d[:, 8:13].max() > 0 and d[:, 8:13].max() < (1 << 40):      # MCEDM_CONV_TIMELINE build: per-phase cycle sums of wave 0
    nch = -(-cin // (8 if k == 3 else 16))
    clk = np.median((d[:, 6] - d[:, 5]) / ((d[:, 2] - d[:, 1]) * 10e-9) / 1e9)
    print(f"   shader clock {clk:.2f} GHz; per-iteration cycles ({nch} iterations): " +
          "  ".join(f"{nm} {np.mean(d[:, 8 + j]) / nch:.0f}" for j, nm in enumerate(["commit|dma-issue", "barrier1|mfma", "load-issue|wait", "mfma|barrier", "barrier2|-"])))
    if d[:, 13:16].max() > 0:      # resident kernel: prologue phases
        print("   prologue cycles: " + "  ".join(f"{nm} {np.mean(d[:, j]):.0f}" for nm, j in [("dma+issue", 13), ("init_acc+coef rows+sync", 14), ("commit", 15), ("wait+sync", 7)]))
