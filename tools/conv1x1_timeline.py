"""In-kernel phase counters of the 1x1 conv at 128^2 (decoder skip projection 256 -> 128): needs the -DMCEDM_CONV_TIMELINE build
    tools/build_ab.sh ctl "-DMCEDM_CONV_TIMELINE" conv_mfma.hip;  MCEDM_LIB=m-cedm_amd/_ab/ctl.so python tools/conv1x1_timeline.py [cin cout hw]"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcedm_amd  # noqa
from mcedm_amd import lib
B = 32
cin, cout, hw = (int(v) for v in (sys.argv[1:4] + ["256", "128", "128"][len(sys.argv) - 1:]))
x = torch.randn(B, cin, hw, hw, device="cuda"); w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
b = torch.randn(cout, device="cuda")
wpk, bpk = lib.op_pack_conv(w, b); out = torch.empty(B, cout, hw, hw, device="cuda")
run = lambda: lib.op_conv(x, None, wpk, bpk, cout, 1, out=out)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
print(f"{cin}->{cout} 1x1 @{hw}^2 B={B}: {us:.1f} us = {2.0 * B * hw * hw * cin * cout / us / 1e6:.1f} TFLOP/s, {4.0 * B * hw * hw * (cin + cout) / us / 1e6:.2f} TB/s")
nb = B * (hw // 8) * (hw // 32) * max(1, cout // 128)
dbg = torch.zeros(nb * 16 * 4, dtype=torch.int64, device="cuda")
l = lib._bind_ops(); l.mcedm_op_set_conv_debug.argtypes = [C.c_void_p]
l.mcedm_op_set_conv_debug(dbg.data_ptr())
for _ in range(4): run()
torch.cuda.synchronize()
l.mcedm_op_set_conv_debug(None)
d = dbg.cpu().numpy().reshape(-1, 16)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
st, pro, loop, end = [(d[:, i] - t0) / 100.0 for i in range(4)]
print(f"{len(d)} workgroups, span {end.max():.1f} us | prologue {np.mean(pro - st):.2f} us | K loop {np.mean(loop - pro):.2f} us | epilogue {np.mean(end - loop):.2f} us")
nch = cin // 16
clk = np.median((d[:, 6] - d[:, 5]) / ((d[:, 2] - d[:, 1]) * 10e-9) / 1e9)
print(f"clock {clk:.2f} GHz; per-chunk cycles (wave 0): " + "  ".join(f"{nm} {np.mean(d[:, 8 + k]) / nch:.0f}" for k, nm in enumerate(["commit", "barrier1", "load-issue", "mfma-loop", "barrier2"])) + f"   (MFMAs of a chunk: {64 * 64} cycles per wave)")
