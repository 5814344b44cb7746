"""In-kernel timeline of the Winograd conv: per-workgroup stamps (start, prologue end, K loop end, end).
    MCEDM_WINO_MODE=1 python tools/wino_timeline.py [B cin hw [cout]]      (cout 64: the 256-thread WinoCfg<2> variant)"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("m-cedm_amd.lib")
B, cin, hw = (int(v) for v in (sys.argv[1:4] + ["32", "128", "128"][len(sys.argv) - 1:]))
cout = int(sys.argv[4]) if len(sys.argv) > 4 else 128
x = torch.randn(B, cin, hw, hw, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
b = torch.randn(cout, device="cuda")
res = torch.randn(B, cout, hw, hw, device="cuda")
coef = torch.stack([torch.zeros(B, cin), torch.ones(B, cin), torch.zeros(B, cin), torch.zeros(B, cin)], -1).cuda()
wino = lib.op_pack_conv_wino(w)
out = torch.empty(B, cout, hw, hw, device="cuda")
run = lambda: lib.op_conv_wino(x, None, wino, b, cout, coef=coef, act=1, res=res, out=out)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
nb = B * (hw // 8) * (hw // 16)
dbg = torch.zeros(nb * 16, dtype=torch.int64, device="cuda")
l = lib._bind_ops()
l.mcedm_op_set_conv_debug.argtypes = [C.c_void_p]
l.mcedm_op_set_conv_debug(dbg.data_ptr())
run()
torch.cuda.synchronize()
l.mcedm_op_set_conv_debug(None)
d = dbg.cpu().numpy().reshape(nb, 16)
d = d[d[:, 0] != 0]
per = int(d[0, 4])
t0 = d[:, 0].min()
st, pro, loop, end = [(d[:, i] - t0) / 100.0 for i in range(4)]
flops = 2.0 * B * hw * hw * cout * cin * 9
cyc = np.median(d[:, 6] - d[:, 5]) / per
clk = np.median((d[:, 6] - d[:, 5]) / ((d[:, 2] - d[:, 1]) * 10e-9) / 1e9)
nch = cin // 8
print(f"B={B} {cin}->{cout} {hw}x{hw}: {us:.1f} us/launch ({flops / us / 1e6:.1f} algorithmic TFLOP/s), {len(d)} workgroups x {per} tiles, "
      f"span {end.max():.1f} us | prologue {np.mean(pro - st):.2f} us | {np.mean(loop - pro) / per:.2f} us per tile (K loop + epilogue)")
print(f"   {cyc:.0f} cycles per tile = {cyc / nch:.0f} per chunk incl. the epilogue (matrix floor 4096 per chunk and SIMD), clock {clk:.2f} GHz")
if os.environ.get("MCEDM_WINO1", "1") != "0" and cout % 128 == 0:
    print(f"   one wave per SIMD: K loop {np.mean(d[:, 8]) / per / nch:.0f} cycles per chunk, epilogue + accumulator init {np.mean(d[:, 9]) / per:.0f} cycles per tile "
          f"(= {np.mean(d[:, 9]) / per / nch:.0f} per chunk)")
elif int(os.environ.get("MCEDM_WINO_MODE", "0")) & 16:
    nst = nch // 2
    print("   wave 0 cycles per slot of a stage (2 chunks = 8 slots of 8 MFMAs; floor 512 alone / 1024 with the SIMD's other wave): "
          + "  ".join(f"s{j} {np.mean(d[:, 8 + j]) / nst / per:.0f}" for j in range(8)))
elif int(os.environ.get("MCEDM_WINO_MODE", "0")) & 32:
    print("   wave 0 cycles per TILE in the epilogue: " + "  ".join(f"{nm} {np.mean(d[:, 8 + j]) / per:.0f}" for j, nm in enumerate(
        ["nu-transform + requests", "exchange rounds", "stores", "statistics", "accumulator init"])))
elif d[:, 8:13].max() > 0:
    print("   wave 0 cycles per chunk: " + "  ".join(f"{nm} {np.mean(d[:, 8 + j]) / nch / per:.0f}" for j, nm in enumerate(["top", "mfma stream", "barrier", "epilogue"])))
    if d[:, 13:16].max() > 0:      # the SIMD partner of wave 0 (wave MB: the same channel block, the other row half)
        print("   its SIMD partner:        " + "  ".join(f"{nm} {np.mean(d[:, c]) / nch / per:.0f}" for nm, c in (("top", 7), ("mfma stream", 13), ("barrier", 14), ("epilogue", 15))))
