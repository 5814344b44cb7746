"""Diagnostic: per-workgroup timeline of the dominant conv (timestamps written by the kernel itself)."""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcedm_amd  # noqa
from mcedm_amd import lib
B, cin, cout, hw = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 128, 128
x = torch.randn(B, cin, hw, hw, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
b = torch.randn(cout, device="cuda"); res = torch.randn(B, cout, hw, hw, device="cuda")
coef = torch.stack([torch.zeros(B, cin), torch.ones(B, cin), torch.zeros(B, cin), torch.zeros(B, cin)], -1).cuda()
wpk, bpk = lib.op_pack_conv(w, b); out = torch.empty(B, cout, hw, hw, device="cuda")
for _ in range(3): lib.op_conv(x, None, wpk, bpk, cout, 3, coef=coef, act=1, res=res, out=out)
nb = B * (hw // 8) * (hw // 32)
dbg = torch.zeros(nb * 16, dtype=torch.int64, device="cuda")
l = lib._bind_ops(); l.mcedm_op_set_conv_debug.argtypes = [C.c_void_p]
l.mcedm_op_set_conv_debug(dbg.data_ptr())
for _ in range(12):   # steady state (clocks, caches): every launch overwrites the records, the last one is read
    lib.op_conv(x, None, wpk, bpk, cout, 3, coef=coef, act=1, res=res, out=out)
torch.cuda.synchronize()
l.mcedm_op_set_conv_debug(None)
d = dbg.cpu().numpy().reshape(nb, 16)
d = d[d[:, 0] != 0]; nb = len(d)   # the 8-wave kernel launches fewer, larger workgroups
if os.path.isdir("gpurun_out"): np.save("gpurun_out/timeline.npy", d)
t0 = d[:, 0].min()
st, pro, loop, end = [(d[:, i] - t0) / 100.0 for i in range(4)]   # microseconds
print(f"kernel span {end.max():.1f} us; {nb} workgroups")
print(f"prologue   (start->first chunk): mean {np.mean(pro - st):.1f} us  p95 {np.percentile(pro - st, 95):.1f}")
print(f"K loop     : mean {np.mean(loop - pro):.1f} us  min {np.min(loop - pro):.1f} max {np.max(loop - pro):.1f}")
print(f"epilogue   : mean {np.mean(end - loop):.1f} us  p95 {np.percentile(end - loop, 95):.1f} max {np.max(end - loop):.1f}")
clk = (d[:, 6] - d[:, 5]) / ((d[:, 2] - d[:, 1]) * 10e-9) / 1e9
print(f"shader clock inside the K loop: median {np.median(clk):.3f} GHz (min {clk.min():.3f}, max {clk.max():.3f}); MFMA-bound K loop at that clock: {2 * (cin // 8) * 18432 / np.median(clk) / 1e3:.1f} us")
nch = cin // 8
print("per-chunk cycles (wave 0 of each WG): " + "  ".join(f"{nm} {np.mean(d[:, 8 + k]) / nch:.0f}" for k, nm in enumerate(["commit", "barrier1", "load-issue", "mfma-loop", "barrier2"])))
if os.environ.get("MCEDM_CONV8_SEG"):
    print("8-wave per-chunk cycles  group0 (MFMA first): " + "  ".join(f"{nm} {np.mean(d[:, 8 + k]) / nch:.0f}" for k, nm in enumerate(["pre", "mfma", "post", "barrier"])) + "   group1 (commit first): " + "  ".join(f"{nm} {np.mean(d[:, 12 + k]) / nch:.0f}" for k, nm in enumerate(["pre", "mfma", "post", "barrier"])))
elif d[:, 8:16].max() < (1 << 32) and d[0, 15] != 0:
    for i in (0, 1, 2):
        print("wave -> (simd, wave slot):", [((int(h) >> 4) & 3, int(h) & 15) for h in d[i, 8:16]])
order = np.argsort(st)
for k in (0, 255, 511, 512, 767, 1023, 1535, 2047):
    i = order[k]; print(f"  wg#{k:4d} by start: start {st[i]:7.1f} loop_begin {pro[i]:7.1f} loop_end {loop[i]:7.1f} end {end[i]:7.1f}  cu {d[i,4] & 0xffffffff:#x} xcc {d[i,4] >> 32}")
hist, edges = np.histogram(st, bins=12)
print("start-time histogram (us):", [f"{e:.0f}:{h}" for h, e in zip(hist, edges)])
