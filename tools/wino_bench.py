"""Winograd vs direct 3x3 convolution, one launch each (the dominant shapes of the S128 workload).
    python tools/wino_bench.py [--B 32 --cin 128 --hw 128]"""
import argparse, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("m-cedm_amd.lib")
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--cin", type=int, default=128)
ap.add_argument("--cout", type=int, default=128)
ap.add_argument("--hw", type=int, default=128)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
torch.manual_seed(0)
x = torch.randn(a.B, a.cin, a.hw, a.hw, device="cuda")
w = torch.randn(a.cout, a.cin, 3, 3, device="cuda") / (a.cin * 9) ** 0.5
b = torch.randn(a.cout, device="cuda") * 0.1
res = torch.randn(a.B, a.cout, a.hw, a.hw, device="cuda")
coef = torch.stack([torch.randn(a.B, a.cin) * 0.1, 1 + 0.1 * torch.randn(a.B, a.cin), 0.1 * torch.randn(a.B, a.cin),
                    torch.zeros(a.B, a.cin)], -1).cuda()
wpk, bpk = lib.op_pack_conv(w, b)
wino = lib.op_pack_conv_wino(w)
out = torch.empty(a.B, a.cout, a.hw, a.hw, device="cuda")
flops = 2.0 * a.B * a.hw * a.hw * a.cout * a.cin * 9


def timed(fn, name):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"{name:10s} B={a.B} {a.cin}->{a.cout} {a.hw}x{a.hw}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:6.1f} algorithmic TFLOP/s", flush=True)
    return out.clone()


d = timed(lambda: lib.op_conv(x, None, wpk, bpk, a.cout, 3, coef=coef, act=1, res=res, out=out), "direct")
y = timed(lambda: lib.op_conv_wino(x, None, wino, b, a.cout, coef=coef, act=1, res=res, out=out), "winograd")
print("max |winograd - direct| =", float((y - d).abs().max()), " max |direct| =", float(d.abs().max()))
