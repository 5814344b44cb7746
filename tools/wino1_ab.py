"""conv_wino1_kernel (one wave per SIMD) against conv_wino_kernel<WinoCfg<4>> (two): bit-identity and time, one launch each.
    python tools/wino1_ab.py [--B 32 --cin 128 --hw 128 --res 0|1|2 --up 0|1]"""
import argparse, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("m-cedm_amd.lib")
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--cin", type=int, default=128)
ap.add_argument("--cout", type=int, default=128)
ap.add_argument("--hw", type=int, default=128)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--res", type=int, default=0)       # residual mode: 0 same size, 1 up, 2 down
ap.add_argument("--up", type=int, default=0)        # input nearest-upsampled
a = ap.parse_args()
torch.manual_seed(0)
hs = a.hw // 2 if a.up else a.hw
x = torch.randn(a.B, a.cin, hs, hs, device="cuda")
w = torch.randn(a.cout, a.cin, 3, 3, device="cuda") / (a.cin * 9) ** 0.5
b = torch.randn(a.cout, device="cuda") * 0.1
rhw = a.hw // 2 if a.res == 1 else a.hw * 2 if a.res == 2 else a.hw
res = torch.randn(a.B, a.cout, rhw, rhw, device="cuda")
coef = torch.stack([torch.randn(a.B, a.cin) * 0.1, 1 + 0.1 * torch.randn(a.B, a.cin), 0.1 * torch.randn(a.B, a.cin),
                    torch.zeros(a.B, a.cin)], -1).cuda()
wino = lib.op_pack_conv_wino(w)
out = torch.empty(a.B, a.cout, a.hw, a.hw, device="cuda")
flops = 2.0 * a.B * a.hw * a.hw * a.cout * a.cin * 9


def run():
    return lib.op_conv_wino(x, None, wino, b, a.cout, coef=coef, act=1, resample=1 if a.up else 0, res=res, res_mode=a.res, out=out)


def timed(name):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"{name:22s} B={a.B} {a.cin}->{a.cout} {a.hw}x{a.hw} up={a.up} res={a.res}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:6.1f} algorithmic TFLOP/s", flush=True)
    return out.clone()


lib.set_conv_wino1(0)
y0 = timed("two waves per SIMD")
lib.set_conv_wino1(1)
y1 = timed("one wave per SIMD")
lib.set_conv_wino1(-1)
print("bit-identical:", bool(torch.equal(y0, y1)), " max |d| =", float((y0 - y1).abs().max()), " finite:", bool(torch.isfinite(y1).all()))
