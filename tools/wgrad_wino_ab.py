"""Winograd F(3x3, 2x2) weight-gradient kernel (csrc/wgrad_wino.hip) against the direct split-K kernel on the S128 training shapes:
per-kernel time from the library's event profiler (GEMM kernel and split reduction separately), agreement of the two results.
    python tools/wgrad_wino_ab.py [--B 32 --iters 10] [--shapes 128,128,128 256,128,128 ...]   (cin,cout,hw)"""
import argparse, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("m-cedm_amd.lib")
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--shapes", nargs="*", default=["128,128,128", "256,128,128", "128,128,64", "256,128,64", "128,128,32", "256,128,32"])
a = ap.parse_args()
torch.manual_seed(0)
for sh in a.shapes:
    cin, cout, hw = map(int, sh.split(","))
    x = torch.randn(a.B, cin, hw, hw, device="cuda")
    dy = torch.randn(a.B, cout, hw, hw, device="cuda")
    flops = 2.0 * a.B * hw * hw * cout * cin * 9
    res = {}
    for mode, name in ((0, "direct"), (1, "winograd")):      # (MCEDM_WGRAD_WINO64=0: the 64-channel form off)
        lib.set_wgrad_wino(mode)
        for _ in range(2):
            dw, db = lib.op_conv_wgrad(dy, x, None, 3)
        torch.cuda.synchronize()
        lib.prof_enable(True)
        for _ in range(a.iters):
            dw, db = lib.op_conv_wgrad(dy, x, None, 3)
        torch.cuda.synchronize()
        rows = lib.prof_report()
        lib.prof_enable(False)
        res[name] = (dw.clone(), db.clone())
        parts = []
        tot = 0.0
        for r in rows:
            us = r["total_ms"] / r["launches"] * 1e3
            tot += us
            parts.append(f"{r['name'].split('<')[0]} {us:.1f}")
        print(f"B={a.B} {cin}->{cout} {hw}x{hw} {name:9s}: {tot:8.1f} us  ({flops / tot / 1e6:6.1f} algorithmic TFLOP/s)   " + " | ".join(parts), flush=True)
    lib.set_wgrad_wino(-1)
    d = (res["direct"][0] - res["winograd"][0]).abs().max().item()
    print(f"      max |dW_direct - dW_winograd| = {d:.3e} (max |dW| {res['direct'][0].abs().max().item():.3e}); db equal: "
          f"{bool(torch.allclose(res['direct'][1], res['winograd'][1], rtol=1e-5, atol=1e-3))}", flush=True)
