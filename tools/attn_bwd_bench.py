"""Attention backward (mcedm_op_attention_bwd) kernel by kernel: run under `rocprofv3 --kernel-trace --stats` for the per-kernel
split, or alone for the total from the library's event profiler.
    python tools/attn_bwd_bench.py [--B 32] [--shapes 1,32,32 2,16,16]      (heads,H,W)"""
import argparse, importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("m-cedm_amd.lib")
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--shapes", nargs="*", default=["1,32,32", "2,16,16"])
a = ap.parse_args()
torch.manual_seed(0)
for sh in a.shapes:
    heads, H, W = map(int, sh.split(","))
    qkv = torch.randn(a.B, 3 * 64 * heads, H, W, device="cuda")
    out = lib.op_attention(qkv, heads)
    da = torch.randn_like(out)
    for _ in range(2):
        lib.op_attention_bwd(qkv, out, da, heads)
    torch.cuda.synchronize()
    lib.prof_enable(True)
    for _ in range(a.iters):
        lib.op_attention_bwd(qkv, out, da, heads)
    torch.cuda.synchronize()
    rows = lib.prof_report()
    lib.prof_enable(False)
    T = H * W
    for r in rows:
        us = r["total_ms"] / r["launches"] * 1e3
        print(f"B={a.B} heads={heads} T={T}: {r['name']} {us:.1f} us  ({10.0 * a.B * heads * T * T * 64 / us / 1e6:.1f} TFLOP/s)", flush=True)
