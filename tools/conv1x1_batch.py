import os, sys
import torch
sys.path.insert(0, "/root/repo")
import mcedm_amd  # noqa
from mcedm_amd import lib
for B in (32, 8, 2):
  for cin, cout, hw in ((256, 128, 128), (128, 128, 128), (64, 128, 128), (256, 128, 64)):
    torch.manual_seed(0)
    x = torch.randn(B, cin, hw, hw, device="cuda"); w = torch.randn(cout, cin, 1, 1, device="cuda") / cin ** 0.5
    b = torch.randn(cout, device="cuda")
    wpk, bpk = lib.op_pack_conv(w, b); out = torch.empty(B, cout, hw, hw, device="cuda")
    run = lambda: lib.op_conv(x, None, wpk, bpk, cout, 1, out=out)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"B={B} {cin}->{cout} 1x1 @{hw}^2: {us:.1f} us = {2.0 * B * hw * hw * cin * cout / us / 1e6:.1f} TFLOP/s, {4.0 * B * hw * hw * (cin + cout) / us / 1e6:.2f} TB/s", flush=True)
