import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcedm_amd  # noqa
from mcedm_amd import lib
B, cin, cout, hw = 32, 128, 128, 128
x = torch.randn(B, cin, hw, hw, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
b = torch.randn(cout, device="cuda"); res = torch.randn(B, cout, hw, hw, device="cuda")
coef = torch.stack([torch.zeros(B, cin), torch.ones(B, cin), torch.zeros(B, cin), torch.zeros(B, cin)], -1).cuda()
wpk, bpk = lib.op_pack_conv(w, b); out = torch.empty(B, cout, hw, hw, device="cuda")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6
t_end = time.time() + secs
n = 0
while time.time() < t_end:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): lib.op_conv(x, None, wpk, bpk, cout, 3, coef=coef, act=1, res=res, out=out)
    e1.record(); torch.cuda.synchronize(); n += 1
    if n % 5 == 0: print(f"[{time.time():.1f}] {e0.elapsed_time(e1) / 100:.4f} ms/launch", flush=True)
