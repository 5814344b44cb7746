"""Finish-time spread of the persistent Winograd workgroups (one per CU): how much of a launch its fastest CUs idle.
    python tools/wino_spread.py [B cin hw]"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = importlib.import_module("m-cedm_amd.lib")
B, cin, hw = (int(v) for v in (sys.argv[1:4] + ["32", "128", "128"][len(sys.argv) - 1:]))
cout = 128
x = torch.randn(B, cin, hw, hw, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
b = torch.randn(cout, device="cuda")
res = torch.randn(B, cout, hw, hw, device="cuda")
coef = torch.stack([torch.zeros(B, cin), torch.ones(B, cin), torch.zeros(B, cin), torch.zeros(B, cin)], -1).cuda()
wino = lib.op_pack_conv_wino(w)
out = torch.empty(B, cout, hw, hw, device="cuda")
run = lambda: lib.op_conv_wino(x, None, wino, b, cout, coef=coef, act=1, res=res, out=out)
for _ in range(5):
    run()
torch.cuda.synchronize()
nb = B * (hw // 8) * (hw // 16)
l = lib._bind_ops()
l.mcedm_op_set_conv_debug.argtypes = [C.c_void_p]
for rep in range(3):
    dbg = torch.zeros(nb * 16, dtype=torch.int64, device="cuda")
    l.mcedm_op_set_conv_debug(dbg.data_ptr())
    run()
    torch.cuda.synchronize()
    l.mcedm_op_set_conv_debug(None)
    d = dbg.cpu().numpy().reshape(nb, 16)
    d = d[d[:, 0] != 0]
    t0 = d[:, 0].min()
    st, end = (d[:, 0] - t0) / 100.0, (d[:, 3] - t0) / 100.0
    dur = end - st
    q = lambda a: " ".join(f"{np.percentile(a, p):.1f}" for p in (0, 10, 50, 90, 100))
    print(f"{len(d)} workgroups x {int(d[0, 4])} tiles | start (us) p0/10/50/90/100: {q(st)} | end: {q(end)} | duration: {q(dur)} | "
          f"idle share of the launch = {1 - dur.sum() / (len(d) * end.max()):.3f}")
