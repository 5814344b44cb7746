"""Diagnostic: how much of a conv launch is workgroup TURNOVER?  Every workgroup stamps its start / end (10 ns clock) and its
XCC / CU id; per CU the workgroups are sorted by start time and the idle time between one workgroup's end and the next
start in the same resident slot is summed.

    python tools/conv_slot_gaps.py B cin cout hw [res]
"""
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcedm_amd  # noqa
from mcedm_amd import lib

B, cin, cout, hw = (int(v) for v in sys.argv[1:5])
use_res = len(sys.argv) > 5 and sys.argv[5] == "1"
x = torch.randn(B, cin, hw, hw, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5
b = torch.randn(cout, device="cuda")
res = torch.randn(B, cout, hw, hw, device="cuda") if use_res else None
coef = torch.stack([torch.zeros(B, cin), torch.ones(B, cin), torch.zeros(B, cin), torch.zeros(B, cin)], -1).cuda()
wpk, bpk = lib.op_pack_conv(w, b)
out = torch.empty(B, cout, hw, hw, device="cuda")
run = lambda: lib.op_conv(x, None, wpk, bpk, cout, 3, coef=coef, act=1, res=res, out=out)
for _ in range(3):
    run()
torch.cuda.synchronize()
nb = 8192
dbg = torch.zeros(nb * 16, dtype=torch.int64, device="cuda")
l = lib._bind_ops()
l.mcedm_op_set_conv_debug.argtypes = [C.c_void_p]
l.mcedm_op_set_conv_debug(dbg.data_ptr())
run()
torch.cuda.synchronize()
l.mcedm_op_set_conv_debug(None)
d = dbg.cpu().numpy().reshape(nb, 16)
d = d[d[:, 0] != 0]
t0 = d[:, 0].min()
start, end = (d[:, 0] - t0) / 100.0, (d[:, 3] - t0) / 100.0
cu = d[:, 4]                       # (xcc << 32) | HW_ID: CU / SE / SH bits identify the CU
hwid = cu & 0xffffffff
cu_key = (cu >> 32) * 4096 + ((hwid >> 8) & 0xf) + 16 * ((hwid >> 12) & 0x1) + 32 * ((hwid >> 13) & 0x7)   # cu_id, sh_id, se_id
per = collections.defaultdict(list)
for k, s, e in zip(cu_key, start, end):
    per[int(k)].append((s, e))
span = end.max()
busy_frac, gaps, nslots = [], [], []
for k, v in per.items():
    v.sort()
    # greedy slot assignment: a workgroup takes the slot that freed earliest before its start
    slots = []
    for s, e in v:
        best = None
        for i, fe in enumerate(slots):
            if fe <= s + 0.5 and (best is None or fe > slots[best]):
                best = i
        if best is None:
            slots.append(e)
        else:
            gaps.append(s - slots[best])
            slots[best] = e
    nslots.append(len(slots))
    busy_frac.append(sum(e - s for s, e in v) / (len(slots) * span))
dur = end - start
print(f"B={B} {cin}->{cout} {hw}x{hw} res={int(use_res)}: {len(d)} workgroups on {len(per)} CUs, slots/CU {np.mean(nslots):.2f}, span {span:.1f} us, "
      f"workgroup duration {dur.mean():.1f} +- {dur.std():.1f} us")
print(f"   slot busy fraction {np.mean(busy_frac):.3f}; turnover gap (end of one workgroup -> start of the next in its slot): "
      f"mean {np.mean(gaps):.1f} us, median {np.median(gaps):.1f}, p90 {np.percentile(gaps, 90):.1f}, n={len(gaps)}")
first = np.sort(start)[: sum(nslots)]
print(f"   first-round start ramp: last of the first {len(first)} workgroups starts at {first.max():.1f} us; last workgroup ends at {span:.1f}; "
      f"earliest end of a final-round workgroup {np.sort(end)[-sum(nslots):].min():.1f} us")
