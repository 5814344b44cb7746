"""Sweep the conv tile configurations on the low-resolution shapes of the bench workloads (A/B inside one process)."""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcedm_amd  # noqa
from mcedm_amd import lib

TILES = [(128, 8, 32), (64, 8, 32), (32, 8, 32), (128, 16, 16), (64, 16, 16), (32, 16, 16), (128, 8, 16), (64, 8, 16), (128, 8, 8), (64, 8, 8), (32, 8, 8)]
SHAPES = [  # B, cin, cout, hw, k
    (32, 128, 128, 64, 3), (32, 128, 128, 32, 3), (32, 256, 128, 32, 3), (32, 128, 128, 16, 3), (32, 256, 128, 16, 3),
    (32, 256, 128, 32, 1), (32, 128, 384, 16, 1), (32, 128, 128, 16, 1),
    (64, 64, 64, 32, 3), (64, 128, 64, 32, 3), (64, 64, 64, 16, 3), (64, 64, 64, 8, 3), (64, 128, 64, 8, 3), (64, 64, 192, 8, 1),
]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    SHAPES = [(32, 128, 128, 128, 3), (32, 256, 128, 128, 3), (32, 128, 128, 64, 3), (32, 256, 128, 64, 3), (32, 256, 128, 128, 1),
              (32, 64, 64, 128, 3), (32, 128, 64, 128, 3)]
    TILES = [(128, 8, 32), (128, 16, 16), (128, 8, 16), (64, 8, 32), (64, 16, 16), (64, 8, 16)]
for B, cin, cout, hw, k in SHAPES:
    x = torch.randn(B, cin, hw, hw, device="cuda")
    w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
    b = torch.randn(cout, device="cuda")
    coef = torch.stack([torch.zeros(B, cin), torch.ones(B, cin), torch.zeros(B, cin), torch.zeros(B, cin)], -1).cuda()
    res = torch.randn(B, cout, hw, hw, device="cuda")
    wpk, bpk = lib.op_pack_conv(w, b)
    out = torch.empty(B, cout, hw, hw, device="cuda")
    flops = 2.0 * B * hw * hw * cout * cin * k * k
    line = []
    for tile in [None] + TILES:
        if tile and ((cout + 31) // 32 * 32) % tile[0]:
            continue
        lib.set_conv_tile(*(tile or (0, 0, 0)))
        try:
            for _ in range(2):
                lib.op_conv(x, None, wpk, bpk, cout, k, coef=coef, act=1, res=res, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.op_conv(x, None, wpk, bpk, cout, k, coef=coef, act=1, res=res, out=out)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            line.append(f"{'auto' if not tile else 'x'.join(map(str, tile))}:{flops / ms / 1e9:5.1f}")
        finally:
            lib.set_conv_tile()
    print(f"B{B} {cin}->{cout} {hw}^2 k{k}: " + "  ".join(line), flush=True)
