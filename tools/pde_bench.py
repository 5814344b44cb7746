"""HBM throughput of the PDE residual kernels (csrc/pde.hip) on batches large enough to leave the caches.

    python tools/pde_bench.py        # prints algorithmic GB/s (SWE: 24 B / cell, Darcy: 12 B / output cell)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mcedm_amd  # noqa: F401,E402
from mcedm_amd import pde_loss  # noqa: E402


class Norm:
    def __init__(self, d):
        self.subtract, self.divide = torch.tensor(0.0), torch.tensor(d).cuda()


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in (32, 4096):
    T = X = 128
    s = torch.rand(B, T, X, 2, device="cuda") + 1.0
    gt = s.clone()
    f = pde_loss.SweFvLoss(Tn=0.128, x_min=-0.5, x_max=0.5)
    nh, nu = Norm(0.4), Norm(0.2)
    ms = timeit(lambda: f(s, gt, nh, nu))
    print(f"swe_fv_residual  B={B:5d} 128x128: {ms * 1e3:8.1f} us  {B * T * X * 24 / ms / 1e6:8.1f} GB/s algorithmic")
    d = pde_loss.DarcyLoss()
    ms = timeit(lambda: d(s, s, None, None))
    print(f"darcy_residual   B={B:5d} 128x128: {ms * 1e3:8.1f} us  {B * (T - 4) * (X - 4) * 12 / ms / 1e6:8.1f} GB/s algorithmic")
