"""One data-parallel training step of the S128 network (fused trainer: noise -> forward -> loss -> backward -> clip/Adam/EMA),
timed with the host clock around synchronised steps; the program rocprofv3 traces for profiles/r3_train_*.

    python tools/train_step_run.py [steps] [B]          MCEDM_TRAIN_GRAPH=0: eager launches
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mcedm_amd  # noqa: E402,F401
from mcedm_amd import lib  # noqa: E402
from mcedm_amd.train import FlatTrainState  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
wl = bench.WORKLOADS["s128"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else wl["batch"]
dev = torch.device("cuda", 0)
plan = lib.Plan(2, 2, 2, wl["ch"], wl["ch_mult"], 1, wl["attn"], 128)
params = bench.synth_params(plan, 7, dev)
cond, mask, _ = bench.synth_inputs(B, wl["H"], wl["W"], 1000, dev)
gen = torch.Generator(device="cpu").manual_seed(7)
xs = torch.randn(B, 2, wl["H"], wl["W"], generator=gen).to(dev)
nz = torch.randn(B, 2, wl["H"], wl["W"], generator=gen).to(dev)
rn = torch.randn(B, generator=gen).to(dev)
ts = FlatTrainState(plan, params)
ts.step(xs, cond, mask, nz, rn)
ts.step(xs, cond, mask, nz, rn)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = ts.step(xs, cond, mask, nz, rn)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
flops = 3 * 70.843e9 * B          # forward + dgrad + wgrad, SURVEY.md 8d
print(json.dumps({"train_step_ms": ms, "batch": B, "graph": ts._graph is not None and ts._graph[1] is not None,
                  "samples_per_s": B / ms * 1e3, "fp32_frac": flops / (ms * 1e-3) / 1e12 / bench.PEAK_FP32_MFMA_TFLOPS,
                  "loss": float(loss)}))
