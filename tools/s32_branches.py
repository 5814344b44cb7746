"""Config 2 (S32: 32x32, ch = 64, B = 64) with the sampler graph captured as k parallel branches over batch shards:
each branch = the whole 18-step Heun call of B / k states on its own stream (fork at the start of the capture, join at its end),
its own workspace; replay time and bit-identity against the unbranched graph.
    python tools/s32_branches.py [workload=s32] [B]        DEBUG_HIP_FORCE_GRAPH_QUEUES=n can be set from outside"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mcedm_amd  # noqa: E402,F401
from mcedm_amd import lib  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "s32"
wl = bench.WORKLOADS[key]
B = int(sys.argv[2]) if len(sys.argv) > 2 else wl["batch"]
dev = torch.device("cuda", 0)
plan = lib.Plan(2, 2, 2, wl["ch"], wl["ch_mult"], 1, wl["attn"], 128)
params = bench.synth_params(plan, 7, dev)
packed = plan.pack(params)
cond, mask, init = bench.synth_inputs(B, wl["H"], wl["W"], 1000, dev)


class SP:
    timesteps, sigma_min, sigma_max, rho, S_churn, S_min, S_max, S_noise, w = 18, 0.002, 80.0, 7.0, 0.0, 0.0, float("inf"), 1.0, 0.0


sd = lib.sampler_desc(SP)
ref = None
for k in (1, 2, 4, 8):
    if B % k:
        continue
    sh = B // k
    wss = [lib.Workspace() for _ in range(k)]
    outs = [torch.empty((sh, 1, wl["H"], wl["W"], 2), dtype=torch.float64, device=dev) for _ in range(k)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(k - 1)]

    def run():
        cur = torch.cuda.current_stream(dev)
        for i in range(k):
            sl = slice(i * sh, (i + 1) * sh)
            if i == 0:
                plan.sample(packed, sd, cond[sl], mask[sl], init[sl], None, return_last=True, ws=wss[i], out=outs[i])
            else:
                s = streams[i - 1]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    plan.sample(packed, sd, cond[sl], mask[sl], init[sl], None, return_last=True, ws=wss[i], out=outs[i])
        for s in streams:
            cur.wait_stream(s)
    g = lib._capture(run, dev)
    g.replay(); torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    got = torch.cat(outs, 0)
    if ref is None:
        ref = got.clone()
    print(f"{key} B={B}: {k} branch(es) of {sh}: {dt * 1e3:8.2f} ms per call = {B / dt:8.1f} states/s; identical to one branch: {bool(torch.equal(got, ref))}", flush=True)
    del g
