#!/usr/bin/env python
"""bench.py -- denoised states/sec of the 18-step EDM Heun sampler (35 U-Net evaluations per state).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload s128|s32|ref128|s128l3] [--batch B]

`--gpus N` (N > 1) works as typed: before anything touches the GPU the process starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...`
as a child (one rank per GPU over RCCL) and exits with its status.  Launched by torch.distributed.run directly
(RANK / WORLD_SIZE in the environment) it is a rank.

A "step" is one pass of the hot path over one batch: ``sample_edm`` (models/mcedm.py:570-638) on B states per GPU,
inputs already resident in HBM.  The batch axis is sharded across ranks with no data-path collective, so
scaling is weak: per-GPU batch is fixed.  Rank 0 prints ONE JSON line.

What is timed, and how:
  * headline `value`: K sampler calls between barrier + synchronize brackets, profiler OFF, the whole call replayed
    from one HIP graph (mcedm_amd.lib.GraphedSampler; `--no-graph` launches the ~4000 kernels eagerly instead);
  * `roofline` / `kernels`: a SEPARATE pass of `--profile-steps` eager calls with HIP event pairs around every launch
    on the launch stream (mcedm_prof_enable): per-kernel average duration against algorithmic flops / bytes;
  * `cpu_baseline`: the oracle (CPU restatement of the reference) on a bounded sample, 1 warm-up + median of 3, on all
    usable physical cores (host cores capped by the container's CPU quota) and on one thread.

Workloads (BASELINE.json `configs`):
  s128    SWE-periodic 128x128, EDM U-Net ch=128, ch_mult [1,1,1,1], attention at 16^2, 32 states / GPU  (config 3; default,
          the configuration the metric is quoted on)
  s128l3  the same with the literal ch_mult [1,1,1] (attention only in the bottleneck block, SURVEY.md 8d "both ways")
  s32     SWE-periodic 32x32, ch=64, ch_mult [1,1,1], 64 states / GPU                                    (config 2)
  ref128  the reference's own adm_edm_mcedm_res32 network (ch=64) on 128x128 fields, 32 states / GPU
  darcy128  the single-task conditional EDM (1 + 1 -> 1 channels, ch=128) on 128x128 Darcy-sized fields, 32 states / GPU (config 4)
  repaint128  RePaint-style EDM sampling of the DDPM U-Net, 18 steps x 32 resampling loops, 32 states / GPU       (config 5)
  ref_default  ref128's network with the reference's SHIPPED sampler config (50 steps, S_churn 15, n_samples 5): 99 evaluations / state
The default run also reports s32 / ref128 / s128l3 / darcy128 / ref_default / repaint128 as `secondary` entries.
"""
import argparse
import hashlib
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "s128": dict(ch=128, ch_mult=(1, 1, 1, 1), attn=(16,), H=128, W=128, batch=32,
                 name="SWE-periodic 128x128, EDM U-Net ch=128 ch_mult=[1,1,1,1] attn@16^2 (BASELINE config 3)"),
    "s128l3": dict(ch=128, ch_mult=(1, 1, 1), attn=(16,), H=128, W=128, batch=32,
                   name="SWE-periodic 128x128, EDM U-Net ch=128 ch_mult=[1,1,1] (config 3 read literally: 3 levels)"),
    "s32": dict(ch=64, ch_mult=(1, 1, 1), attn=(32,), H=32, W=32, batch=64,
                name="SWE-periodic 32x32, EDM U-Net ch=64 ch_mult=[1,1,1] (BASELINE config 2)"),
    "ref128": dict(ch=64, ch_mult=(1, 1, 1), attn=(32,), H=128, W=128, batch=32,
                   name="SWE-periodic 128x128, reference adm_edm_mcedm_res32 U-Net ch=64"),
    # the reference's SHIPPED sampler (configs/diff_sampler/edm_sampler.yaml:1-20: timesteps 50, S_churn 15, n_samples 5) on its own
    # network: stochastic Heun, 2 * 50 - 1 = 99 U-Net evaluations per state, churn noise drawn on the device (mcedm_heun_sample_rng)
    "ref_default": dict(ch=64, ch_mult=(1, 1, 1), attn=(32,), H=128, W=128, batch=30, steps=50, S_churn=15.0,
                        name="SWE-periodic 128x128, reference U-Net ch=64, shipped sampler config (50 steps, S_churn 15, "
                             "n_samples 5 x 6 inputs = 30 states), churn noise generated on the device"),
    # BASELINE config 4 read as the single-task model (configs/model/adm_edm_cond_h_res32.yaml: a -> u, 1 + 1 -> 1 channels,
    # PlCondEdm.sample_edm, no mask): 2 inputs x n_samples = 16 draws = 32 states per GPU
    "darcy128": dict(ch=128, ch_mult=(1, 1, 1, 1), attn=(16,), H=128, W=128, batch=32, single=True,
                     name="Darcy 128x128 conditional EDM (single-task U-Net ch=128, n_samples=16 x 2 inputs; BASELINE config 4)"),
}
# BASELINE config 5: RePaint-style EDM sampling of the joint DDPM (PlDdim.sample_edm, models/ddim.py:959-1051) with
# configs/model/ddim_res32.yaml (DDPM U-Net ch=64, attention at 32^2) on 128x128 fields: u known for the first 64 time
# rows, h unknown (n_time_h=0, n_time_u=64), 32 resampling loops per step -> 18*32*2 - 32 = 1120 U-Net evaluations / state
REPAINT = dict(ch=64, ch_mult=(1, 1, 1), attn=(32,), H=128, W=128, batch=32, n_repeat=32, n_time_h=0, n_time_u=64,
               name="SWE dam-break 128x128 RePaint (n_time_h=0, n_time_u=64, 32 resample loops/step), DDPM U-Net ch=64 "
                    "(BASELINE config 5)")
# Algorithmic cost of ONE U-Net forward per sample (2 x MAC of conv / linear / attention; bytes of the fused
# schedule): (GFLOP, MB).  darcy128 is the s128 network with 1-channel input / output (the difference is < 0.1 %).
ALGORITHMIC = {"s32": (1.1115, 20.3), "ref128": (18.787, 230.0), "s128l3": (70.759, 452.5), "s128": (70.843, 463.9),
               "darcy128": (70.843, 463.9), "ref_default": (18.787, 230.0)}
# kernels bound by HBM rather than by the matrix pipe (their `gbps` against the 8 TB/s roofline is the number that matters)
HBM_BOUND_KERNELS = ("gn_bwd_kernel", "gn_bwd_lds_kernel", "gn_bwd_reg_kernel", "conv_small_cout_kernel", "act_materialize_kernel", "gn_coef_kernel", "wgrad_reduce_kernel",
                     "wgrad_wino_reduce_kernel", "wgrad_thin", "adam_ema_kernel", "heun", "edm_loss_kernel", "sqnorm_kernel", "gelu", "pack_batch_kernel",
                     "pack_conv_kernel", "wino_pack_kernel")
TRAIN_LEG_TIMEOUT_S = 300      # watchdog of the multi-rank training leg (an untimed extra of the line)
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBPS = 8000.0
STEPS = 18                      # Heun steps -> 2*18 - 1 = 35 U-Net evaluations per state


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="s128", choices=sorted(WORKLOADS) + ["repaint128"])
    ap.add_argument("--batch", type=int, default=0, help="states per GPU (default: the workload's)")
    ap.add_argument("--no-graph", action="store_true", help="launch the sampler's kernels eagerly instead of one HIP graph")
    ap.add_argument("--profile-steps", type=int, default=1, help="sampler calls in the separate event-timed pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the (untimed) training-step measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other workloads")
    return ap.parse_args()


def spawn_ranks(args):
    """N > 1 typed directly: this process never initialises HIP; N fresh ranks do the work."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def synth_params(plan, seed, device):
    """Random-init weights of the architecture, keyed like DhariwalUNet.state_dict(): conv / linear weights
    U(-1,1)/sqrt(fan_in), GroupNorm gains 1 + 0.2 U, everything else 0.1 U (the reference's own init zeroes conv1 / proj /
    out_conv, adm_blocks.py:145,157,317, which would make every output identically zero)."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    P = {}
    for name, shape in zip(plan.param_names, plan.param_shapes):
        u = torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1
        if name.endswith(".weight") and len(shape) >= 2:
            v = u / math.sqrt(math.prod(shape[1:]))
        elif ".norm" in name or name.startswith("out_norm"):
            v = 1 + 0.2 * u if name.endswith(".weight") else 0.1 * u
        else:
            v = 0.1 * u
        P[name] = v.to(torch.float32).to(device)
    return P


def synth_inputs(B, H, W, seed, device):
    """Seeded N(0,1) normalised state, 'u'-task mask (h observed, u missing; datamodules/h5_dataset.py:245-247),
    cond = state*(1-mask) + N(0,1)*mask (models/mcedm.py:247), initial noise."""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    state = torch.randn(B, 2, H, W, generator=g)
    mask = torch.zeros(B, 2, H, W)
    mask[:, 1] = 1.0
    cond = state * (1 - mask) + torch.randn(B, 2, H, W, generator=g) * mask
    init = torch.randn(B, 2, H, W, generator=g)
    return cond.to(device), mask.to(device), init.to(device)


def host_cpu():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
    except OSError:
        pass
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    physical = min(len(cores), logical) if cores else logical
    # the share of the host this process may actually use (container CPU quota): more threads than that only thrash
    usable = physical
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            usable = max(1, min(physical, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                usable = max(1, min(physical, int(math.ceil(q / per))))
        except (OSError, ValueError):
            pass
    return model, physical, logical, usable


def cpu_baseline(wl, params_cpu):
    """The oracle (CPU restatement of the reference, oracle/mcedm_oracle.py) on this host's cores, on a BOUNDED sample of
    the same workload: one state through a shortened Heun pass (same network, same field size; cost per U-Net evaluation
    is what matters, and a state costs 35 of them), 1 warm-up + median of 3, scaled to states/s of the 18-step sampler.
    Run twice: all physical cores, and one thread."""
    import torch
    from oracle import mcedm_oracle as orc
    cfg = orc.UNetConfig(ch=wl["ch"], ch_mult=wl["ch_mult"], attn_resolutions=wl["attn"])
    model, physical, logical, usable = host_cpu()
    big = wl["H"] * wl["W"] * wl["ch"] >= 128 * 128 * 64
    out = {}
    for label, threads, states, nsteps in (("all_cores", usable, 2 if big else 8, 4 if big else STEPS),
                                           ("one_thread", 1, 1, 2 if big else 6)):
        torch.set_num_threads(threads)
        cond, mask, init = synth_inputs(states, wl["H"], wl["W"], 1, "cpu")
        sp = orc.SamplerParams(timesteps=nsteps)
        nfe = 2 * nsteps - 1
        times = []
        with torch.no_grad():
            for rep in range(4):
                t0 = time.perf_counter()
                orc.sample_edm(params_cpu, cfg, cond, mask, sp, init)
                times.append(time.perf_counter() - t0)
        med = statistics.median(times[1:])
        out[label] = {"states_per_s": states / (med * (2 * STEPS - 1) / nfe), "threads": threads,
                      "sample": f"{states} state(s) x {nfe} U-Net evaluations ({nsteps}-step Heun), median of 3 after 1 warm-up: "
                                f"{med:.2f} s; scaled x{(2 * STEPS - 1) / nfe:.2f} to the 35 evaluations of an 18-step state"}
    torch.set_num_threads(usable)
    a = out["all_cores"]
    return {"value": a["states_per_s"], "unit": "states/s", "cores": a["threads"], "kind": "port",
            "sample": a["sample"] + f"; torch {torch.__version__} CPU fp32", "cpu_model": model,
            "physical_cores": physical, "logical_cpus": logical, "usable_cores_cpu_quota": usable,
            "one_thread": {"value": out["one_thread"]["states_per_s"], "unit": "states/s", "cores": 1,
                           "sample": out["one_thread"]["sample"]}}


def csrc_digest():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "m-cedm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC passes (tools/rocpd_summary.py traffic writes
    profiles/r3_traffic.json with the digest of the kernel sources it was taken on); refused when the sources changed."""
    tr = None
    for name in ("r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json"):       # newest committed pass first
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", name)))
            break
        except Exception:
            continue
    if tr is None:
        return None, "no committed PMC pass"
    if tr.get("csrc_digest") != csrc_digest():
        return None, f"stale: PMC pass taken at csrc {tr.get('csrc_digest')}, sources are now {csrc_digest()}"
    for k, v in tr.get("kernels", {}).items():
        if k.replace("mcedm::", "").replace(" ", "") == kernel_name.replace(" ", ""):
            return v["traffic_bytes"], "rocprofv3 FETCH_SIZE + WRITE_SIZE (gfx950 corrections of MI355X_MICROARCH.md), avg per launch"
    return None, "kernel not in the committed PMC pass"


class Runner:
    """One workload on this rank's GPU: plan, weights, inputs, graph."""

    def __init__(self, key, B, device, rank, use_graph):
        import torch
        from mcedm_amd import lib
        self.lib, self.torch = lib, torch
        wl = WORKLOADS[key]
        self.wl, self.B, self.H, self.W = wl, B or wl["batch"], wl["H"], wl["W"]
        single = bool(wl.get("single"))
        nc = 1 if single else 2
        self.plan = lib.Plan(nc, nc, nc, wl["ch"], wl["ch_mult"], 1, wl["attn"], 128)
        self.params = synth_params(self.plan, 7, device)
        self.packed = self.plan.pack(self.params)
        self.cond, self.mask, self.init = synth_inputs(self.B, self.H, self.W, 1000 + rank, device)
        if single:      # conditioning field a (observed), state u generated everywhere: no mask (models/ddim.py:1532-1601)
            self.cond, self.mask, self.init = self.cond[:, :1].contiguous(), None, self.init[:, :1].contiguous()

        steps, churn = int(wl.get("steps", STEPS)), float(wl.get("S_churn", 0.0))

        class SP:      # configs/diff_sampler/edm_sampler.yaml; the headline workloads with S_churn = 0 (deterministic Heun), w = 0
            timesteps, sigma_min, sigma_max, rho, S_churn, S_min, S_max, S_noise, w = steps, 0.002, 80.0, 7.0, churn, 0.0, float("inf"), 1.0, 0.0
        self.sd = lib.sampler_desc(SP)
        self.nfe = 2 * steps - 1
        self.churn = churn > 0
        self.calls = 0
        self.seed = torch.zeros(1, dtype=torch.int64, device=device) if self.churn else None
        self.ws = lib.Workspace()
        self.graph = lib.GraphedSampler(self.plan, self.packed, self.sd, self.B, self.H, self.W, masked=not single,
                                        churn=self.churn, device_noise=self.churn) if use_graph else None

    def step(self):
        self.calls += 1
        if self.graph is not None:
            if self.churn:
                return self.graph(self.cond, self.mask, self.init, seed=4242 + self.calls)      # fresh noise every call
            return self.graph(self.cond, self.mask, self.init)
        return self.eager()

    def eager(self):
        if self.churn:
            self.seed.fill_(4242 + self.calls)
        return self.plan.sample(self.packed, self.sd, self.cond, self.mask, self.init, None, return_last=True, ws=self.ws,
                                rng_seed=self.seed)

    def profile(self, steps):
        """Separate event-timed pass (eager launches; HIP events cannot be recorded inside a graph)."""
        lib, torch = self.lib, self.torch
        self.eager()
        torch.cuda.synchronize()
        if steps <= 0:
            return []
        lib.prof_enable(True)
        for _ in range(steps):
            self.eager()
        torch.cuda.synchronize()
        lib.prof_enable(False)
        return lib.prof_report()

    def fwd_ms(self, reps=5):
        torch = self.torch
        x32 = self.init * 3.0
        sig = torch.tensor([1.5], device=x32.device)
        self.plan.denoise(self.packed, x32, sig, cond=self.cond, ws=self.ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            self.plan.denoise(self.packed, x32, sig, cond=self.cond, ws=self.ws)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps


class RepaintRunner:
    """BASELINE config 5 on this rank's GPU (same interface as Runner).  The per-step / per-loop noise is generated on the
    device (mcedm_repaint_sample_rng: no 576-tensor noise buffer) and the whole call -- 18 steps x 32 resampling loops =
    1120 U-Net evaluations per state, ~70 000 launches -- is replayed from ONE HIP graph (lib.GraphedRepaint)."""

    def __init__(self, B, device, rank, use_graph=True):
        import torch
        from mcedm_amd import lib
        self.lib, self.torch = lib, torch
        wl = REPAINT
        self.wl, self.B, self.H, self.W = wl, B or wl["batch"], wl["H"], wl["W"]
        self.plan = lib.DdpmPlan(2, 2, wl["ch"], wl["ch_mult"], 1, wl["attn"], wl["H"])
        self.params = synth_params(self.plan, 7, device)
        half = wl["ch"] // 2
        freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))   # ddim_blocks.py:22-24
        self.packed = self.plan.pack(self.params, freqs.to(device))
        g = torch.Generator(device="cpu").manual_seed(2000 + rank)
        self.hu = torch.randn(self.B, 2, self.H, self.W, generator=g).to(device)
        self.init = torch.randn(self.B, 2, self.H, self.W, generator=g).to(device)
        betas = torch.linspace(1e-4, 0.02, 1000, dtype=torch.float64).float()                             # ddim_res32.yaml diffusion
        ab = (1.0 - betas).cumprod(dim=0)
        steps = ((1 - ab) / ab).sqrt().flip(dims=(0,))
        aext = (1 - torch.cat([torch.zeros(1), betas])).cumprod(dim=0)

        class SP:      # configs/diff_sampler/edm_sampler_inv.yaml with the benchmark's 18 deterministic steps
            timesteps, sigma_min, sigma_max, rho, S_churn, S_min, S_max, S_noise, w = STEPS, 0.002, 80.0, 7.0, 0.0, 0.0, float("inf"), 1.0, 0.0
            n_repeat, n_time_h, n_time_u = wl["n_repeat"], wl["n_time_h"], wl["n_time_u"]
        self.rd, self._keep = lib.repaint_desc(SP, steps, aext, 1, 1)
        self.seed = torch.tensor([4242 + rank], dtype=torch.int64, device=device)
        self.ws = lib.Workspace()
        self.nfe = STEPS * wl["n_repeat"] * 2 - wl["n_repeat"]
        self.graph, self.capture_s, self.calls = None, None, 0
        if use_graph:
            t0 = time.perf_counter()
            self.graph = lib.GraphedRepaint(self.plan, self.packed, self.rd, self._keep, self.B, True, ws=self.ws)
            torch.cuda.synchronize()
            self.capture_s = time.perf_counter() - t0          # one warm-up call + capture + instantiation

    def step(self):
        self.calls += 1
        if self.graph is not None:
            return self.graph(self.hu, self.init, 4242 + self.calls)       # fresh noise every call
        return self.eager()

    def eager(self):
        return self.plan.repaint_sample(self.packed, self.rd, self.hu, self.init, return_last=True, ws=self.ws, rng_seed=self.seed)

    profile = Runner.profile

    def fwd_ms(self, reps=5):
        torch = self.torch
        self.plan.denoise(self.packed, self.init, 1.5, 500.0, ws=self.ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            self.plan.denoise(self.packed, self.init, 1.5, 500.0, ws=self.ws)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps


def kernel_table(prof):
    rows = sorted(({"name": r["name"], "launches": r["launches"], "total_ms": round(r["total_ms"], 3),
                    "tflops": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 2),
                    "gbps": round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1)} for r in prof if r["total_ms"] > 0),
                  key=lambda r: -r["total_ms"])
    return rows


def hbm_bound_rows(prof):
    """The HBM-bound kernels of a profile with their achieved fraction of the 8 TB/s roofline on ALGORITHMIC bytes."""
    rows = [r for r in kernel_table(prof) if any(k in r["name"] for k in HBM_BOUND_KERNELS)]
    return [dict(r, hbm_frac=round(r["gbps"] / PEAK_HBM_GBPS, 3)) for r in rows]


def fractions(key, states_per_s, fwd_ms, B, nfe=None):
    """Both roofline fractions SURVEY.md 8(d) asks for next to every number, on the algorithmic cost of the workload:
    per U-Net forward (batch B in fwd_ms) and per denoised state (35 forwards) at the measured whole-job rate."""
    if key not in ALGORITHMIC:
        return {}
    gf, mb = ALGORITHMIC[key]
    nfe = nfe or 2 * STEPS - 1
    return {"forward_fp32_frac_algorithmic": gf * 1e9 * B / (fwd_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "forward_hbm_frac": mb * 1e6 * B / (fwd_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
            "per_state": {"gflop": gf * nfe, "gbyte": mb * nfe / 1e3,
                          "fp32_frac": states_per_s * gf * nfe * 1e9 / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                          "hbm_frac": states_per_s * mb * nfe * 1e6 / 1e9 / PEAK_HBM_GBPS,
                          "note": "fp32 MFMA binds (ridge 19.7 flop/B, the network's AI is 55-156): at the fp32 peak the HBM "
                                  "fraction cannot exceed 0.13 (ch=128) / 0.24 (ch=64, 128^2) / 0.36 (32^2)"}}


def roofline_of(prof):
    if not prof:
        return None
    total_ms = sum(r["total_ms"] for r in prof) or 1.0
    dom = max(prof, key=lambda r: r["total_ms"])
    avg_ms = dom["total_ms"] / dom["launches"]
    achieved = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12
    traffic, note = committed_traffic(dom["name"])
    extra = {}
    if dom["name"].startswith("conv_wino"):
        # Winograd F(2x2, 3x3): the profiler records the convolution's ALGORITHMIC flops (2 x MAC of the
        # direct form).  `achieved` / `frac` are the flops the kernel actually ISSUES as MFMAs (4/9 of them) over
        # the launch time against the matrix peak (<= 1: how busy the matrix pipe is); the direct-equivalent rate -- what a
        # direct convolution would have to sustain to match the launch time -- is reported separately and may exceed the peak
        extra = {"algorithm": "Winograd F(2x2,3x3): 16 multiplies per 2x2 outputs and (cin, cout) instead of 36",
                 "direct_equivalent_tflops": achieved, "direct_equivalent_frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                 "executed_over_algorithmic": 4.0 / 9.0}
        achieved *= 4.0 / 9.0
    executed = 4.0 / 9.0 if extra else 1.0
    return {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP32_MFMA_TFLOPS, **extra, "traffic": traffic, "traffic_note": note,
            "kernel": dom["name"], "launches": dom["launches"], "avg_launch_ms": avg_ms,
            "flops_per_launch": dom["flops"] / dom["launches"] * executed, "algorithmic_flops_per_launch": dom["flops"] / dom["launches"],
            "bytes_per_launch": dom["bytes"] / dom["launches"],
            "share_of_kernel_time": dom["total_ms"] / total_ms,
            "hbm_frac_on_algorithmic_bytes": dom["bytes"] / dom["launches"] / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
            "measured": "HIP event pairs on the launch stream, separate eager pass after the timed region"}


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MCEDM_BENCH_ONE_CARD=1: rehearsal of the N > 1 control flow on a one-GPU box -- every rank on cuda:0, process group over
    # gloo (RCCL refuses two ranks per device).  Not a measurement: the ranks share the card.
    one_card = os.environ.get("MCEDM_BENCH_ONE_CARD") == "1"
    if one_card:
        local_rank = 0

    import torch
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_card:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)   # RCCL over xGMI; used only for the barrier / max-reduce

    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib

    repaint = args.workload == "repaint128"
    run = RepaintRunner(args.batch, device, rank, not args.no_graph) if repaint else Runner(args.workload, args.batch, device, rank, not args.no_graph)
    wl, B, H, W = run.wl, run.B, run.H, run.W
    if repaint or wl.get("single"):
        args.no_train = True          # the training and CPU legs are defined on the joint-model workloads
        args.no_cpu_baseline = True

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run.step()
    barrier()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    prof = run.profile(args.profile_steps) if rank == 0 else []
    fwd_ms = run.fwd_ms()

    def headline_line(train_ms_):
        states = B * world * args.steps
        return {
            "metric": "denoised_states_per_sec_18step_edm_heun", "value": states / elapsed, "unit": "states/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "states_per_gpu": B, "global_batch": B * world, "H": H, "W": W,
                       "sampler": (f"EDM Heun + RePaint, 18 steps x {wl['n_repeat']} resampling loops, {run.nfe} NFE/state, S_churn=0, "
                                   "re-noising draws generated on the device (Philox), fp64 state / fp32 net") if repaint else
                                  "EDM Heun, 18 steps, 35 NFE/state, S_churn=0, w=0, fp64 state / fp32 net",
                       "parallelism": f"batch-sharded x{world}, no data-path collective",
                       "launch": "eager" if args.no_graph else "one HIP graph per sampler call"},
            "timed_with_profiler": False, "unet_fwd_ms": fwd_ms, "unet_fwd_batch": B, "train_step_ms": train_ms_,
            "train_samples_per_sec": (B * world / (train_ms_ * 1e-3)) if train_ms_ else None,
            "roofline": roofline_of(prof), "kernels": kernel_table(prof)[:8], "hbm_bound_kernels": hbm_bound_rows(prof),
            "csrc_digest": csrc_digest(),
            **(fractions(args.workload, states / elapsed / world, fwd_ms, B) if not repaint else {}),
        }

    # one data-parallel training step (models/mcedm.py:254-281 + clip/Adam/EMA), outside the timed region:
    # noise -> denoise(training) -> loss -> backward -> gradient all-reduce -> fused clip+Adam+EMA
    train_ms, train_prof, train_error = None, None, None

    train_multi, phase = None, {"name": "not started"}

    def train_leg():
        from mcedm_amd.train import FlatTrainState
        gen = torch.Generator(device="cpu").manual_seed(7 + rank)
        xs = torch.randn(B, 2, H, W, generator=gen).to(device)
        nz = torch.randn(B, 2, H, W, generator=gen).to(device)
        rn = torch.randn(B, generator=gen).to(device)

        def timed(ts, what, nt=3, **kw):
            """1 warm-up + nt steps between barrier + synchronize brackets; MAX over ranks."""
            phase["name"] = what
            ts.step(xs, run.cond, run.mask, nz, rn, **kw)
            barrier()
            t1 = time.perf_counter()
            for _ in range(nt):
                loss = ts.step(xs, run.cond, run.mask, nz, rn, **kw)
            barrier()
            ms_ = (time.perf_counter() - t1) / nt * 1e3
            assert torch.isfinite(loss).all()
            if world > 1:
                t = torch.tensor([ms_], dtype=torch.float64, device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms_ = float(t)
            return ms_

        ts = FlatTrainState(run.plan, run.params, packed=run.packed)          # up to 4 buckets (the default of the trainer)
        nb = len(ts.sync.ranges)
        ms = timed(ts, f"step, {nb} bucket(s)")
        tp, multi = None, None
        if world == 1:            # one more step, launched eagerly with HIP event pairs around every kernel (single rank
                                  # only: a step contains the gradient all-reduce, which every rank must enter)
            ts.use_graph = False
            ts.step(xs, run.cond, run.mask, nz, rn)
            torch.cuda.synchronize()
            lib.prof_enable(True)
            ts.step(xs, run.cond, run.mask, nz, rn)
            torch.cuda.synchronize()
            lib.prof_enable(False)
            tp = lib.prof_report()
        else:
            # What the first multi-GPU record needs to separate compute, exchange and overlap: the same
            # step WITHOUT the gradient exchange (backward + clip + Adam + EMA only), and with ONE bucket (the whole flat
            # gradient in one all-reduce behind the backward: no overlap) next to the bucketed, overlapped step above.
            ms_noex = timed(ts, "step without the exchange", exchange=False)
            nbytes = ts.flat_g.numel() * ts.flat_g.element_size()
            bucket_bytes = [(hi - lo) * 4 for lo, hi in ts.sync.ranges]
            del ts
            ts1 = FlatTrainState(run.plan, run.params, packed=run.packed, max_buckets=1)
            ms1 = timed(ts1, "step, 1 bucket")
            ts = ts1
            multi = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "allreduce_bytes_per_step": nbytes,
                     "bucket_bytes": bucket_bytes, "step_ms": {f"buckets_{nb}": ms, "buckets_1": ms1},
                     "step_ms_no_exchange": ms_noex,
                     "exchange_exposed_ms": {f"buckets_{nb}": ms - ms_noex, "buckets_1": ms1 - ms_noex},
                     "note": "MAX over ranks, 1 warm-up + 3 steps each; exposed = step - step without the all-reduce; "
                             "buckets_1 = no overlap with the backward (one all-reduce of the whole flat gradient after it)"}
        del ts
        phase["name"] = "done"
        return ms, tp, multi

    if not args.no_train:
        if world == 1:
            train_ms, train_prof, train_multi = train_leg()
        else:
            # The multi-rank step (bucketed RCCL all-reduce on a side stream under the backward) is an untimed extra of this
            # line: neither an exception nor a stuck collective in it may take the headline measurement with it.  A watchdog
            # ends every rank after TRAIN_LEG_TIMEOUT_S; rank 0 first prints the line it has (train fields null, train_error set).
            import threading
            emergency = {"line": None}

            def bail():
                # the headline line still goes out (the sampler measurement stands), but the process ends NON-ZERO: a stuck
                # collective or backward must not read as success to torchrun / the driver
                if rank == 0 and emergency["line"] is not None:
                    print(json.dumps(dict(emergency["line"], status="train_leg_timeout",
                                          train_error=f"multi-rank training leg exceeded {TRAIN_LEG_TIMEOUT_S} s in phase "
                                                      f"'{phase['name']}' (rank 0's view)")), flush=True)
                sys.stderr.write(f"[bench rank {rank}] training-leg watchdog fired in phase '{phase['name']}'\n")
                sys.stderr.flush()
                os._exit(3)

            if rank == 0:
                emergency["line"] = headline_line(None)
            dog = threading.Timer(TRAIN_LEG_TIMEOUT_S, bail)
            dog.daemon = True
            dog.start()
            try:
                train_ms, train_prof, train_multi = train_leg()
            except Exception as e:      # noqa: BLE001  (reported in the line; the sampler measurement stands)
                train_error = f"{type(e).__name__} in phase '{phase['name']}': {str(e)[:300]}"
            dog.cancel()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    line = headline_line(train_ms)
    line["status"] = "ok"
    if train_error:
        line["train_error"] = train_error
        line["status"] = "train_leg_failed"
    if train_multi:
        line["train_multi_gpu"] = train_multi
    if train_ms and train_prof:
        tflops = sum(r["flops"] for r in train_prof)           # forward + data-gradient + weight-gradient GEMM flops
        dom = sorted(train_prof, key=lambda r: -r["total_ms"])[:3]
        line["train_roofline"] = {
            "bound": "mfma", "unit": "TFLOP/s", "peak": PEAK_FP32_MFMA_TFLOPS,
            "achieved": tflops / (train_ms * 1e-3) / 1e12, "frac": tflops / (train_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "flops_per_step": tflops, "step_ms": train_ms, "kernel_ms_sum": sum(r["total_ms"] for r in train_prof),
            "note": "whole optimisation step (noise, forward, loss, backward, clip, Adam, EMA) against the fp32 MFMA peak; "
                    "step timed over 3 graph-replayed steps, per-kernel rows from one extra eager step with event pairs",
            "kernels": [{"name": r["name"], "launches": r["launches"], "total_ms": round(r["total_ms"], 3),
                         "tflops": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 1),
                         "frac": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 3)} for r in dom],
            "hbm_bound_kernels": hbm_bound_rows(train_prof)}
    if prof and args.profile_steps > 0:
        flops_state = sum(r["flops"] for r in prof) / (args.profile_steps * B)
        line["forward_fp32_frac"] = flops_state * B * world * args.steps / elapsed / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world)

    # the other workloads, briefly (driver-timed numbers for BASELINE config 2 and the reference's own network)
    line["secondary"] = {}
    if not args.no_secondary and world == 1 and args.workload == "s128" and not args.batch:
        del run.graph
        torch.cuda.empty_cache()
        for key in ("s32", "ref128", "s128l3", "darcy128", "ref_default"):
            r2 = Runner(key, 0, device, 0, not args.no_graph)
            r2.step()
            torch.cuda.synchronize()
            n2 = 5 if key == "s32" else 2
            t2 = time.perf_counter()
            for _ in range(n2):
                r2.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t2) / n2
            p2 = r2.profile(1)
            ro = roofline_of(p2)
            line["secondary"][key] = {"workload": r2.wl["name"], "states_per_gpu": r2.B, "value": r2.B / dt, "unit": "states/s",
                                      "ms_per_step": dt * 1e3, "steps": n2, "nfe_per_state": r2.nfe, "unet_fwd_ms": r2.fwd_ms(),
                                      "fp32_frac_whole_sampler": sum(r["flops"] for r in p2) / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                      "dominant_kernel": ro["kernel"], "dominant_kernel_tflops": ro["achieved"],
                                      "dominant_kernel_frac": ro["frac"], "dominant_kernel_share": ro["share_of_kernel_time"]}
            line["secondary"][key].update(fractions(key, r2.B / dt, line["secondary"][key]["unet_fwd_ms"], r2.B, r2.nfe))
            if key in ("s32", "ref128") and not args.no_train:
                # config 2 is "train + sample": one optimisation step of this network (noise, forward, loss, backward, clip, Adam,
                # EMA; the fused trainer) at the workload's batch, 1 warm-up + 3 steps
                try:
                    from mcedm_amd.train import FlatTrainState
                    gen = torch.Generator(device="cpu").manual_seed(11)
                    xs2 = torch.randn(r2.B, 2, r2.H, r2.W, generator=gen).to(device)
                    nz2 = torch.randn(r2.B, 2, r2.H, r2.W, generator=gen).to(device)
                    rn2 = torch.randn(r2.B, generator=gen).to(device)
                    ts2 = FlatTrainState(r2.plan, r2.params, packed=r2.packed)
                    ts2.step(xs2, r2.cond, r2.mask, nz2, rn2)
                    torch.cuda.synchronize()
                    t3 = time.perf_counter()
                    for _ in range(3):
                        l2 = ts2.step(xs2, r2.cond, r2.mask, nz2, rn2)
                    torch.cuda.synchronize()
                    tms = (time.perf_counter() - t3) / 3 * 1e3
                    assert torch.isfinite(l2).all()
                    gf = ALGORITHMIC[key][0]
                    line["secondary"][key].update({"train_step_ms": tms, "train_samples_per_sec": r2.B / (tms * 1e-3),
                                                   "train_fp32_frac": 3 * gf * 1e9 * r2.B / (tms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS})
                    del ts2
                except Exception as e:                                   # an extra of the line: never fatal
                    line["secondary"][key]["train_error"] = f"{type(e).__name__}: {str(e)[:200]}"
            del r2
            torch.cuda.empty_cache()

        # BASELINE config 5 (RePaint on the DDPM U-Net): 32 states per call = 35 840 U-Net evaluations, graph-replayed,
        # 1 warm-up + 2 timed calls, then one eager call with event pairs for its own roofline
        r5 = RepaintRunner(0, device, 0, not args.no_graph)
        o5 = r5.step()
        torch.cuda.synchronize()
        n5 = 2
        t5 = time.perf_counter()
        for _ in range(n5):
            o5 = r5.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t5) / n5
        assert torch.isfinite(o5).all()
        p5 = r5.profile(1)
        ro = roofline_of(p5)
        line["secondary"]["repaint128"] = {"workload": r5.wl["name"], "states_per_gpu": r5.B, "value": r5.B / dt, "unit": "states/s",
                                           "ms_per_step": dt * 1e3, "steps": n5, "warmup": 1, "nfe_per_state": r5.nfe,
                                           "unet_evals_per_s": r5.B * r5.nfe / dt, "unet_fwd_ms": r5.fwd_ms(),
                                           "launch": "eager" if args.no_graph else "one HIP graph per call",
                                           "graph_capture_s": r5.capture_s,
                                           "fp32_frac_whole_sampler": sum(r["flops"] for r in p5) / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                                           "gflop_per_unet_eval_per_state": sum(r["flops"] for r in p5) / (r5.B * r5.nfe) / 1e9,
                                           "dominant_kernel": ro["kernel"], "dominant_kernel_tflops": ro["achieved"],
                                           "dominant_kernel_frac": ro["frac"], "dominant_kernel_share": ro["share_of_kernel_time"],
                                           "kernels": kernel_table(p5)[:6]}
        del r5
        torch.cuda.empty_cache()

    line["cpu_baseline"] = None          # timed on rank 0 of the 1-GPU run only
    if repaint:
        args.no_cpu_baseline = True      # the CPU leg is defined on the headline workloads
    if not args.no_cpu_baseline and world == 1:
        line["cpu_baseline"] = cpu_baseline(wl, {k: v.cpu() for k, v in run.params.items()})
        line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
