#!/usr/bin/env python
"""bench.py -- denoised states/sec of the 18-step EDM Heun sampler (35 U-Net evaluations per state).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload s128|s32|ref128] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W          # N > 1: one rank per GPU

A "step" is one pass of the hot path over one batch: ``sample_edm`` (models/mcedm.py:570-638) on B states per
GPU, inputs already resident in HBM.  The batch axis is sharded across ranks with no data-path collective
(SURVEY.md section 8e), so scaling is weak: per-GPU batch is fixed.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json `configs`):
  s128    SWE-periodic 128x128, EDM U-Net ch=128, ch_mult [1,1,1,1], attention at 16^2, 32 states / GPU  (config 3; default,
          the configuration the metric is quoted on)
  s32     SWE-periodic 32x32, ch=64, ch_mult [1,1,1], 64 states / GPU                                    (config 2)
  ref128  the reference's own adm_edm_mcedm_res32 network (ch=64) on 128x128 fields, 32 states / GPU
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "s128": dict(ch=128, ch_mult=(1, 1, 1, 1), attn=(16,), H=128, W=128, batch=32,
                 name="SWE-periodic 128x128, EDM U-Net ch=128 ch_mult=[1,1,1,1] attn@16^2 (BASELINE config 3)"),
    "s32": dict(ch=64, ch_mult=(1, 1, 1), attn=(32,), H=32, W=32, batch=64,
                name="SWE-periodic 32x32, EDM U-Net ch=64 ch_mult=[1,1,1] (BASELINE config 2)"),
    "ref128": dict(ch=64, ch_mult=(1, 1, 1), attn=(32,), H=128, W=128, batch=32,
                   name="SWE-periodic 128x128, reference adm_edm_mcedm_res32 U-Net ch=64"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBPS = 8000.0


def synth_inputs(B, H, W, seed, device):
    """Seeded N(0,1) normalised state, 'u'-task mask (h observed, u missing; datamodules/h5_dataset.py:245-247),
    cond = state*(1-mask) + N(0,1)*mask (models/mcedm.py:247), initial noise."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    state = torch.randn(B, 2, H, W, generator=g)
    mask = torch.zeros(B, 2, H, W)
    mask[:, 1] = 1.0
    cond = state * (1 - mask) + torch.randn(B, 2, H, W, generator=g) * mask
    init = torch.randn(B, 2, H, W, generator=g)
    return cond.to(device), mask.to(device), init.to(device)


def cpu_baseline(cfg, wl, sample_states, steps=18):
    """The oracle (CPU restatement of the reference, oracle/mcedm_oracle.py) timed on this host's cores on a
    bounded sample of the same workload."""
    from oracle import mcedm_oracle as orc
    threads = torch.get_num_threads()
    P = orc.make_params(cfg, 7)
    cond, mask, init = synth_inputs(sample_states, wl["H"], wl["W"], 1, "cpu")
    sp = orc.SamplerParams(timesteps=steps)
    t0 = time.perf_counter()
    with torch.no_grad():
        orc.sample_edm(P, cfg, cond, mask, sp, init)
    dt = time.perf_counter() - t0
    return {"value": sample_states / dt, "unit": "states/s", "cores": threads, "kind": "port",
            "sample": f"{sample_states} state(s) of the same workload, one 18-step Heun pass (35 NFE), {dt:.1f} s, "
                      f"torch {torch.__version__} CPU fp32, {threads} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="s128", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="states per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the (untimed) training-step measurement")
    ap.add_argument("--cpu-states", type=int, default=0, help="states in the CPU baseline sample (default: auto)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs one rank per GPU: launch with python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)   # RCCL over xGMI; used only for the barrier / max-reduce

    import mcedm_amd  # noqa: F401
    from mcedm_amd import lib
    from oracle import mcedm_oracle as orc       # only for the architecture record + deterministic weights + cpu_baseline

    wl = WORKLOADS[args.workload]
    B = args.batch or wl["batch"]
    H, W = wl["H"], wl["W"]
    cfg = orc.UNetConfig(ch=wl["ch"], ch_mult=wl["ch_mult"], attn_resolutions=wl["attn"])
    plan = lib.Plan(cfg.in_channels, cfg.cond_channels, cfg.out_ch, cfg.ch, cfg.ch_mult, cfg.num_res_blocks,
                    cfg.attn_resolutions, cfg.resolution)
    params = {k: v.to(device) for k, v in orc.make_params(cfg, 7).items()}     # random-init weights of that architecture
    packed = plan.pack(params)
    cond, mask, init = synth_inputs(B, H, W, 1000 + rank, device)
    sd = lib.sampler_desc(orc.SamplerParams(timesteps=18))                     # S_churn=0, w=0: deterministic Heun
    ws = lib.Workspace()

    def step():
        return plan.sample(packed, sd, cond, mask, init, None, return_last=True, ws=ws)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    lib.prof_enable(rank == 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    elapsed = time.perf_counter() - t0
    lib.prof_enable(False)
    prof = lib.prof_report() if rank == 0 else []
    assert torch.isfinite(out).all()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    # U-Net forward latency (one model_precond call, models/mcedm.py:199-211), outside the timed region
    x32 = init * 3.0
    sig = torch.tensor([1.5], device=device)
    plan.denoise(packed, x32, sig, cond=cond, ws=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    nf = 5
    for _ in range(nf):
        plan.denoise(packed, x32, sig, cond=cond, ws=ws)
    e1.record()
    torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / nf

    # one data-parallel training step (models/mcedm.py:254-281 + clip/Adam/EMA), outside the timed region:
    # noise -> denoise(training) -> loss -> backward -> gradient all-reduce -> fused clip+Adam+EMA
    train_ms = None
    if not args.no_train:
        from mcedm_amd.train import allreduce_mean_, views_like
        names = plan.param_names
        flat_p = torch.cat([params[n].reshape(-1) for n in names])
        pviews = dict(zip(names, views_like(flat_p, [params[n] for n in names])))
        flat_g, flat_m, flat_v, flat_e = (torch.zeros_like(flat_p) for _ in range(4))
        flat_e.copy_(flat_p)
        gviews = views_like(flat_g, [params[n] for n in names])
        sq = torch.zeros(1, dtype=torch.float64, device=device)
        gen = torch.Generator(device="cpu").manual_seed(7 + rank)
        xs = torch.randn(B, 2, H, W, generator=gen).to(device)
        nz = torch.randn(B, 2, H, W, generator=gen).to(device)
        rn = torch.randn(B, generator=gen).to(device)
        tws = lib.Workspace()

        def train_step(k):
            pk = plan.pack(pviews, packed)
            x_noise, sigma = lib.edm_noise_inputs(xs, mask, nz, rn)
            D = plan.denoise(pk, x_noise, sigma, cond=cond, ws=tws, training=True)
            loss, dD = lib.edm_loss(D, xs, mask, sigma)
            plan.denoise_backward(pk, pviews, x_noise, sigma, cond, dD, gviews, tws)
            allreduce_mean_(flat_g, average=False)
            lib.sqnorm(flat_g, sq)
            lib.adam_ema_step(flat_p, flat_g, flat_m, flat_v, flat_e, k, sqnorm_t=sq, grad_scale=1.0 / world)
            return loss

        train_step(1)
        barrier()
        t1 = time.perf_counter()
        nt = 3
        for k in range(nt):
            loss = train_step(2 + k)
        barrier()
        train_ms = (time.perf_counter() - t1) / nt * 1e3
        assert torch.isfinite(loss).all()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    states = B * world * args.steps
    value = states / elapsed
    # dominant kernel of the timed region, timed live with HIP event pairs on the launch stream
    total_ms = sum(r["total_ms"] for r in prof) or 1.0
    dom = max(prof, key=lambda r: r["total_ms"])
    avg_ms = dom["total_ms"] / dom["launches"]
    achieved = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12
    roofline = {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                "kernel": dom["name"], "launches": dom["launches"], "avg_launch_ms": avg_ms,
                "flops_per_launch": dom["flops"] / dom["launches"], "bytes_per_launch": dom["bytes"] / dom["launches"],
                "share_of_kernel_time": dom["total_ms"] / total_ms,
                "hbm_frac_on_algorithmic_bytes": dom["bytes"] / dom["launches"] / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS}
    # HBM traffic of that kernel from the committed rocprofv3 PMC passes (tools/rocpd_summary.py traffic), per launch
    try:
        tr = json.load(open(os.path.join(ROOT, "profiles", "r1_traffic.json")))
        for k, v in tr.items():
            norm = k.replace("mcedm::", "").replace(" ", "")
            if norm == dom["name"].replace(" ", ""):
                roofline["traffic"] = v["traffic_bytes"]
                roofline["traffic_note"] = "rocprofv3 FETCH_SIZE (x2 gfx950 bracket upper end) + WRITE_SIZE, avg per launch"
    except Exception:
        pass
    kernels = sorted(({"name": r["name"], "launches": r["launches"], "total_ms": round(r["total_ms"], 3),
                       "tflops": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 2),
                       "gbps": round(r["bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1)} for r in prof),
                     key=lambda r: -r["total_ms"])
    line = {
        "metric": "denoised_states_per_sec_18step_edm_heun", "value": value, "unit": "states/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["name"], "states_per_gpu": B, "global_batch": B * world, "H": H, "W": W,
                   "sampler": "EDM Heun, 18 steps, 35 NFE/state, S_churn=0, w=0, fp64 state / fp32 net",
                   "parallelism": f"batch-sharded x{world}, no data-path collective"},
        "unet_fwd_ms": fwd_ms, "unet_fwd_batch": B, "train_step_ms": train_ms,
        "train_samples_per_sec": (B * world / (train_ms * 1e-3)) if train_ms else None,
        "roofline": roofline, "kernels": kernels[:8],
    }
    line["cpu_baseline"] = None          # timed on rank 0 of the 1-GPU run only
    if not args.no_cpu_baseline and world == 1:
        n_cpu = args.cpu_states or (2 if H * W * wl["ch"] >= 128 * 128 * 64 else 8)
        line["cpu_baseline"] = cpu_baseline(cfg, wl, n_cpu)
        line["gpu_over_cpu"] = value / world / line["cpu_baseline"]["value"]
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
