"""Golden vectors for the evaluation loops and the classifier-free sampler, made by RUNNING
THE REFERENCE (build container only; needs /root/reference):

  * PlMcedm.test_step        models/mcedm.py:343-441   n_samples in {2, 16}, system in {swe_per, darcy}, down_factor 1 / 2
  * PlMcedm.validation_step  models/mcedm.py:283-341   evaluated epoch (current_epoch = 0)
  * PlMcedm.sample_edm with w = 0.5 (classifier-free branch of get_denoised, :453-458) through all 35 evaluations

Every random draw of the reference is injected (torch.randn_like is replaced by a queue) and every ``self.log`` call is
recorded.  The oracle (oracle/mcedm_oracle.py eval_test_step / eval_validation_step / sample_edm) is cross-checked on
every case before anything is written.  Also prints the CPU noise floor of the 18-step sampler (1 thread vs 8 threads
of the same oracle), which the sampler tolerances of tests/ are derived from.

    python oracle/make_golden_steps.py        # rewrites tests/golden/steps.npz
"""
import os
import sys

import make_golden as mg            # sets up the reference import, the Lightning stand-in and helpers

import numpy as np
import torch

from oracle import fixtures as fx
from oracle import mcedm_oracle as orc

CFG_P = fx.CFG_P


class _Datamodule:
    def __init__(self, down_factor, down_interp):
        self.down_factor, self.down_interp = down_factor, down_interp


class _Trainer:
    def __init__(self, dm):
        self.datamodule = dm


def reference_module(sampler, system, logs):
    pl_mod = mg.build_reference(CFG_P, seed=7, sampler=sampler)
    st = fx.STEP_NORM_STATS
    pl_mod.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    pl_mod.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    pl_mod.set_pde_loss_function(system, False)
    pl_mod.log = lambda name, value, **k: logs.__setitem__(name, torch.as_tensor(value).detach().clone())
    pl_mod.current_epoch = 0
    return pl_mod


def golden_steps():
    out = {}
    P = orc.make_params(CFG_P, 7)
    st = fx.STEP_NORM_STATS
    for tag, c in fx.STEP_CASES.items():
        n = c["n_samples"]
        sp = mg.sampler_dict(n_samples=n)
        logs = {}
        pl_mod = reference_module(sp, c["system"], logs)
        pl_mod.set_test_sampler_params(mg._wrap(sp))
        pl_mod.trainer = _Trainer(_Datamodule(c["down_factor"], c["down_interp"]))
        h, u, masks, noises = fx.step_inputs(tag)
        nB = n * fx.STEP_B
        queue = []
        for name in masks:          # per task: get_cond_in, the unused `noise`, sample_edm's hu_noise, 18 per-step draws
            queue += [noises[name][0], torch.zeros(nB, 2, fx.STEP_T, fx.STEP_X), noises[name][1]]
            queue += [torch.zeros(nB, 2, fx.STEP_T, fx.STEP_X, dtype=torch.float64)] * 18
        with torch.no_grad(), mg._Inject(queue) as inj:
            res = pl_mod.test_step((h, None, None, u, masks), 0)
        assert not inj.like_queue, "the reference drew fewer tensors than injected"
        o = orc.eval_test_step(P, CFG_P, h, u, masks, noises, st, orc.SamplerParams(), n, c["system"],
                               c["down_factor"] if c["down_interp"] else 1)
        keys = sorted(res)
        assert keys == sorted(k for k in o if not k.startswith("log::")), (keys, sorted(o))
        if n >= 15:
            assert not any(k.startswith("traj_") or k.startswith("gt_") for k in res)
        for k in keys:
            mg.check(f"test_step[{tag}] {k}", o[k], res[k], rtol=1e-4, atol=1e-5)
            out[f"{tag}::{k}"] = res[k]
        for k, v in logs.items():
            if k.startswith("test_mae_"):
                name = k[len("test_mae_"):]
                mg.check(f"test_step[{tag}] log {k}", o[f"loss_{name}"], v, rtol=1e-4, atol=1e-6)
            else:
                mg.check(f"test_step[{tag}] log {k}", o[f"log::{k}"], v, rtol=2e-3, atol=1e-6)
            out[f"{tag}::log::{k}"] = v
        if n < 15:
            assert tuple(res["traj_u"].shape) == (fx.STEP_B, 1, fx.STEP_T, fx.STEP_X, n, 2) and res["traj_u"].dtype == torch.float64

    # validation_step on an evaluated epoch; and the early return on the others (mcedm.py:284-285)
    tag = "swe_n2"
    sp = mg.sampler_dict()
    logs = {}
    pl_mod = reference_module(sp, "swe_per", logs)
    h, u, masks, noises = fx.step_inputs(tag)
    vnoise = {k: (noises[k][0], noises[k][1][:fx.STEP_B]) for k in masks}
    queue = [torch.zeros(fx.STEP_B, 2, fx.STEP_T, fx.STEP_X)]                   # `noise`, drawn once, shape only
    for name in masks:
        queue += [vnoise[name][0], vnoise[name][1]] + [torch.zeros(fx.STEP_B, 2, fx.STEP_T, fx.STEP_X, dtype=torch.float64)] * 18
    with torch.no_grad(), mg._Inject(queue) as inj:
        res = pl_mod.validation_step((h, None, None, u, masks), 0)
    assert not inj.like_queue
    o = orc.eval_validation_step(P, CFG_P, h, u, masks, vnoise, st, orc.SamplerParams(), "swe_per")
    assert res.pop("epoch") == 0
    for k in sorted(res):
        mg.check(f"validation_step {k}", o[k], res[k], rtol=1e-4, atol=1e-5)
        out[f"val::{k}"] = res[k]
    for k, v in logs.items():
        if k.startswith("val_pde"):
            mg.check(f"validation_step log {k}", o[f"log::{k}"], v, rtol=2e-3, atol=1e-6)
        out[f"val::log::{k}"] = v
    pl_mod.current_epoch = 7
    assert pl_mod.validation_step((h, None, None, u, masks), 0) == {"epoch": 7}

    # classifier-free sampler: w = 0.5 through the whole Heun loop (two U-Net evaluations per denoiser call)
    w = fx.CFG_SAMPLER_W
    sp = mg.sampler_dict(w=w)
    pl_mod = mg.build_reference(CFG_P, seed=7, sampler=sp)
    cond, m, init, steps = fx.sampler_inputs("det_u")
    with torch.no_grad(), mg._Inject([init] + steps):
        xs = pl_mod.sample_edm(torch.zeros(4, 2, 32, 32), cond, m, mg._wrap(sp), return_last=False)
    xo = orc.sample_edm(P, CFG_P, cond, m, orc.SamplerParams(w=w), init, steps, return_last=False)
    mg.check("sample_edm cfg w=0.5", xo, xs, rtol=1e-3, atol=1e-4)
    out["cfg_u_xs_last"] = xs[:, -1:].contiguous()
    out["cfg_u_xs_traj"] = xs[:, ::6].contiguous()
    mg.save("steps.npz", seed=7, **out)


def noise_floor():
    """CPU-vs-CPU deviation of the oracle's 18-step sampler: 1 thread vs 8 threads, same code."""
    P = orc.make_params(CFG_P, 7)
    for tag in ("det_u", "det_h"):
        cond, m, init, _ = fx.sampler_inputs(tag)
        res = []
        for th in (1, 8):
            torch.set_num_threads(th)
            with torch.no_grad():
                res.append(orc.sample_edm(P, CFG_P, cond, m, orc.SamplerParams(), init))
        d = (res[0] - res[1]).abs()
        rel = (d / (res[1].abs() + 1e-5)).max()
        print(f"noise floor {tag}: 1 vs 8 threads max|d| = {float(d.max()):.3e}, max|d|/(|x|+1e-5) = {float(rel):.3e}, "
              f"max|x| = {float(res[1].abs().max()):.3f}")
    torch.set_num_threads(8)


if __name__ == "__main__":
    golden_steps()
    noise_floor()
