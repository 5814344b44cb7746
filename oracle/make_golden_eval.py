"""Golden vectors for the evaluation loops of models/ddim.py, made by RUNNING THE REFERENCE (build container only):

  * PlDdim.test_step        models/ddim.py:372-533   type 'edm' (RePaint sampler), n_repeat 2 / 3, n_samples 1 / 5 --
                                                     the way BASELINE config 5 is driven (eval_model.py -> trainer.test)
  * PlDdim.validation_step  models/ddim.py:294-370   evaluated epoch; early return on the others
  * PlCondEdm.test_step     models/ddim.py:1219-1319 n_samples 2 (SWE) / 16 (Darcy: BASELINE config 4 read as the
                                                     single-task model) and a guided Darcy run with saturated residuals
  * PlCondEdm.validation_step  :1154-1217

Every random draw is injected, every ``self.log`` call is recorded.  The oracles (ddpm_oracle.eval_*,
mcedm_oracle.eval_cond_*) are cross-checked on every case before anything is written.

    python oracle/make_golden_eval.py        # rewrites tests/golden/eval_steps.npz
"""
import make_golden as mg
import make_golden_ddpm as mgd

import torch

from models.ddim import PlCondEdm   # reference
from oracle import ddpm_oracle as dorc
from oracle import fixtures as fx
from oracle import mcedm_oracle as orc


def _record(module, logs):
    module.log = lambda name, value, **k: logs.__setitem__(name, torch.as_tensor(value).detach().clone())


def _stats(module, st):
    module.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    module.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))


def _compare(what, o, res, logs, out, prefix):
    keys = sorted(res)
    assert keys == sorted(k for k in o if not k.startswith("log::")), (what, keys, sorted(o))
    for k in keys:
        tol = dict(rtol=1e-4, atol=1e-5 * max(1.0, float(torch.as_tensor(res[k]).abs().max())))
        mg.check(f"{what} {k}", o[k], res[k], **tol)
        out[f"{prefix}::{k}"] = res[k]
    for k, v in logs.items():
        if f"log::{k}" in o:
            ov = o[f"log::{k}"]
            if torch.isnan(v):
                assert torch.isnan(ov), (what, k)
            else:
                mg.check(f"{what} log {k}", ov, v, rtol=2e-3 if "pde" in k else 1e-4, atol=1e-6)
        out[f"{prefix}::log::{k}"] = v


def ddpm_cases(out):
    cfg = fx.CFG_D
    st = fx.EVAL_DDPM_STATS
    for tag, (system, n, N, R, churn, nth, ntu) in fx.EVAL_DDPM_CASES.items():
        sp = mgd.sampler_dict(timesteps=N, n_repeat=R, S_churn=churn, n_time_h=nth, n_time_u=ntu, n_samples=n)
        m, P = mgd.build(cfg, 21, sp)
        logs = {}
        _record(m, logs)
        _stats(m, st)
        m.set_pde_loss_function(system, False)
        h, u, init, steps, reps, _ = fx.eval_ddpm_inputs(tag)
        queue = [init]
        for i in range(N):
            queue += [steps[i]] + reps[i]
        with torch.no_grad(), mg._Inject(queue) as inj:
            res = m.test_step((h, None, None, u), 0)
        assert not inj.like_queue
        spo = dorc.RepaintParams(timesteps=N, n_repeat=R, S_churn=churn, n_time_h=nth, n_time_u=ntu)
        with torch.no_grad():
            o = dorc.eval_test_step(P, cfg, h, u, st, spo, n, system, init, steps, reps)
        o["log::test_mae_h"], o["log::test_mae_u"] = o["loss_h"], o["loss"]
        o["log::test_mae_h_un"], o["log::test_mae_u_un"] = o["loss_h_un"], o["loss_u_un"]
        o["log::test_mae_u_scaled"] = o["test_mae_u_scaled"]
        _compare(f"PlDdim.test_step[{tag}]", o, res, logs, out, f"ddpm_{tag}")
        assert tuple(res["traj"].shape) == (fx.EVAL_B, 1, cfg.resolution, cfg.resolution, n, 2)

    # the same loop with the default sampler type: diff_sampler: ddim_sampler -> sample_with_repeat (models/ddim.py:393-394)
    system, n, N, skip, eta, R, nth, ntu = fx.EVAL_DDIM
    sp = mgd.sampler_dict(type="ddim", skip_type=skip, eta=eta, timesteps=N, n_repeat=R, n_time_h=nth, n_time_u=ntu, n_samples=n)
    m, P = mgd.build(cfg, 21, sp)
    logs = {}
    _record(m, logs)
    _stats(m, st)
    m.set_pde_loss_function(system, False)
    h, u, init, _ = fx.ddim_inputs("eval", B=n * fx.EVAL_B)
    h, u = h[:fx.EVAL_B] * st[1] + st[0], u[:fx.EVAL_B] * st[3] + st[2]
    with torch.no_grad(), mg._Inject([init]) as inj:
        res = m.test_step((h, None, None, u), 0)
    assert not inj.like_queue and res["traj"].dtype == torch.float32
    spo = dorc.DdimParams(timesteps=N, skip_type=skip, eta=eta, n_repeat=R, n_time_h=nth, n_time_u=ntu)
    with torch.no_grad():
        o = dorc.eval_test_step(P, cfg, h, u, st, spo, n, system, init)
    o["log::test_mae_h"], o["log::test_mae_u"] = o["loss_h"], o["loss"]
    o["log::test_mae_h_un"], o["log::test_mae_u_un"] = o["loss_h_un"], o["loss_u_un"]
    o["log::test_mae_u_scaled"] = o["test_mae_u_scaled"]
    _compare("PlDdim.test_step[ddim sampler]", o, res, logs, out, "ddpm_ddim")

    system, n, N, R, churn, nth, ntu = fx.EVAL_DDPM_VAL
    sp = mgd.sampler_dict(timesteps=N, n_repeat=R, S_churn=churn, n_time_h=nth, n_time_u=ntu)
    m, P = mgd.build(cfg, 21, sp)
    logs = {}
    _record(m, logs)
    _stats(m, st)
    m.set_pde_loss_function(system, False)
    m.current_epoch = 0
    h, u, init, steps, reps, u_noise = fx.eval_ddpm_inputs("val")
    queue = [u_noise, init]
    for i in range(N):
        queue += [steps[i]] + reps[i]
    with torch.no_grad(), mg._Inject(queue) as inj:
        res = m.validation_step((h, None, None, u), 0)
    assert not inj.like_queue and res.pop("epoch") == 0
    with torch.no_grad():
        o = dorc.eval_validation_step(P, cfg, h, u, st, dorc.RepaintParams(timesteps=N, n_repeat=R, S_churn=churn, n_time_h=nth,
                                                                            n_time_u=ntu), system, u_noise, init, steps, reps)
    _compare("PlDdim.validation_step", o, res, logs, out, "ddpm_val")
    m.current_epoch = 7
    assert m.validation_step((h, None, None, u), 0) == {"epoch": 7}


def cond_module(sp, system, st, logs):
    cfg = fx.CFG_C
    P = orc.make_params(cfg, 13)
    m = PlCondEdm(mg.make_cond_hparams(cfg, sp))
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    _record(m, logs)
    _stats(m, st)
    m.set_pde_loss_function(system, False)
    m.current_epoch = 0
    return m, P


def cond_cases(out):
    cfg = fx.CFG_C
    zeros18 = lambda nB: [torch.zeros(nB, 1, 32, 32, dtype=torch.float64)] * 18          # S_churn = 0: the draws are multiplied by 0
    for tag, (system, n, guided, st) in fx.EVAL_COND_CASES.items():
        sp = mg.sampler_dict(n_samples=n, guide_dx=guided)
        logs = {}
        m, P = cond_module(sp, system, st, logs)
        m.set_test_sampler_params(mg._wrap(sp))
        h, u, init = fx.eval_cond_inputs(tag)
        with torch.no_grad(), mg._Inject([init] + zeros18(n * fx.EVAL_B)) as inj:
            res = m.test_step((h, None, None, u), 0)
        assert not inj.like_queue
        with torch.no_grad():
            o = orc.eval_cond_test_step(P, cfg, h, u, st, orc.SamplerParams(), n, system, init, guidance=guided)
        o["log::test_mae_u"], o["log::test_mae_u_un"], o["log::test_mae_u_scaled"] = o["loss"], o["loss_u_un"], o["test_mae_u_scaled"]
        _compare(f"PlCondEdm.test_step[{tag}]", o, res, logs, out, f"cond_{tag}")
        if guided:       # the guided run equals the unguided one: every cell of the log-probability form is saturated
            with torch.no_grad():
                o0 = orc.eval_cond_test_step(P, cfg, h, u, st, orc.SamplerParams(), n, system, init, guidance=False)
            assert torch.equal(o0["traj"], o["traj"]), "the saturated Darcy guidance must be exactly zero"
            print("  guided == unguided (saturated residuals): the guided loop runs and contributes exactly zero")

    logs = {}
    m, P = cond_module(mg.sampler_dict(), "swe_per", fx.STEP_NORM_STATS, logs)
    h, u, init = fx.eval_cond_inputs("val")
    with torch.no_grad(), mg._Inject([init] + zeros18(fx.EVAL_B)) as inj:
        res = m.validation_step((h, None, None, u), 0)
    assert not inj.like_queue and res.pop("epoch") == 0
    with torch.no_grad():
        o = orc.eval_cond_validation_step(P, cfg, h, u, fx.STEP_NORM_STATS, orc.SamplerParams(), "swe_per", init)
    o["log::val_mae_u"], o["log::val_mae_u_un"], o["log::val_mae_u_scaled"] = o["loss"], o["loss_u_un"], o["val_loss_u_scaled"]
    _compare("PlCondEdm.validation_step", o, res, logs, out, "cond_val")
    m.current_epoch = 7
    assert m.validation_step((h, None, None, u), 0) == {"epoch": 7}


if __name__ == "__main__":
    out = {}
    ddpm_cases(out)
    cond_cases(out)
    mg.save("eval_steps.npz", **out)
