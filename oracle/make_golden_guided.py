"""Golden vectors for PDE guidance inside the sampler, made by RUNNING THE REFERENCE (build
container only): PlCondEdm.sample_edm(..., guide_dx=True) (models/ddim.py:1532-1601 with get_dx_log_prob :641-650 and
get_dx_pde :1424-1450) for the SWE and Darcy residuals.  (The joint model's hook, models/mcedm.py:500-518, slices the
last axis of an NCHW tensor and raises in the reference: the fact is recorded as 'joint_model_guidance_raises' and checked by
tests/test_hip_cond_edm.py::test_joint_model_guidance_is_rejected_like_the_reference.)

    python oracle/make_golden_guided.py        # rewrites tests/golden/guided.npz
"""
import make_golden as mg

import torch

from models.ddim import PlCondEdm   # reference
from models.mcedm import PlMcedm    # reference
from oracle import fixtures as fx
from oracle import mcedm_oracle as orc


def main():
    cfg = fx.CFG_C
    P = orc.make_params(cfg, 13)
    st = fx.STEP_NORM_STATS
    out = {}
    for system in ("swe_per", "darcy"):
        sp = mg.sampler_dict(guide_dx=True)
        m = PlCondEdm(mg.make_cond_hparams(cfg, sp))
        with torch.no_grad():
            for n, p in m.model.named_parameters():
                p.copy_(P[n])
            for n, p in m.ema_model.ma_model.named_parameters():
                p.copy_(P[n])
        m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
        m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
        m.set_pde_loss_function(system, False)
        m.h_ch, m.u_ch = 1, 1
        h, u_noise, steps = fx.cond_sampler_inputs("det")
        with torch.no_grad(), mg._Inject(steps):
            xs = m.sample_edm(h, u_noise, mg._wrap(sp), return_last=False, guide_dx=True)
        hc = h.permute(0, 3, 1, 2)
        with torch.no_grad():
            xo = orc.sample_edm_cond(P, cfg, hc, orc.SamplerParams(), u_noise.permute(0, 3, 1, 2), steps, return_last=False,
                                     guidance=lambda hh, d: orc.guidance_dx_cond(system, hh, d, st))
            x0 = orc.sample_edm_cond(P, cfg, hc, orc.SamplerParams(), u_noise.permute(0, 3, 1, 2), steps)
            x0_all = orc.sample_edm_cond(P, cfg, hc, orc.SamplerParams(), u_noise.permute(0, 3, 1, 2), steps, return_last=False)
        mg.check(f"PlCondEdm.sample_edm guide_dx {system}", xo, xs, rtol=1e-4, atol=1e-5 * float(xs.abs().max()))
        print(f"  guidance moves the sample by max {float((xs[:, -1:] - x0).abs().max()):.3e} (max|x| {float(xs.abs().max()):.2f})")
        out[f"{system}_xs_last"] = xs[:, -1:].contiguous()
        out[f"{system}_xs_traj"] = xs[:, ::6].contiguous()
        out[f"{system}_xs_head"] = xs[:, :3].contiguous()
        first = next((k for k in range(xs.shape[1]) if float((xs[:, k] - torch.as_tensor(x0_all[:, k])).abs().max()) > 0), -1)
        print(f"  first trajectory entry the guidance changes: {first}")
        out[f"{system}_first_guided_entry"] = torch.tensor(first)
    # the joint model's hook fails in the reference: record that fact (not a vector)
    pl = mg.build_reference(fx.CFG_P, seed=7, sampler=mg.sampler_dict(guide_dx=True))
    pl.set_pde_loss_function("swe_per", False)
    cond, mk, init, steps = fx.sampler_inputs("det_u", B=2)
    try:
        with torch.no_grad(), mg._Inject([init] + steps):
            pl.sample_edm(torch.zeros(2, 2, 32, 32), cond, mk, mg._wrap(mg.sampler_dict(guide_dx=True)), guide_dx=True)
        raised = 0
    except RuntimeError as e:
        raised = 1
        print("  PlMcedm.sample_edm(guide_dx=True) raises in the reference:", str(e)[:90])
    out["joint_model_guidance_raises"] = torch.tensor(raised)
    mg.save("guided.npz", seed=13, **out)


if __name__ == "__main__":
    main()
