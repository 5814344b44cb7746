"""Golden vectors for the dx-conditioned network (second clause), made by RUNNING THE REFERENCE
(build container only): hparams.model.dx_cond = True in both of its forms -- cat_dx=True (dx concatenated to conv_in's input,
models/adm_blocks.py:238, 334-339) and cat_dx=False (dx_enc = Conv3x3 -> GELU -> Conv3x3 and combine_enc, :266-280, 352-362):

  * DhariwalUNet.forward(x, noise_labels, cond, dx=dx) with dx given and dx=None (:364-388)
  * PlCondEdm.get_denoised(..., dx=dx, w=0.5): the classifier-free branch drops cond AND dx (models/ddim.py:1745-1763)
  * PlCondEdm.sample_edm of a dx_cond model, dx_norm='prob': dx_in = get_dx_input(h, x_hat) before every denoiser call
    (:601-613, 1424-1450, 1571, 1584), SWE and Darcy residuals; one case with guide_dx=True on top
  * PlCondEdm.training_step with the dx branch taken (torch.rand(1) > 0.1, :1673-1681) and not taken: loss and every gradient,
    incl. dx_enc.* / combine_enc.* (dx itself carries no gradient: torch.autograd.grad without create_graph, pde_loss.py)

The other dx_norm values cannot run for the single-task model in the reference: PlCondEdm.get_dx_pde returns a 3-D tensor
with calc_prob=False and get_dx_input fails to unpack it (recorded as 'dx_norm_l2_raises').

    python oracle/make_golden_dxcond.py        # rewrites tests/golden/dxcond.npz
"""
import dataclasses

import make_golden as mg

import torch

from models.ddim import PlCondEdm   # reference
from oracle import fixtures as fx
from oracle import mcedm_oracle as orc


def module(cfg, P, sp, system, st, dx_norm="prob"):
    hp = mg.make_cond_hparams(cfg, sp)
    hp.model.update(dx_cond=True, cat_dx=cfg.dx_mode == "cat", dx_norm=dx_norm, dx_detach=True)
    m = PlCondEdm(hp)
    assert [(n, tuple(p.shape)) for n, p in m.model.named_parameters()] == [(n, tuple(s)) for n, s in orc.param_shapes(cfg)], \
        "param_shapes drifted from the reference"
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
        for n, p in m.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    m.set_pde_loss_function(system, False)
    m.h_ch, m.u_ch = 1, 1
    return m


class _Rand:
    """torch.rand(1) -> a fixed value (the dx / cond_p coin flips of PlCondEdm.forward)."""

    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self._r = torch.rand
        torch.rand = lambda *a, **k: torch.tensor([self.value])
        return self

    def __exit__(self, *a):
        torch.rand = self._r


def main():
    out = {}
    st = fx.STEP_NORM_STATS
    for mode in ("cat", "enc"):
        cfg = dataclasses.replace(fx.CFG_C, dx_channels=1, dx_mode=mode)
        P = orc.make_params(cfg, 17)
        # ---- network and preconditioning
        m = module(cfg, P, mg.sampler_dict(), "swe_per", st)
        x, cond, dx, sig = fx.dxcond_net_inputs()
        with torch.no_grad():
            for tag, d in (("dx", dx), ("none", None)):
                F = m.model(x, sig.log() / 4, cond, dx=d)
                mg.check(f"{mode} DhariwalUNet.forward dx={tag}", orc.unet_forward(P, cfg, x, sig.log() / 4, cond, dx=d), F)
                out[f"{mode}_F_{tag}"] = F
            D, Fw = m.get_denoised(m.model, x.double(), sig, cond=cond, dx=dx, w=0.5)
            Do, Fo = orc.get_denoised(P, cfg, x.double(), sig, cond=cond, w=0.5, dx=dx)
            mg.check(f"{mode} get_denoised w=0.5 dx", Do, D)
            out[f"{mode}_D_w"], out[f"{mode}_F_w"] = D, Fw
        # ---- sampler
        for system in ("swe_per", "darcy"):
            for guided in ((False, True) if (mode, system) == ("enc", "swe_per") else (False,)):
                sp = mg.sampler_dict(guide_dx=guided)
                m = module(cfg, P, sp, system, st)
                h, u_noise, steps = fx.cond_sampler_inputs("det")
                with torch.no_grad(), mg._Inject(steps):
                    xs = m.sample_edm(h, u_noise, mg._wrap(sp), return_last=False, guide_dx=guided)
                hc = h.permute(0, 3, 1, 2)
                g = lambda hh, d: orc.guidance_dx_cond(system, hh, d, st)      # noqa: E731
                with torch.no_grad():
                    xo = orc.sample_edm_cond(P, cfg, hc, orc.SamplerParams(), u_noise.permute(0, 3, 1, 2), steps, return_last=False,
                                             guidance=g if guided else None, dx_input=g)
                    x0 = orc.sample_edm_cond(P, cfg, hc, orc.SamplerParams(), u_noise.permute(0, 3, 1, 2), steps)
                key = f"{mode}_{system}{'_guided' if guided else ''}"
                mg.check(f"PlCondEdm.sample_edm dx_cond {key}", xo, xs, rtol=1e-4, atol=1e-5 * float(xs.abs().max()))
                print(f"  dx conditioning moves the sample by max {float((xs[:, -1:] - x0).abs().max()):.3e} (max|x| {float(xs.abs().max()):.2f})")
                out[f"{key}_xs_last"] = xs[:, -1:].contiguous()
                out[f"{key}_xs_traj"] = xs[:, ::6].contiguous()
        # ---- training step, dx branch taken (rand = 0.5) and not (rand = 0.05)
        h, u, noise, rnd_normal = fx.cond_training_inputs()
        ts = fx.TRAIN_NORM_STATS
        hn, un = ((h - ts[0]) / ts[1]).permute(0, 3, 1, 2), ((u - ts[2]) / ts[3]).permute(0, 3, 1, 2)
        for tag, coin in (("on", 0.5), ("off", 0.05)):
            m = module(cfg, P, mg.sampler_dict(), "swe_per", ts)
            with mg._Inject([noise], [rnd_normal]), _Rand(coin):
                loss = m.training_step((h, None, None, u), 0)
            loss.backward()
            Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
            dxin = (lambda c, xn: orc.guidance_dx_cond("swe_per", c, xn, ts)) if tag == "on" else None
            lo = orc.training_loss_cond(Pg, cfg, un, hn, noise, rnd_normal, dx_input=dxin)
            lo.backward()
            mg.check(f"{mode} training_step loss (dx {tag})", lo, loss, rtol=1e-5, atol=1e-6)
            ref = {n: p.grad for n, p in m.model.named_parameters()}
            for n in ref:
                if ref[n] is None:       # dx_enc.* when the dx branch is not taken
                    assert Pg[n].grad is None or float(Pg[n].grad.abs().max()) == 0.0, n
                    continue
                mg.check(f"{mode} grad {n}", Pg[n].grad, ref[n], rtol=1e-3, rel_to_max=2e-6)
            out[f"{mode}_loss_{tag}"] = loss.detach()
            for n in fx.DXCOND_GRAD_NAMES[mode]:
                out[f"{mode}_grad_{tag}::{n}"] = torch.zeros_like(P[n]) if ref[n] is None else ref[n]
            out[f"{mode}_grad_sqnorm_{tag}"] = torch.tensor([0.0 if g is None else float((g.double() ** 2).sum()) for g in ref.values()])
    # the other normalisations cannot run for the single-task model
    cfg = dataclasses.replace(fx.CFG_C, dx_channels=1, dx_mode="cat")
    m = module(cfg, orc.make_params(cfg, 17), mg.sampler_dict(), "swe_per", st, dx_norm="l2")
    h, u_noise, steps = fx.cond_sampler_inputs("det")
    try:
        with torch.no_grad(), mg._Inject(steps):
            m.sample_edm(h, u_noise, mg._wrap(mg.sampler_dict()), return_last=True)
        raised = 0
    except ValueError as e:
        raised = 1
        print("  PlCondEdm.sample_edm with dx_norm='l2' raises in the reference:", str(e)[:90])
    out["dx_norm_l2_raises"] = torch.tensor(raised)
    mg.save("dxcond.npz", seed=17, **out)


if __name__ == "__main__":
    main()
