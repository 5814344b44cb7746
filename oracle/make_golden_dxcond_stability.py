"""How reproducible are the REFERENCE's own Darcy dx_cond trajectories?  

tests/test_hip_dxcond.py compares the device's Darcy-conditioned trajectories with the reference's at the north_star bar only
through step 6 and allows a few per cent of the final state's entries outside it, on the grounds that the network input
there -- the log-probability residual gradient, a step function of the residual with a 1e-5 wide transition
(models/pde_loss.py:60-75) -- flips whole cells on last-bit differences.  This script puts a number on that claim by running
the reference AGAINST ITSELF (build container only):

  * the same trajectory with torch on 1 thread instead of 8 (a different summation order inside the convolutions),
  * with one input perturbed by one unit in the last place: the conditioning field h, the initial noise, one conv weight.

For every variant it counts the entries of the FINAL state that differ from the unperturbed 8-thread run by more than the
bar the GPU test uses, and the first sampled trajectory index (0, 6, 12, 18) at which any entry does -- twice: against the
round-4 bar (1e-5 * max|trajectory| + 1e-4 |ref|; keys `final_bad_frac`, `first_bad_step`) and against the round-5 bar that
holds every sampled step to ITS OWN magnitude (1e-5 * max|that step's state| + 1e-4 |ref|; keys `*_own`; the final state
is ~60x smaller than the sigma = 80 initial one).  Written to tests/golden/dxcond_stability.npz; the GPU test takes its
allowance from the `_own` numbers.

    cd oracle && PYTHONPATH=/root/repo python make_golden_dxcond_stability.py
"""
import dataclasses

import make_golden as mg

import numpy as np
import torch

import make_golden_dxcond as mgd
from oracle import fixtures as fx
from oracle import mcedm_oracle as orc


def run(cfg, P, system, st, h, u_noise, steps):
    sp = mg.sampler_dict()
    m = mgd.module(cfg, P, sp, system, st)
    with torch.no_grad(), mg._Inject(list(steps)):
        return m.sample_edm(h, u_noise, mg._wrap(sp), return_last=False)


def ulp(t, k=1):
    """t with every element moved k units in the last place towards +inf."""
    out = t.clone()
    for _ in range(k):
        out = torch.nextafter(out, torch.full_like(out, float("inf")))
    return out


def main():
    st = fx.STEP_NORM_STATS
    out = {}
    for mode in ("cat", "enc"):
        cfg = dataclasses.replace(fx.CFG_C, dx_channels=1, dx_mode=mode)
        P = orc.make_params(cfg, 17)
        h, u_noise, steps = fx.cond_sampler_inputs("det")
        for system in ("darcy", "swe_per"):
            torch.set_num_threads(8)
            base = run(cfg, P, system, st, h, u_noise, steps)
            scale = float(base[:, ::6].abs().max())
            variants = {}
            torch.set_num_threads(1)
            variants["threads1"] = run(cfg, P, system, st, h, u_noise, steps)
            torch.set_num_threads(8)
            variants["h_ulp"] = run(cfg, P, system, st, ulp(h), u_noise, steps)
            variants["noise_ulp"] = run(cfg, P, system, st, h, ulp(u_noise), steps)
            P2 = dict(P)
            name = "enc.32x32_block0.conv0.weight" if "enc.32x32_block0.conv0.weight" in P else sorted(k for k in P if k.endswith("conv0.weight"))[0]
            P2[name] = ulp(P[name])
            variants["weight_ulp"] = run(cfg, P2, system, st, h, u_noise, steps)
            key = f"{mode}_{system}"
            fracs, firsts, fracs_own, firsts_own = [], [], [], []
            for vname, xs in variants.items():
                bad = (xs - base).abs() > 1e-5 * scale + 1e-4 * base.abs()
                frac = float(bad[:, -1].double().mean())
                per = [bool(bad[:, i].any()) for i in (0, 6, 12, 18)]
                first = ([i for i, b in zip((0, 6, 12, 18), per) if b] + [-1])[0]
                own = [(xs[:, i] - base[:, i]).abs() > 1e-5 * float(base[:, i].abs().max()) + 1e-4 * base[:, i].abs() for i in (0, 6, 12, 18)]
                fracs_own.append(float(own[-1].double().mean()))
                firsts_own.append(([i for i, b in zip((0, 6, 12, 18), own) if bool(b.any())] + [-1])[0])
                print(f"      per-step bar: {fracs_own[-1] * 100:6.3f} % of the final state outside; first sampled step outside: {firsts_own[-1]}")
                rel = float(((xs[:, -1] - base[:, -1]).abs().max()) / scale)
                print(f"  reference vs itself, {key:14s} {vname:11s}: {frac * 100:6.3f} % of the final state outside the bar "
                      f"(max |d| / max|x| = {rel:.2e}); first sampled step outside: {first}")
                fracs.append(frac)
                firsts.append(first)
            out[f"{key}::variants"] = np.array(list(variants))
            out[f"{key}::final_bad_frac"] = np.array(fracs)
            out[f"{key}::first_bad_step"] = np.array(firsts)
            out[f"{key}::final_bad_frac_own"] = np.array(fracs_own)
            out[f"{key}::first_bad_step_own"] = np.array(firsts_own)
    mg.save("dxcond_stability.npz", **out)


if __name__ == "__main__":
    main()
