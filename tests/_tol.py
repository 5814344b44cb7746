"""Tolerance helper shared by the sampler-trajectory tests: every entry of a trajectory [b, t, ...] is held to
the north_star bar relative to ITS OWN magnitude -- rtol 1e-4, atol 1e-5 x max|that entry| -- not to the trajectory's maximum
(the sigma = 80 initial state is ~100x larger than the final one, which a trajectory-wide atol lets through 6-10x looser)."""
import torch


def close_per_entry(got, ref, rtol=1e-4, rel_atol=1e-5, what="", time_dim=1):
    got = torch.as_tensor(got).detach().cpu().double()
    ref = torch.as_tensor(ref).detach().cpu().double()
    assert got.shape == ref.shape, (what, tuple(got.shape), tuple(ref.shape))
    worst = 0.0
    for t in range(ref.shape[time_dim]):
        g, r = got.select(time_dim, t), ref.select(time_dim, t)
        atol = rel_atol * float(r.abs().max())
        err = (g - r).abs()
        lim = atol + rtol * r.abs()
        bad = err > lim
        worst = max(worst, float((err / lim.clamp_min(1e-300)).max()))
        assert not bad.any(), (f"{what}: entry {t} of {ref.shape[time_dim]}: {int(bad.sum())}/{bad.numel()} outside rtol {rtol} / atol "
                               f"{atol:.3e} (= {rel_atol} x max|entry| {float(r.abs().max()):.3e}); max err {float(err.max()):.3e}")
    return worst
