"""PDE residuals (forward): oracle vs the reference's golden vectors on CPU; HIP kernels vs the
oracle and the golden vectors through the C ABI on the GPU (bit-exact: same evaluation order, no fma contraction)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fixtures as fx          # noqa: E402
from oracle import pde_oracle as po        # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "pde.npz"))


def same(a, b):
    a, b = torch.as_tensor(a).cpu(), torch.as_tensor(b).cpu()
    return a.shape == b.shape and torch.allclose(a, b, rtol=0, atol=0, equal_nan=True)


@pytest.mark.parametrize("name", list(fx.PDE_SWE_CASES))
def test_oracle_swe_matches_reference_golden(name):
    B, T, X, Tn, xmin, xmax = fx.PDE_SWE_CASES[name]
    pred, gt, sh, su = fx.pde_swe_inputs(name)
    assert same(po.swe_fv_step(pred, Tn / T, xmin, xmax), GOLD[f"swe_{name}_step"])
    for clamp in (0, 1):
        assert same(po.swe_fv_residual(pred, gt, sh, su, Tn, xmin, xmax, 2, bool(clamp)), GOLD[f"swe_{name}_clamp{clamp}"])


@pytest.mark.parametrize("name", list(fx.PDE_DARCY_CASES))
def test_oracle_darcy_matches_reference_golden(name):
    pred = fx.pde_darcy_inputs(name)
    for clamp in (0, 1):
        assert same(po.darcy_residual(pred, bool(clamp)), GOLD[f"darcy_{name}_clamp{clamp}"])


class Norm:          # the two attributes of models/normalizer.py:Normalizer the losses read
    def __init__(self, divide):
        self.subtract = torch.tensor(0.0)
        self.divide = divide


@pytest.fixture(scope="module")
def pde():
    import mcedm_amd  # noqa: F401
    from mcedm_amd import pde_loss
    assert torch.cuda.is_available()
    return pde_loss


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(fx.PDE_SWE_CASES))
def test_hip_swe_residual_bit_exact(pde, name):
    B, T, X, Tn, xmin, xmax = fx.PDE_SWE_CASES[name]
    pred, gt, sh, su = fx.pde_swe_inputs(name)
    f = pde.SweFvLoss(Tn=Tn, x_min=xmin, x_max=xmax)
    assert same(f.f_t_swp1d(pred.cuda(), Tn / T), GOLD[f"swe_{name}_step"])
    for clamp in (0, 1):
        out = f(pred.cuda(), gt.cuda(), Norm(sh.cuda()), Norm(su.cuda()), return_d=False, calc_prob=False, clamp_loss=bool(clamp))
        assert same(out, GOLD[f"swe_{name}_clamp{clamp}"])
        assert same(out, po.swe_fv_residual(pred, gt, sh, su, Tn, xmin, xmax, 2, bool(clamp)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(fx.PDE_DARCY_CASES))
def test_hip_darcy_residual_bit_exact(pde, name):
    pred = fx.pde_darcy_inputs(name)
    f = pde.DarcyLoss()
    for clamp in (0, 1):
        assert same(f(pred.cuda(), pred.cuda(), None, None, clamp_loss=bool(clamp)), GOLD[f"darcy_{name}_clamp{clamp}"])


@pytest.mark.gpu
def test_hip_swe_flip_and_full_size_properties(pde):
    # flip_xy: channels arrive as (u, h); a consistent trajectory (gt = step of itself) has zero residual after row 0
    B, T, X = 4, 128, 128
    g = torch.Generator().manual_seed(0)
    h = 1.5 + 0.2 * torch.rand(B, T, X, generator=g)
    u = 0.1 * torch.randn(B, T, X, generator=g)
    s = torch.stack((h, u), -1).cuda()
    f = pde.SweFvLoss(Tn=0.128, x_min=-0.5, x_max=0.5)
    nxt = f.f_t_swp1d(s, 0.128 / T)
    gt = torch.cat((s[:, :1], nxt[:, :-1]), 1)
    r = f(s, gt, Norm(torch.tensor(1.0).cuda()), Norm(torch.tensor(1.0).cuda()))
    assert float(r.abs().max()) == 0.0
    ff = pde.SweFvLoss(Tn=0.128, x_min=-0.5, x_max=0.5, flip_xy=True)
    r2 = ff(s.flip(-1), gt.flip(-1), Norm(torch.tensor(1.0).cuda()), Norm(torch.tensor(1.0).cuda()))
    assert float(r2.abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        f(s, gt, Norm(torch.tensor(1.0).cuda()), Norm(torch.tensor(1.0).cuda()), return_d=True)
