"""Golden vectors for the PDE residuals: runs the reference's own models/pde_loss.py classes (build container only),
cross-checks oracle/pde_oracle.py bit for bit, writes tests/golden/pde.npz (outputs only; inputs are regenerated from
the tagged streams of oracle/fixtures.py).

    python oracle/make_golden_pde.py
"""
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("MCEDM_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from models.normalizer import Normalizer          # reference
from models.pde_loss import DarcyLoss, SweFvLoss   # reference
from oracle import fixtures as fx
from oracle import pde_oracle as po

out = {}
for name, (B, T, X, Tn, xmin, xmax) in fx.PDE_SWE_CASES.items():
    pred, gt, sh, su = fx.pde_swe_inputs(name)
    ref = SweFvLoss(Tn=Tn, x_min=xmin, x_max=xmax)
    nh, nu = Normalizer(0.0, sh), Normalizer(0.0, su)
    for clamp in (False, True):
        r = ref(pred.clone(), gt.clone(), nh, nu, return_d=False, calc_prob=False, clamp_loss=clamp)
        o = po.swe_fv_residual(pred, gt, sh, su, Tn, xmin, xmax, 2, clamp)
        assert torch.equal(r, o), (name, clamp, float((r - o).abs().max()))
        out[f"swe_{name}_clamp{int(clamp)}"] = r.numpy()
    out[f"swe_{name}_step"] = ref.f_t_swp1d(pred, Tn / T).numpy()
    assert torch.allclose(ref.f_t_swp1d(pred, Tn / T), po.swe_fv_step(pred, Tn / T, xmin, xmax), rtol=0, atol=0, equal_nan=True)
for name, (B, S) in fx.PDE_DARCY_CASES.items():
    pred = fx.pde_darcy_inputs(name)
    ref = DarcyLoss()
    for clamp in (False, True):
        r = ref(pred.clone(), pred.clone(), None, None, return_d=False, calc_prob=False, clamp_loss=clamp)
        o = po.darcy_residual(pred, clamp)
        assert torch.equal(r, o), (name, clamp)
        out[f"darcy_{name}_clamp{int(clamp)}"] = r.numpy()
# ---- guidance gradients: forward(..., return_d=True) (models/pde_loss.py:231-242, 60-75) ---------------------------------
for name, (B, T, X, Tn, xmin, xmax) in fx.PDE_SWE_CASES.items():
    pred, gt, sh, su = fx.pde_swe_inputs(name)
    ref = SweFvLoss(Tn=Tn, x_min=xmin, x_max=xmax)
    nh, nu = Normalizer(0.0, sh), Normalizer(0.0, su)
    for tag, target in (("self", pred), ("gt", gt)):          # the sampler passes the state as its own target
        r = ref(pred.clone(), target.clone(), nh, nu, return_d=True, calc_prob=True)
        o = po.swe_fv_guidance(pred, target, sh, su, Tn, xmin, xmax, 2)
        assert torch.equal(r, o), (name, tag, float((r - o).abs().max()))
        out[f"swe_{name}_d_{tag}"] = r.numpy()
for name, (B, S) in fx.PDE_DARCY_CASES.items():
    if S < 8:
        continue
    pred = fx.pde_darcy_inputs(name)
    ref = DarcyLoss()
    for prob in (False, True):
        r = ref(pred.clone(), pred.clone(), None, None, return_d=True, calc_prob=prob)
        o = po.darcy_guidance(pred, prob)
        assert torch.equal(r, o), (name, prob, float((r - o).abs().max()))
        out[f"darcy_{name}_d_prob{int(prob)}"] = r.numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "pde.npz"), **out)
print("wrote tests/golden/pde.npz:", {k: v.shape for k, v in out.items()})
