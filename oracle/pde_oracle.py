"""CPU restatement of the reference's PDE residuals (models/pde_loss.py) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(m-cedm_amd/pde_loss.py -> libmcedm_hip.so) never does.  Pinned by tests/golden/pde.npz, which
oracle/make_golden_pde.py writes by running the reference's own classes.

Covered (forward residual only; the guidance gradient `return_d=True` is SURVEY.md section 8 f3, not built yet):
  swe_fv_step        SweFvLoss.f_t_swp1d        models/pde_loss.py:131-165 (FORCE finite-volume step along x)
  swe_fv_residual    SweFvLoss.calculate_loss   models/pde_loss.py:211-225, forward :227-249
  darcy_residual     DarcyLoss.calculate_loss   models/pde_loss.py:30-54,  forward :56-88
"""
import torch
import torch.nn.functional as F


def swe_grid_dx(nx, x_min, x_max, n_ghosts, like):
    """dx exactly as gen_x builds it (models/pde_loss.py:102-118): fp32 linspace, x[1] - x[0]."""
    step = (x_max - x_min) / nx
    n = nx + 2 * n_ghosts
    if n % 2 == 0:
        x = torch.linspace(x_min + step / 2 - step * n_ghosts, x_max - step / 2 + step * n_ghosts, n)
    else:
        x = torch.linspace(x_min - step * n_ghosts, x_max + step * n_ghosts, n)
    x = x.type_as(like)
    return x[1] - x[0]


def swe_fv_step(s_t, dt, x_min=-2.5, x_max=2.5, n_ghosts=2, g=1.0, eps=1e-8):
    """models/pde_loss.py:131-165.  s_t (b, t, x, 2) = (h, u) -> one FORCE step for every (b, t) row."""
    nx = s_t.shape[2]
    dx = swe_grid_dx(nx, x_min, x_max, n_ghosts, s_t)
    ext = F.pad(s_t, (0, 0, n_ghosts, n_ghosts), mode="replicate")          # :120-129
    h = ext[..., 0]
    hu = ext[..., 1] * ext[..., 0]
    hm = 0.5 * (h[..., :-1] + h[..., 1:]) - 0.5 * dt * (hu[..., 1:] - hu[..., :-1]) / dx
    hum_upd = hu ** 2 / (h + eps) + 0.5 * g * h ** 2
    hum = 0.5 * (hu[..., :-1] + hu[..., 1:]) - 0.5 * dt * (hum_upd[..., 1:] - hum_upd[..., :-1]) / dx
    h_next = 0.5 * (hm[..., :-1] + hm[..., 1:]) - 0.5 * dt * (hum[..., 1:] - hum[..., :-1]) / dx
    hu_upd = hum ** 2 / (hm + eps) + 0.5 * g * hm ** 2
    hu_next = 0.5 * (hum[..., :-1] + hum[..., 1:]) - 0.5 * dt * (hu_upd[..., 1:] - hu_upd[..., :-1]) / dx
    h_out = h_next[..., n_ghosts - 1:-n_ghosts + 1]
    u_out = hu_next[..., n_ghosts - 1:-n_ghosts + 1] / (h_out + eps)
    return torch.stack((h_out, u_out), dim=-1)


def swe_fv_residual(pred, gt, scale_h, scale_u, Tn=0.128, x_min=-2.5, x_max=2.5, n_ghosts=2, clamp_loss=False):
    """models/pde_loss.py:211-225 + the return_d=False branch of forward (:243-247), flip_xy = False.
    pred, gt (b, t, x, 2); scale_* = normalizer.divide (0-dim)."""
    n_times = pred.shape[1]
    dt = Tn / n_times
    nxt = swe_fv_step(pred, dt, x_min, x_max, n_ghosts)
    with_ic = torch.cat((pred[:, 0:1], nxt[:, :-1]), dim=1)
    with_ic[torch.isnan(with_ic)] = 0.0
    scale = torch.stack((torch.as_tensor(scale_h, dtype=pred.dtype), torch.as_tensor(scale_u, dtype=pred.dtype)), dim=-1) ** 2
    loss = (with_ic - gt) ** 2 / scale
    if clamp_loss:
        loss = torch.clamp(loss, max=1.0)
    return loss


def darcy_residual(pred, clamp_loss=False, D=1.0):
    """models/pde_loss.py:30-54 + forward :80-86.  pred (b, s, s, 2) = (a, u) -> (b, s-4, s-4), divided by (s-4)^2."""
    b, size = pred.shape[0], pred.shape[1]
    a = pred[..., 0].reshape(b, size, size)
    u = pred[..., 1].reshape(b, size, size)
    dx = D / size
    dy = dx
    ux = (u[:, 2:, 1:-1] - u[:, :-2, 1:-1]) / (2 * dx)
    uy = (u[:, 1:-1, 2:] - u[:, 1:-1, :-2]) / (2 * dy)
    a = a[:, 1:-1, 1:-1]
    aux = a * ux
    auy = a * uy
    auxx = (aux[:, 2:, 1:-1] - aux[:, :-2, 1:-1]) / (2 * dx)
    auyy = (auy[:, 1:-1, 2:] - auy[:, 1:-1, :-2]) / (2 * dy)
    Du = -(auxx + auyy)
    loss = (Du - 1.0) ** 2
    _, t, n = loss.shape
    loss = loss / (t * n)
    if clamp_loss:
        loss = torch.clamp(loss, max=1.0)
    return loss


# --------------------------------------------------------------------------- #
# guidance gradients: the return_d=True branches
# --------------------------------------------------------------------------- #
def swe_fv_guidance(pred, gt, scale_h, scale_u, Tn=0.128, x_min=-2.5, x_max=2.5, n_ghosts=2):
    """SweFvLoss.forward(return_d=True), models/pde_loss.py:231-242: d mean(calculate_loss(pred, gt)) / d pred with NaNs
    of the gradient set to zero.  `gt` carries no gradient (the caller passes the same values as a separate tensor)."""
    with torch.enable_grad():
        p = pred.detach().clone().requires_grad_(True)
        loss = swe_fv_residual(p, gt.detach(), scale_h, scale_u, Tn, x_min, x_max, n_ghosts, clamp_loss=False).mean()
        d = torch.autograd.grad(loss, p)[0]
    d[torch.isnan(d)] = 0.0
    return d


def darcy_guidance(pred, calc_prob=False, D=1.0):
    """DarcyLoss.forward(return_d=True), models/pde_loss.py:60-75: gradient of mean(L) or, with calc_prob, of
    mean(log(2 (1 - sigmoid(1e5 L)) + 1e-12)), L = (Du - 1)^2 (NOT divided by the number of locations)."""
    with torch.enable_grad():
        p = pred.detach().clone().requires_grad_(True)
        b, size = p.shape[0], p.shape[1]
        a = p[..., 0].reshape(b, size, size)
        u = p[..., 1].reshape(b, size, size)
        dx = D / size
        ux = (u[:, 2:, 1:-1] - u[:, :-2, 1:-1]) / (2 * dx)
        uy = (u[:, 1:-1, 2:] - u[:, 1:-1, :-2]) / (2 * dx)
        ai = a[:, 1:-1, 1:-1]
        aux, auy = ai * ux, ai * uy
        auxx = (aux[:, 2:, 1:-1] - aux[:, :-2, 1:-1]) / (2 * dx)
        auyy = (auy[:, 1:-1, 2:] - auy[:, 1:-1, :-2]) / (2 * dx)
        L = (-(auxx + auyy) - 1.0) ** 2
        if calc_prob:
            L = torch.log(2 * (1.0 - torch.sigmoid(1e5 * L)) + 1e-12)
        d = torch.autograd.grad(L.mean(), p)[0]
    d[torch.isnan(d)] = 0.0
    return d
