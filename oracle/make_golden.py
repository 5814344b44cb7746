"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference); the fixtures it
writes are data (inputs + the reference's outputs), never reference source.
It also cross-checks ``oracle/mcedm_oracle.py`` against the reference on every
captured case and aborts on a mismatch, so a committed fixture set implies the
oracle was pinned when it was made.

    python oracle/make_golden.py            # rewrites tests/golden/*.npz

Reference entry points exercised (paths relative to the reference checkout):
models/adm_blocks.py (GroupNorm, Conv2d, AttentionOp, PositionalEmbedding,
UNetBlock, DhariwalUNet), models/mcedm.py (model_precond, get_denoised,
training_step, sample_edm), models/ddim_blocks.py (EmaModel), torch.optim.Adam.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("MCEDM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

# --- in-memory stand-in for the absent pytorch_lightning package ---
_pl = types.ModuleType("pytorch_lightning")


class _LightningModule(torch.nn.Module):
    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass


_pl.LightningModule = _LightningModule
sys.modules["pytorch_lightning"] = _pl

from models import adm_blocks as ref_blocks          # noqa: E402  (reference)
from models.mcedm import PlMcedm                      # noqa: E402  (reference)
from models.ddim import PlCondEdm                     # noqa: E402  (reference)

from oracle import mcedm_oracle as orc                # noqa: E402
from oracle import fixtures as fx                     # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__

    def __hasattr__(self, k):
        return k in self


def _wrap(d):
    if isinstance(d, dict):
        return AttrDict({k: _wrap(v) for k, v in d.items()})
    return d


def make_hparams(cfg: orc.UNetConfig, sampler: dict):
    """Mirror of configs/model/adm_edm_mcedm_res32.yaml as an attribute dict."""
    return _wrap(dict(
        name="adm_edm_mcedm",
        model=dict(in_channels=cfg.in_channels, cond_channels=cfg.cond_channels, cat_cond=True, out_ch=cfg.out_ch,
                   ch=cfg.ch, ch_mult=list(cfg.ch_mult), num_res_blocks=cfg.num_res_blocks,
                   attn_resolutions=list(cfg.attn_resolutions), dropout=0.0, label_dim=0, augment_dim=0,
                   label_dropout=0, ema_rate=0.999, ema=True, resamp_with_conv=True, resolution=cfg.resolution,
                   self_cond=False, cond_p=1.0, dx_cond=False, cat_dx=False, dx_norm="l2", dx_detach=False,
                   add_cond_mask=False, add_xt=False),
        data=dict(normalization="gauss", uniform_dequantization=False, gaussian_dequantization=False,
                  rescaled=False),
        optimization=dict(optimizer="Adam", lr=0.0002, weight_decay=0.0, beta1=0.9, amsgrad=False, eps=1e-8,
                          grad_clip=1.0, loss="l2", pde_loss_lambda=0.0, pde_loss_prop_t=False, use_gt_pde=False,
                          factor=0.3, step_size=50),
        sampler=sampler,
    ))


def sampler_dict(**over):
    d = dict(name="edm", type="edm", timesteps=18, sigma_min=0.002, sigma_max=80, rho=7, S_churn=0.0, S_min=0,
             S_max="inf", S_noise=1, n_samples=1, n_repeat=2, n_time_h=128, n_time_u=0, return_last=True,
             select_by_pde=False, use_gt_pde_select=True, guide_dx=False, w=0.0, plot_scaled=False)
    d.update(over)
    return d


def build_reference(cfg: orc.UNetConfig, seed: int, sampler=None) -> PlMcedm:
    hp = make_hparams(cfg, sampler or sampler_dict())
    pl_mod = PlMcedm(hp)
    # parameter names / shapes of the oracle must match the reference state_dict exactly
    ref_named = [(n, tuple(p.shape)) for n, p in pl_mod.model.named_parameters()]
    assert ref_named == [(n, tuple(s)) for n, s in orc.param_shapes(cfg)], "param_shapes drifted from reference"
    P = orc.make_params(cfg, seed)
    with torch.no_grad():
        for n, p in pl_mod.model.named_parameters():
            p.copy_(P[n])
        for n, p in pl_mod.ema_model.ma_model.named_parameters():
            p.copy_(P[n])
    pl_mod.h_ch, pl_mod.u_ch = 1, 1
    return pl_mod


def check(name, got, ref, rtol=1e-5, atol=1e-6, rel_to_max=0.0):
    got, ref = torch.as_tensor(got), torch.as_tensor(ref)
    err = (got.double() - ref.double()).abs().max().item()
    scale = ref.double().abs().max().item()
    ok = torch.allclose(got.double(), ref.double(), rtol=rtol, atol=atol + rel_to_max * scale)
    print(f"  oracle-vs-reference {name:44s} max|d|={err:.3e} (max|ref|={scale:.3e}) {'OK' if ok else 'MISMATCH'}")
    if not ok:
        raise SystemExit(f"oracle disagrees with reference on {name}")


def save(fname, **arrays):
    path = os.path.join(OUT, fname)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# --------------------------------------------------------------------------- #
def golden_ops():
    """Per-op vectors: GroupNorm, Conv2d k=3 plain/up/down, k=1, k=0 up/down, attention fwd+bwd, pos-emb.
    Inputs/parameters come from oracle/fixtures.py tags; only the reference outputs are stored."""
    out = {}
    for C in (64, 128, 256):                       # 16x4, 32x4, 32x8 groups
        gn = ref_blocks.GroupNorm(C)
        with torch.no_grad():
            gn.weight.copy_(fx.param(f"ops/gn{C}", "norm.weight", (C,)))
            gn.bias.copy_(fx.param(f"ops/gn{C}", "norm.bias", (C,)))
        x = fx.randn(f"ops/gn{C}/x", 2, C, 8, 8) * 1.5 + 0.3
        y = gn(x)
        check(f"group_norm C={C}", orc.group_norm(x, gn.weight, gn.bias), y)
        out[f"gn{C}_y"] = y
    for tag, kw in fx.CONV_CASES.items():
        cin, cout = fx.conv_channels(tag)
        conv = ref_blocks.Conv2d(cin, cout, **kw)
        w = b = None
        if conv.weight is not None:
            w = fx.param(f"ops/conv_{tag}", "conv.weight", tuple(conv.weight.shape))
            b = fx.param(f"ops/conv_{tag}", "conv.bias", (cout,))
            with torch.no_grad():
                conv.weight.copy_(w); conv.bias.copy_(b)
        x = fx.randn(f"ops/conv_{tag}/x", 2, cin, 8, 12)
        y = conv(x.clone())
        check(f"conv2d {tag}", orc.conv2d(x, w, b, up=kw.get("up", False), down=kw.get("down", False)), y)
        out[f"conv_{tag}_y"] = y
    # attention forward + the hand-written backward (adm_blocks.py:111-118)
    for T, (B, hw) in fx.ATTN_CASES.items():
        heads = 2
        qkv = fx.randn(f"ops/attn{T}/qkv", B, 3 * 128, *hw).requires_grad_(True)
        q, k, v = qkv.reshape(B * heads, 64, 3, -1).unbind(2)
        w = ref_blocks.AttentionOp.apply(q, k)
        a = torch.einsum("nqk,nck->ncq", w, v).reshape(B, 128, *hw)
        da = fx.randn(f"ops/attn{T}/da", *a.shape)
        (dqkv,) = torch.autograd.grad(a, qkv, da)
        qkv_o = qkv.detach().clone().requires_grad_(True)
        ao = orc.attention(qkv_o, heads)
        (dqkv_o,) = torch.autograd.grad(ao, qkv_o, da)
        check(f"attention fwd T={T}", ao, a)
        check(f"attention bwd T={T}", dqkv_o, dqkv, rel_to_max=1e-6)
        out.update({f"attn{T}_a": a, f"attn{T}_dqkv": dqkv})
    pe = ref_blocks.PositionalEmbedding(64)
    y = pe(fx.PE_LABELS)
    check("positional_embedding", orc.positional_embedding(fx.PE_LABELS, 64), y)
    out["pe_y"] = y
    save("ops.npz", **out)


def golden_blocks():
    """One UNetBlock of each shape class of SURVEY.md §3.4, emb rows n=1 (sampling) and n=B (training)."""
    out = {}
    init = dict(init_mode="kaiming_uniform", init_weight=np.sqrt(1 / 3), init_bias=np.sqrt(1 / 3))
    for tag, c in fx.BLOCK_CASES.items():
        blk = ref_blocks.UNetBlock(in_channels=c["cin"], out_channels=c["cout"], up=c.get("up", False),
                                   down=c.get("down", False), attention=c.get("attn", False),
                                   emb_channels=64, channels_per_head=64, dropout=0.0, init=init,
                                   init_zero=dict(init_mode="kaiming_uniform", init_weight=0, init_bias=0))
        P = fx.block_params(tag)
        named = dict(blk.named_parameters())
        assert [f"blk.{n}" for n in named] == list(P), (list(named), list(P))
        with torch.no_grad():
            for n, p in named.items():
                p.copy_(P[f"blk.{n}"])
        spec = fx.block_spec(tag)
        for nemb in (1, 2):
            x, emb = fx.block_inputs(tag, nemb)
            y = blk(x.clone(), emb)
            check(f"UNetBlock {tag} n_emb={nemb}", orc.unet_block(P, spec, x, emb), y, rtol=1e-4, atol=1e-5)
            out[f"{tag}_n{nemb}_y"] = y
    save("blocks.npz", **out)


CFG_P = fx.CFG_P
CFG_W = fx.CFG_W


def golden_unet():
    """Whole DhariwalUNet.forward + model_precond / get_denoised at config P (B=4, 32x32, ch=64)."""
    pl_mod = build_reference(CFG_P, seed=7)
    P = orc.make_params(CFG_P, 7)
    B, H, W = 4, 32, 32
    x = fx.randn("unet_P/x", B, 2, H, W); cond = fx.randn("unet_P/cond", B, 2, H, W)
    out = {}
    with torch.no_grad():
        for tag, labels in fx.UNET_LABELS.items():
            y = pl_mod.model(x, labels, cond)
            check(f"DhariwalUNet.forward {tag}", orc.unet_forward(P, CFG_P, x, labels, cond), y, rtol=1e-4, atol=1e-5)
            out[f"F_{tag}"] = y
        y = pl_mod.model(x, torch.tensor([0.3]), None)       # cond None -> zeros (adm_blocks.py:328-331)
        check("DhariwalUNet.forward cond=None", orc.unet_forward(P, CFG_P, x, torch.tensor([0.3]), None), y,
              rtol=1e-4, atol=1e-5)
        out["F_nocond"] = y
        for i, s in enumerate(fx.PRECOND_SIGMAS):
            sig = torch.tensor(s)
            D = pl_mod.model_precond(x * (1 + s), sig, cond)
            check(f"model_precond sigma={s}", orc.model_precond(P, CFG_P, x * (1 + s), sig, cond), D,
                  rtol=1e-4, atol=1e-5)
            D2, F2 = pl_mod.get_denoised(pl_mod.ema_model, (x * (1 + s)).double(), sig.double(), cond=cond, w=0.0)
            check(f"get_denoised sigma={s}", orc.get_denoised(P, CFG_P, (x * (1 + s)).double(), sig.double(), cond)[0],
                  D2, rtol=1e-4, atol=1e-5)
            assert torch.equal(D, D2)
            out[f"D_sigma{i}"] = D
        sigB = fx.PRECOND_SIGMA_B.reshape(B, 1, 1, 1)
        D = pl_mod.model_precond(x, sigB, cond)
        check("model_precond sigma[B]", orc.model_precond(P, CFG_P, x, sigB, cond), D, rtol=1e-4, atol=1e-5)
        out["D_sigmaB"] = D
        D3, _ = pl_mod.get_denoised(pl_mod.ema_model, x.double(), torch.tensor(0.7).double(), cond=cond, w=0.5)
        check("get_denoised cfg w=0.5", orc.get_denoised(P, CFG_P, x.double(), torch.tensor(0.7).double(), cond, w=0.5)[0],
              D3, rtol=1e-4, atol=1e-5)
        out["D_cfg_w05"] = D3
    save("unet_P.npz", seed=7, **out)

    # wide variant (ch=128, 4 levels, attention at the 16x16-labelled level, 2 heads), tiny spatial size
    pl_w = build_reference(CFG_W, seed=11)
    Pw = orc.make_params(CFG_W, 11)
    xw = fx.randn("unet_W/x", 2, 2, 16, 16); cw = fx.randn("unet_W/cond", 2, 2, 16, 16)
    with torch.no_grad():
        yw = pl_w.model(xw, fx.UNET_W_LABELS, cw)
    check("DhariwalUNet.forward wide", orc.unet_forward(Pw, CFG_W, xw, fx.UNET_W_LABELS, cw), yw,
          rtol=1e-4, atol=1e-5)
    save("unet_W.npz", seed=11, F=yw)


class _Inject:
    """Replace torch.randn_like / torch.randn with a queue of injected tensors (SURVEY.md §7 'RNG')."""

    def __init__(self, like_queue, randn_queue=()):
        self.like_queue = list(like_queue)
        self.randn_queue = list(randn_queue)

    def __enter__(self):
        self._rl, self._rn = torch.randn_like, torch.randn

        def randn_like(t, **k):
            v = self.like_queue.pop(0)
            assert tuple(v.shape) == tuple(t.shape), (v.shape, t.shape)
            return v.to(t.dtype)

        def randn(*a, **k):
            return self.randn_queue.pop(0)

        torch.randn_like, torch.randn = randn_like, randn
        return self

    def __exit__(self, *a):
        torch.randn_like, torch.randn = self._rl, self._rn


def golden_sampler():
    """sample_edm, N=18: deterministic (S_churn=0) and stochastic (S_churn=15), injected noise."""
    B, H, W = 4, 32, 32
    P = orc.make_params(CFG_P, 7)
    out = {}
    for tag, (churn, mask_kind) in fx.SAMPLER_CASES.items():
        sp = sampler_dict(S_churn=churn)
        pl_mod = build_reference(CFG_P, seed=7, sampler=sp)
        cond, m, init, steps = fx.sampler_inputs(tag, B, H, W)
        with torch.no_grad(), _Inject([init] + steps):
            xs_all = pl_mod.sample_edm(torch.zeros(B, 2, H, W), cond, m, _wrap(sp), return_last=False)
        with torch.no_grad(), _Inject([init] + steps):
            xs_last = pl_mod.sample_edm(torch.zeros(B, 2, H, W), cond, m, _wrap(sp), return_last=True)
        assert xs_last.dtype == torch.float64 and tuple(xs_last.shape) == (B, 1, H, W, 2)
        assert torch.equal(xs_last[:, 0], xs_all[:, -1])
        obs = (m == 0).permute(0, 2, 3, 1)
        assert torch.equal(xs_last[:, 0][obs], cond.permute(0, 2, 3, 1).double()[obs]), "observed region must be preserved"
        spo = orc.SamplerParams(S_churn=churn)
        xo = orc.sample_edm(P, CFG_P, cond, m, spo, init, steps, return_last=False)
        check(f"sample_edm {tag} all steps", xo, xs_all, rtol=1e-3, atol=1e-4)
        check(f"sample_edm {tag} last", orc.sample_edm(P, CFG_P, cond, m, spo, init, steps), xs_last, rtol=1e-3, atol=1e-4)
        out.update({f"{tag}_xs_last": xs_last, f"{tag}_xs_traj": xs_all[:, ::6].contiguous()})
    out["t_steps"] = orc.edm_t_steps(18, 0.002, 80, 7)
    save("sampler_P.npz", seed=7, **out)


def golden_training():
    """training_step loss + gradients of named tensors; one Adam + EMA step (B=4, 32x32, ch=64)."""
    pl_mod = build_reference(CFG_P, seed=7)
    P = orc.make_params(CFG_P, 7)
    h, u, mask, cond_noise, noise, rnd_normal = fx.training_inputs()
    st = fx.TRAIN_NORM_STATS
    pl_mod.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    pl_mod.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    with _Inject([cond_noise, noise], [rnd_normal]):
        loss = pl_mod.training_step((h, None, None, u, mask), 0)
    loss.backward()
    ref_grads = {n: p.grad.detach().clone() for n, p in pl_mod.model.named_parameters()}
    # oracle
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xc, cond_in, mc = fx.training_nchw(h, u, mask, cond_noise)
    lo = orc.training_loss(Pg, CFG_P, xc, cond_in, mc, noise, rnd_normal)
    lo.backward()
    check("training_step loss", lo, loss, rtol=1e-5, atol=1e-6)
    for n in ref_grads:
        check(f"grad {n}", Pg[n].grad, ref_grads[n], rtol=1e-3, rel_to_max=2e-6)
    out = dict(loss=loss.detach())
    for n in fx.TRAIN_GRAD_NAMES:
        out[f"grad::{n}"] = ref_grads[n]
    out["grad_sqnorm_total"] = torch.tensor(sum(float((g.double() ** 2).sum()) for g in ref_grads.values()))
    out["grad_sqnorm_each"] = torch.tensor([float((g.double() ** 2).sum()) for g in ref_grads.values()])

    # one optimizer step exactly as Lightning does it: clip_grad_norm_(1.0) -> Adam.step -> EmaModel.update
    opt = pl_mod.configure_optimizers()["optimizer"]
    total = torch.nn.utils.clip_grad_norm_(pl_mod.model.parameters(), 1.0)
    opt.step()
    pl_mod.ema_model.update(pl_mod.model)
    coef, tot_o = orc.clip_scale(list(ref_grads.values()), 1.0)
    check("grad total norm", torch.tensor(tot_o), total, rtol=1e-5)
    new_p = dict(pl_mod.model.named_parameters()); new_e = dict(pl_mod.ema_model.ma_model.named_parameters())
    for n in fx.TRAIN_GRAD_NAMES:
        p1, m1, v1, e1 = orc.adam_ema_step(P[n], ref_grads[n], torch.zeros_like(P[n]), torch.zeros_like(P[n]),
                                           P[n], step=1, clip=coef)
        check(f"adam {n}", p1, new_p[n], rtol=1e-5, atol=1e-7)
        check(f"ema {n}", e1, new_e[n], rtol=1e-5, atol=1e-7)
        out[f"adam::{n}"] = new_p[n].detach(); out[f"ema::{n}"] = new_e[n].detach()
    out["clip_total_norm"] = total.detach()
    save("training_P.npz", seed=7, **out)


def make_cond_hparams(cfg: orc.UNetConfig, sampler: dict):
    """Mirror of configs/model/adm_edm_cond_h_res32.yaml."""
    hp = make_hparams(cfg, sampler)
    hp["name"] = "adm_edm_cond_h"
    hp.model.update(type="simple", var_type="fixedsmall", node_type=False)
    hp["diffusion"] = _wrap(dict(beta_schedule="linear", beta_start=0.0001, beta_end=0.02, num_diffusion_timesteps=1000))
    return hp


def golden_cond_edm():
    """Section 8(f2): single-task conditional EDM, models/ddim.py PlCondEdm (1608-1773) with the same DhariwalUNet
    (in 1 + cond 1 -> out 1): unmasked Heun sampler (deterministic and churned) and the training step."""
    cfg = fx.CFG_C
    P = orc.make_params(cfg, 13)
    out = {}
    for tag, churn in fx.COND_SAMPLER_CASES.items():
        sp = sampler_dict(S_churn=churn)
        m = PlCondEdm(make_cond_hparams(cfg, sp))
        assert [(n, tuple(p.shape)) for n, p in m.model.named_parameters()] == [(n, tuple(s)) for n, s in orc.param_shapes(cfg)]
        with torch.no_grad():
            for n, p in m.model.named_parameters():
                p.copy_(P[n])
            for n, p in m.ema_model.ma_model.named_parameters():
                p.copy_(P[n])
        h, u_noise, steps = fx.cond_sampler_inputs(tag)
        with torch.no_grad(), _Inject(steps):
            xs = m.sample_edm(h, u_noise, _wrap(sp), return_last=False)
        xo = orc.sample_edm_cond(P, cfg, h.permute(0, 3, 1, 2), orc.SamplerParams(S_churn=churn), u_noise.permute(0, 3, 1, 2),
                                 steps, return_last=False)
        check(f"PlCondEdm.sample_edm {tag}", xo, xs, rtol=1e-3, atol=1e-4)
        assert xs.dtype == torch.float64 and tuple(xs.shape) == (3, 19, 32, 32, 1)
        out[f"{tag}_xs_last"] = xs[:, -1:].contiguous()
        out[f"{tag}_xs_traj"] = xs[:, ::6].contiguous()
    # training step
    m = PlCondEdm(make_cond_hparams(cfg, sampler_dict()))
    with torch.no_grad():
        for n, p in m.model.named_parameters():
            p.copy_(P[n])
    h, u, noise, rnd_normal = fx.cond_training_inputs()
    st = fx.TRAIN_NORM_STATS
    m.normalizer_input.set_stats(torch.tensor(st[0]), torch.tensor(st[1]))
    m.normalizer_target.set_stats(torch.tensor(st[2]), torch.tensor(st[3]))
    with _Inject([noise], [rnd_normal]):
        loss = m.training_step((h, None, None, u), 0)
    loss.backward()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    hn, un = ((h - st[0]) / st[1]).permute(0, 3, 1, 2), ((u - st[2]) / st[3]).permute(0, 3, 1, 2)
    lo = orc.training_loss_cond(Pg, cfg, un, hn, noise, rnd_normal)
    lo.backward()
    check("PlCondEdm.training_step loss", lo, loss, rtol=1e-5, atol=1e-6)
    ref = {n: p.grad for n, p in m.model.named_parameters()}
    for n in ref:
        check(f"cond grad {n}", Pg[n].grad, ref[n], rtol=1e-3, rel_to_max=2e-6)
    out["loss"] = loss.detach()
    for n in fx.COND_GRAD_NAMES:
        out[f"grad::{n}"] = ref[n]
    out["grad_sqnorm_each"] = torch.tensor([float((g.double() ** 2).sum()) for g in ref.values()])
    sd = m.state_dict()
    out["betas"], out["logvar"] = sd["betas"], sd["logvar"]
    save("cond_edm.npz", seed=13, **out)


if __name__ == "__main__":
    golden_ops()
    golden_blocks()
    golden_unet()
    golden_sampler()
    golden_training()
    golden_cond_edm()
    print("all golden vectors written; oracle pinned against the reference on every case")
