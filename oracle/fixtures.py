"""Deterministic, order-independent test inputs shared by make_golden.py and tests/ (TEST INFRASTRUCTURE).

Every tensor is drawn from its own NumPy stream seeded by crc32(tag), so the
golden files only need to hold the reference's OUTPUTS: tests rebuild the exact
inputs and parameters from the tags.
"""
import zlib

import numpy as np
import torch

from . import mcedm_oracle as orc


def _rng(tag: str):
    return np.random.default_rng(zlib.crc32(tag.encode()))


def randn(tag: str, *shape, dtype=np.float32) -> torch.Tensor:
    return torch.from_numpy(_rng(tag).standard_normal(shape).astype(dtype))


def uniform(tag: str, *shape) -> np.ndarray:
    return _rng(tag).random(size=shape, dtype=np.float64) * 2.0 - 1.0


def param(tag: str, name: str, shape) -> torch.Tensor:
    """A test parameter called ``name`` (fill rule of mcedm_oracle.fill_param)."""
    return torch.from_numpy(orc.fill_param(name, shape, uniform(f"{tag}/P/{name}", *shape)).astype(np.float32))


# block shape classes of SURVEY.md §3.4 (+ a 2-head block for ch=128)
BLOCK_CASES = {
    "plain": dict(cin=64, cout=64),
    "down": dict(cin=64, cout=64, down=True),
    "up": dict(cin=64, cout=64, up=True),
    "attn": dict(cin=64, cout=64, attn=True),
    "cat": dict(cin=128, cout=64),
    "catattn": dict(cin=128, cout=64, attn=True),
    "attn2h": dict(cin=128, cout=128, attn=True),
}


def block_spec(tag: str) -> orc.BlockSpec:
    c = BLOCK_CASES[tag]
    cin, cout = c["cin"], c["cout"]
    up, down, attn = c.get("up", False), c.get("down", False), c.get("attn", False)
    skip = (1 if cin != cout else 0) if (cin != cout or up or down) else -1
    return orc.BlockSpec("blk", cin, cout, up, down, attn, cout // 64 if attn else 0, skip)


def block_params(tag: str, emb: int = 64):
    b = block_spec(tag)
    return {n: param(f"blocks/{tag}", n, s) for n, s in orc.block_param_shapes(b, emb)}


def block_inputs(tag: str, n_emb: int, B: int = 2, H: int = 8, W: int = 8, emb: int = 64):
    b = block_spec(tag)
    return randn(f"blocks/{tag}/n{n_emb}/x", B, b.cin, H, W), randn(f"blocks/{tag}/n{n_emb}/emb", n_emb, emb)


# ---- per-op cases ----------------------------------------------------------
CONV_CASES = {"k3": dict(kernel=3), "k3up": dict(kernel=3, up=True), "k3down": dict(kernel=3, down=True),
              "k1": dict(kernel=1), "k0up": dict(kernel=0, up=True), "k0down": dict(kernel=0, down=True)}


def conv_channels(tag):
    return (8, 8) if CONV_CASES[tag]["kernel"] == 0 else (8, 16)


ATTN_CASES = {64: (2, (8, 8)), 256: (1, (16, 16))}        # tokens -> (batch, spatial), 2 heads x 64 ch
PE_LABELS = torch.tensor([-1.5536, -0.1733, 1.0955, 0.25], dtype=torch.float32)

# ---- whole-network cases -----------------------------------------------------
CFG_P = orc.UNetConfig()                                                     # adm_edm_mcedm_res32.yaml, ch=64
CFG_W = orc.UNetConfig(ch=128, ch_mult=(1, 1, 1, 1), attn_resolutions=(16,))  # BASELINE config 3 shape
UNET_LABELS = {"n1": torch.tensor([0.3]), "nB": torch.tensor([-1.2, -0.1, 0.6, 1.1])}
UNET_W_LABELS = torch.tensor([0.1, -0.7])
PRECOND_SIGMAS = (0.002, 0.5, 80.0)
PRECOND_SIGMA_B = torch.tensor([0.05, 0.4, 2.0, 30.0])

# ---- sampler cases: tag -> (S_churn, mask kind) --------------------------------
SAMPLER_CASES = {"det_u": (0.0, "u"), "churn_u": (15.0, "u"), "det_h": (0.0, "h_time")}


def task_mask(kind: str, B: int, H: int, W: int) -> torch.Tensor:
    """NCHW masks, 1 = missing.  'u': h observed, u missing (h5_dataset.py:245-247);
    'h': the converse; 'h_time': h missing everywhere, u missing for t >= H/2 (h5_dataset.py:370-373)."""
    m = torch.zeros(B, 2, H, W)
    if kind == "u":
        m[:, 1] = 1.0
    elif kind == "h":
        m[:, 0] = 1.0
    elif kind == "h_time":
        m[:, 0] = 1.0
        m[:, 1, H // 2:] = 1.0
    else:
        raise ValueError(kind)
    return m


def sampler_inputs(tag: str, B: int = 4, H: int = 32, W: int = 32, n_steps: int = 18):
    churn, kind = SAMPLER_CASES[tag]
    m = task_mask(kind, B, H, W)
    state = randn(f"sampler/{tag}/state", B, 2, H, W)
    cond = state * (1 - m) + randn(f"sampler/{tag}/cond_noise", B, 2, H, W) * m
    init = randn(f"sampler/{tag}/init", B, 2, H, W)
    # per-step noise is fp32-representable so an fp32 transport of it loses nothing
    steps = [randn(f"sampler/{tag}/step{i}", B, 2, H, W).double() for i in range(n_steps)]
    return cond, m, init, steps


# ---- training case -------------------------------------------------------------
TRAIN_NORM_STATS = (1.4, 0.2, 0.0, 0.5)        # input mean/std, target mean/std
TRAIN_GRAD_NAMES = ["enc.128x128_conv.weight", "out_conv.weight", "enc.64x64_down.norm1.weight",
                    "dec.32x32_block0.affine.weight", "dec.32x32_in0.qkv.weight", "dec.128x128_block1.skip.weight",
                    "map_layer0.weight", "dec.32x32_in0.proj.bias", "enc.32x32_block0.conv0.weight",
                    "dec.64x64_up.conv1.bias"]


def training_inputs(B: int = 4, T: int = 32, X: int = 32):
    """The reference's 5-tuple batch pieces (NHWC, h5_dataset.py:257-261) + injected noises."""
    h = randn("train/h", B, T, X, 1) * 0.2 + 1.4
    u = randn("train/u", B, T, X, 1) * 0.5
    mask = torch.zeros(B, T, X, 2)
    mask[0, ..., 1] = 1
    mask[1, ..., 0] = 1
    mask[2, ..., 1] = 1
    mask[3, : T // 2, :, 0] = 1
    mask[3, ..., 1] = 1
    cond_noise = randn("train/cond_noise", B, T, X, 2)
    noise = randn("train/noise", B, 2, T, X)
    rnd_normal = randn("train/rnd_normal", B, 1, 1, 1)
    return h, u, mask, cond_noise, noise, rnd_normal


def training_nchw(h, u, mask, cond_noise):
    """data_transform + get_cond_in + rearranges of training_step (mcedm.py:259-265,274)."""
    st = TRAIN_NORM_STATS
    x = torch.cat([(h - st[0]) / st[1], (u - st[2]) / st[3]], dim=-1)
    cond_in = orc.cond_input(x, mask, cond_noise).permute(0, 3, 1, 2).contiguous()
    return x.permute(0, 3, 1, 2).contiguous(), cond_in, mask.permute(0, 3, 1, 2).contiguous()


# ---- the optional widenings of the joint model's conditioning input (models/mcedm.py:25-34, 241-252) ---------------------
# tag -> (add_cond_mask, add_xt): the observation mask as extra channels (+ in_channels), the dx / dt fields (+ 2)
COND_IN_CASES = {"mask": (True, False), "xt": (False, True), "mask_xt": (True, True)}


def cond_in_cfg(tag: str) -> "orc.UNetConfig":
    add_mask, add_xt = COND_IN_CASES[tag]
    return orc.UNetConfig(cond_channels=2 + (2 if add_mask else 0) + (2 if add_xt else 0))


def cond_in_xt(B: int = 4, T: int = 32, X: int = 32):
    """dx, dt of the 5-tuple batch as fields 'b h w 1' (what get_cond_in concatenates when add_xt is set)."""
    return randn("condin/dx", B, T, X, 1) * 0.1 + 0.5, randn("condin/dt", B, T, X, 1) * 0.1 + 0.3


def cond_in_nchw(tag, h, u, mask, cond_noise, dx, dt):
    """data_transform + get_cond_in of models/mcedm.py:241-252 restated -> (x, cond_in, mask) in NCHW."""
    add_mask, add_xt = COND_IN_CASES[tag]
    st = TRAIN_NORM_STATS
    x = torch.cat([(h - st[0]) / st[1], (u - st[2]) / st[3]], dim=-1)
    cond = torch.cat([x * (1 - mask), 1.0 - mask], dim=-1) if add_mask else orc.cond_input(x, mask, cond_noise)
    if add_xt:
        cond = torch.cat([cond, dx, dt], dim=-1)
    return x.permute(0, 3, 1, 2).contiguous(), cond.permute(0, 3, 1, 2).contiguous(), mask.permute(0, 3, 1, 2).contiguous()


# PlCondEdm with hparams.model.node_type (models/ddim.py:36-38, 1105-1114): the conditioning gains a boundary-flag channel
CFG_NODE = orc.UNetConfig(in_channels=1, cond_channels=2, out_ch=1)


def node_cond(h: torch.Tensor) -> torch.Tensor:
    """get_cond_in of the node_type model restated: cat(h, node) 'b t x 2', node = 1 on the border of the (t, x) grid."""
    node = torch.zeros_like(h[..., :1])
    node[:, 0] = 1
    node[:, -1] = 1
    node[:, :, 0] = 1
    node[:, :, -1] = 1
    return torch.cat([h, node], dim=-1)


# ---- single-task conditional EDM (PlCondEdm, configs/model/adm_edm_cond_h_res32.yaml): h -> u, 1 + 1 -> 1 channels
CFG_C = orc.UNetConfig(in_channels=1, cond_channels=1, out_ch=1)
COND_SAMPLER_CASES = {"det": 0.0, "churn": 15.0}


def cond_sampler_inputs(tag: str, B: int = 3, H: int = 32, W: int = 32, n_steps: int = 18):
    """h (conditioning) and u_noise in the reference's NHWC layout + per-step noise (NCHW, fp32-representable)."""
    h = randn(f"cond/{tag}/h", B, H, W, 1)
    u_noise = randn(f"cond/{tag}/u_noise", B, H, W, 1)
    steps = [randn(f"cond/{tag}/step{i}", B, 1, H, W).double() for i in range(n_steps)]
    return h, u_noise, steps


COND_GRAD_NAMES = ["enc.128x128_conv.weight", "out_conv.weight", "dec.32x32_in0.qkv.weight", "map_layer1.weight",
                   "dec.64x64_block1.conv1.bias", "enc.32x32_down.norm0.weight"]


def cond_training_inputs(B: int = 3, T: int = 32, X: int = 32):
    h = randn("cond/train/h", B, T, X, 1) * 0.2 + 1.4
    u = randn("cond/train/u", B, T, X, 1) * 0.5
    noise = randn("cond/train/noise", B, 1, T, X)
    rnd_normal = randn("cond/train/rnd", B, 1, 1, 1)
    return h, u, noise, rnd_normal


# ---- PDE residuals: (b, t, x) sizes, Tn, x range -----------------------------------
PDE_SWE_CASES = {"per32": (3, 32, 32, 0.128, -0.5, 0.5), "dam128": (2, 128, 128, 1.28, -2.5, 2.5), "ragged": (2, 5, 37, 0.128, -0.5, 0.5)}
PDE_DARCY_CASES = {"d32": (3, 32), "d128": (2, 128), "d7": (2, 7), "d32fit": (2, 32)}


def pde_swe_inputs(name):
    """Physical-looking SWE states: h in ~[1, 2], u in ~[-0.5, 0.5]; gt = pred + small noise; normaliser scales."""
    B, T, X, _, _, _ = PDE_SWE_CASES[name]
    h = 1.5 + 0.25 * randn(f"pde/swe/{name}/h", B, T, X).clamp(-1.9, 1.9)
    u = 0.2 * randn(f"pde/swe/{name}/u", B, T, X)
    pred = torch.stack((h, u), dim=-1)
    gt = pred + 0.01 * randn(f"pde/swe/{name}/gt", B, T, X, 2)
    if name == "ragged":            # a dry cell and a NaN: the reference zeroes NaNs of the stepped state (pde_loss.py:219)
        pred[0, 1, 3, 0] = 0.0
        pred[1, 2, 5, 1] = float("nan")
    return pred, gt, torch.tensor(0.37), torch.tensor(0.21)


def pde_darcy_inputs(name):
    B, S = PDE_DARCY_CASES[name]
    if name == "d32fit":
        # a nearly exact solution (a = 1, u = -x^2/2 + small ripple: -div(a grad u) = 1 + O(3e-3)), so that the log-probability
        # form of the guidance (sigmoid(1e5 * residual^2), models/pde_loss.py:68-70) is in its non-saturated range
        xs = (torch.arange(S, dtype=torch.float64) + 0.5) / S
        X, Y = torch.meshgrid(xs, xs, indexing="ij")
        u = torch.stack([-(X ** 2) / 2 + a * 1e-4 * torch.sin(2 * np.pi * X) * torch.sin(2 * np.pi * Y) for a in (1.0, -0.7)][:B])
        return torch.stack((torch.ones(B, S, S, dtype=torch.float64), u), dim=-1).float()
    a = 1.0 + 0.5 * torch.from_numpy(uniform(f"pde/darcy/{name}/a", B, S, S).astype(np.float32))
    u = 0.1 * randn(f"pde/darcy/{name}/u", B, S, S)
    return torch.stack((a, u), dim=-1)


# ---- evaluation loops: test_step / validation_step of models/mcedm.py:283-441 ----------------
# tag -> system, n_samples, datamodule.down_factor / down_interp.  'darcy_n16' is BASELINE config 4's path (n_samples=16:
# the `n_samples < 15` branch of mcedm.py:438 drops the traj_* / gt_* entries) at plumbing size.
STEP_CASES = {
    "swe_n2": dict(system="swe_per", n_samples=2, down_factor=1, down_interp=False),
    "swe_n2_down": dict(system="swe_per", n_samples=2, down_factor=2, down_interp=True),
    "darcy_n16": dict(system="darcy", n_samples=16, down_factor=1, down_interp=False),
}
STEP_NORM_STATS = (1.4, 0.05, 0.0, 0.1)       # input mean/std, target mean/std: keeps un-normalised h (or a) positive
STEP_B, STEP_T, STEP_X = 2, 32, 32


def step_inputs(tag: str):
    """The reference's evaluation batch (NHWC, h5_dataset.py:244-261): h, u un-normalised, masks {"u", "h"} (1 = missing),
    and per task the injected noises: cond noise (get_cond_in, mcedm.py:247) and the sampler's initial noise (mcedm.py:576)."""
    c = STEP_CASES[tag]
    B, T, X, n = STEP_B, STEP_T, STEP_X, c["n_samples"]
    st = STEP_NORM_STATS
    h = randn(f"steps/{tag}/h", B, T, X, 1) * st[1] + st[0]
    u = randn(f"steps/{tag}/u", B, T, X, 1) * st[3] + st[2]
    masks = {"u": torch.zeros(B, T, X, 2), "h": torch.zeros(B, T, X, 2)}
    masks["u"][..., 1] = 1.0
    masks["h"][..., 0] = 1.0
    noises = {k: (randn(f"steps/{tag}/{k}/cond_noise", B, T, X, 2), randn(f"steps/{tag}/{k}/init", n * B, 2, T, X))
              for k in masks}
    return h, u, masks, noises


CFG_SAMPLER_W = 0.5       # classifier-free guidance weight of the 'cfg_u' sampler case (mcedm.py:453-458)


# ---- RePaint-style EDM sampling on the DDPM U-Net (PlDdim, configs/model/ddim_res32.yaml) --------
from . import ddpm_oracle as dorc  # noqa: E402

# the reference asserts input size == hparams.model.resolution (ddim_blocks.py:411): tests run the res-32 variant of
# ddim_res32.yaml (levels 32 / 16 / 8, attention at 8^2 = 64 tokens and in the middle block)
CFG_D = dorc.DdpmConfig(resolution=32, attn_resolutions=(8,))
DDPM_T = torch.tensor([937.0])                     # a late timestep: large sin/cos arguments in the embedding
DDPM_SIGMAS = (0.0123, 0.9, 78.5)
# tag -> (timesteps, n_repeat, S_churn, n_time_h, n_time_u)
REPAINT_CASES = {"det_r2": (6, 2, 0.0, 0, 16), "churn_r3": (5, 3, 15.0, 8, 0)}
REPAINT_B = 2


def repaint_inputs(tag: str):
    """h, u in the reference's 'b h w c' layout, initial noise (NCHW), per-step and per-repeat noise (fp32-representable)."""
    N, R, _, _, _ = REPAINT_CASES[tag]
    B, S = REPAINT_B, CFG_D.resolution
    h = randn(f"repaint/{tag}/h", B, S, S, 1)
    u = randn(f"repaint/{tag}/u", B, S, S, 1)
    init = randn(f"repaint/{tag}/init", B, 2, S, S)
    steps = [randn(f"repaint/{tag}/step{i}", B, 2, S, S).double() for i in range(N)]
    reps = [[randn(f"repaint/{tag}/rep{i}_{k}", B, 2, S, S).double() for k in range(R - 1)] for i in range(N)]
    return h, u, init, steps, reps


# ---- on-disk formats: a synthetic sample tree in the reference's HDF5 layout -------------------
DATA_SEED = 1234
# tag -> (reference dataset class, constructor keywords)
DATA_CASES = {
    "plain": ("HDF5Dataset", dict(return_abs_coords=False, return_grid=False)),
    "grid_flip": ("HDF5Dataset", dict(return_abs_coords=True, return_grid=True, flip_xy=True, norm_x=True, norm_t=True)),
    "theta_ic": ("HDF5Dataset", dict(return_abs_coords=True, return_grid=False, use_theta=True, use_tar_ic=True)),
    "down_interp": ("HDF5Dataset", dict(return_abs_coords=False, return_grid=False, down_factor=2, down_interp=True)),
    "down_small": ("HDF5Dataset", dict(return_abs_coords=True, return_grid=False, down_factor=2, down_interp=False)),
    "mask_train": ("HDF5MaskDataset", dict(return_abs_coords=True, return_grid=True, is_train=True)),
    "mask_eval": ("HDF5MaskDataset", dict(return_abs_coords=True, return_grid=True, is_train=False)),
    "time_train": ("HDF5TimeMaskDataset", dict(return_abs_coords=True, return_grid=True, is_train=True)),
    "time_eval": ("HDF5TimeMaskDataset", dict(return_abs_coords=True, return_grid=True, is_train=False, add_time_masks=True)),
    "sparse_train": ("HDF5SparseMaskDataset", dict(return_abs_coords=True, return_grid=True, is_train=True)),
    "sparse_eval": ("HDF5SparseMaskDataset", dict(return_abs_coords=True, return_grid=True, is_train=False, add_res_masks=True)),
}


def data_tree(n: int = 5, T: int = 16, X: int = 16):
    """{seed: {"data": {"input", "target"}, "grid": {"x", "t"}, "const": {...}}} + the file attributes, as numpy arrays
    (float64 on disk like the reference's generator writes them; t carries the extra final step some simulators store)."""
    tree = {}
    for i in range(n):
        seed = f"{1000 + 7 * i}"
        inp = 1.4 + 0.2 * randn(f"data/{seed}/inp", T, X, 1).double().numpy()
        tar = 0.5 * randn(f"data/{seed}/tar", T, X, 1).double().numpy()
        tree[seed] = {"data": {"input": inp, "target": tar},
                      "grid": {"x": np.linspace(-0.5, 0.5, X, endpoint=False) + 0.5 / X, "t": np.linspace(0.0, 0.128, T + 1)},
                      "const": {"g": np.array([1.0 + 0.1 * i]), "nu": np.array([0.01 * (i + 1)])}}
    allin = np.stack([v["data"]["input"] for v in tree.values()])
    alltar = np.stack([v["data"]["target"] for v in tree.values()])
    attrs = {"inp_mean": allin.mean(), "inp_std": allin.std(), "tar_mean": alltar.mean(), "tar_std": alltar.std(),
             "inp_min": allin.min(), "inp_max": allin.max(), "tar_min": alltar.min(), "tar_max": alltar.max()}
    return tree, attrs


def data_tree_flat():
    """The same tree in the flattened form mcedm_amd.data.NpzStore reads."""
    tree, attrs = data_tree()
    flat = {}
    for seed, grp in tree.items():
        for a, sub in grp.items():
            for b, arr in sub.items():
                flat[f"{seed}/{a}/{b}"] = np.asarray(arr)
    for k, v in attrs.items():
        flat[f"__attrs__/{k}"] = np.asarray(v)
    return flat


# ---- evaluation loops of models/ddim.py: PlDdim.test_step / validation_step (:294-533, what BASELINE config 5 is run
# through) and PlCondEdm.test_step / validation_step (:1154-1319, BASELINE config 4 read as the single-task model) -----------
# tag -> (system, n_samples, timesteps, n_repeat, S_churn, n_time_h, n_time_u)
EVAL_DDPM_CASES = {
    "n1_r2": ("swe_per", 1, 5, 2, 0.0, 0, 16),         # config 5 at plumbing size: h unknown, first half of u known
    "n5_r3": ("swe_per", 5, 4, 3, 15.0, 8, 0),
}
EVAL_DDPM_VAL = ("swe_per", 1, 4, 2, 0.0, 0, 16)
# input mean/std, target mean/std: a random-weight network drives the states to O(500), so the scales are kept small enough
# that the un-normalised water height stays positive (the residual divides by it)
EVAL_DDPM_STATS = (1.4, 0.002, 0.0, 0.002)
EVAL_B = 2
# tag -> (system, n_samples, guide_dx, norm stats)
EVAL_COND_CASES = {
    "swe_n2": ("swe_per", 2, False, STEP_NORM_STATS),
    "darcy_n16": ("darcy", 16, False, STEP_NORM_STATS),
    # guided Darcy sampling with every cell of the log-probability form saturated (u is scaled down until D u ~ 0, so
    # sigmoid(1e5 (Du - 1)^2) == 1 in fp32 and the gradient is exactly zero): the whole guided loop is comparable
    "darcy_guided_sat": ("darcy", 2, True, (1.4, 0.05, 0.0, 1e-8)),
}


def eval_ddpm_inputs(tag: str):
    """h, u un-normalised 'b t x 1' + the (n b)-batch noises of sample_edm (NCHW; per-step and per-repeat draws fp64)."""
    system, n, N, R, churn, nth, ntu = EVAL_DDPM_VAL if tag == "val" else EVAL_DDPM_CASES[tag]
    B, S, st = EVAL_B, CFG_D.resolution, EVAL_DDPM_STATS
    h = randn(f"evalddpm/{tag}/h", B, S, S, 1) * st[1] + st[0]
    u = randn(f"evalddpm/{tag}/u", B, S, S, 1) * st[3] + st[2]
    init = randn(f"evalddpm/{tag}/init", n * B, 2, S, S)
    steps = [randn(f"evalddpm/{tag}/step{i}", n * B, 2, S, S).double() for i in range(N)]
    reps = [[randn(f"evalddpm/{tag}/rep{i}_{k}", n * B, 2, S, S).double() for k in range(R - 1)] for i in range(N)]
    u_noise = randn(f"evalddpm/{tag}/u_noise", B, S, S, 1)          # validation_step hands noise in as u
    return h, u, init, steps, reps, u_noise


def eval_cond_inputs(tag: str):
    """h, u un-normalised 'b t x 1' and the sampler's initial noise '(n b) t x 1'."""
    if tag == "val":
        n, st = 1, STEP_NORM_STATS
    else:
        _, n, _, st = EVAL_COND_CASES[tag]
    B, T, X = EVAL_B, 32, 32
    h = randn(f"evalcond/{tag}/h", B, T, X, 1) * st[1] + st[0]
    u = randn(f"evalcond/{tag}/u", B, T, X, 1) * st[3] + st[2]
    return h, u, randn(f"evalcond/{tag}/init", n * B, T, X, 1)


# ---- DDIM sampler with RePaint loops (PlDdim.sample_with_repeat, models/ddim.py:808-913) on the res-32 DDPM U-Net -------------
# tag -> (timesteps, skip_type, eta, n_repeat, n_time_h, n_time_u)
# quad16 (round 4): 16 quad steps of 1000 -- int(np.linspace(0, sqrt(800), 16) ** 2) = 0, 3, 14, 31, 56 ...; the
# hi * i / (N - 1) evaluation order lands on 32 / 128 / 288 instead of 31 / 127 / 287
DDIM_CASES = {"uni_r2": (5, "uniform", 0.0, 2, 0, 16), "quad_eta_r3": (4, "quad", 0.01, 3, 8, 0), "quad16": (16, "quad", 0.0, 1, 0, 16)}
EVAL_DDIM = ("swe_per", 2, 4, "uniform", 0.0, 2, 0, 16)          # system, n_samples, then as above: PlDdim.test_step with type 'ddim'


def ddim_inputs(tag: str, B: int = REPAINT_B):
    """h, u 'b h w c', initial noise (NCHW) and the per-step UNIFORM draws of models/ddim.py:893 (used when eta != 0)."""
    N = (EVAL_DDIM[2] if tag == "eval" else DDIM_CASES[tag][0])
    S = CFG_D.resolution
    h = randn(f"ddim/{tag}/h", B, S, S, 1)
    u = randn(f"ddim/{tag}/u", B, S, S, 1)
    init = randn(f"ddim/{tag}/init", B, 2, S, S)
    eta = [torch.from_numpy(_rng(f"ddim/{tag}/eta{i}").random(size=(B, 2, S, S)).astype(np.float32)) for i in range(max(N, 8))]
    return h, u, init, eta


# ---- dx_cond (second clause): the network conditioned on the PDE-residual gradient ------------
def dxcond_net_inputs(B: int = 3, H: int = 32, W: int = 32):
    """x, cond, dx [B, 1, H, W] and sigma [B] for DhariwalUNet.forward(..., dx=dx) / get_denoised of the single-task model."""
    x = randn("dxcond/x", B, 1, H, W)
    cond = randn("dxcond/cond", B, 1, H, W)
    dx = randn("dxcond/dx", B, 1, H, W) * 0.7
    sigma = torch.tensor([0.3, 2.0, 11.0], dtype=torch.float32)[:B]
    return x, cond, dx, sigma


DXCOND_GRAD_NAMES = {
    "cat": ["enc.128x128_conv.weight", "enc.128x128_conv.bias", "out_conv.weight", "map_layer1.weight", "dec.32x32_in0.qkv.weight"],
    "enc": ["dx_enc.0.weight", "dx_enc.0.bias", "dx_enc.2.weight", "dx_enc.2.bias", "combine_enc.weight", "combine_enc.bias",
            "enc.128x128_conv.weight", "out_conv.weight"],
}
