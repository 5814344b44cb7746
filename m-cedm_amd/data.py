"""On-disk formats either side of the hot path: the per-sample HDF5 layout the reference's data
modules read, the batch tuple + mask policies its LightningModules consume, and its normalisation statistics.

Reference: ``datamodules/h5_dataset.py`` (``HDF5Dataset`` :14-186, ``HDF5MaskDataset`` :189-261 with the train-time task
mask :232-255, ``HDF5TimeMaskDataset`` :264-393, ``HDF5SparseMaskDataset`` :396-548) and ``datamodules/pl_datamodule.py``
(statistics from file attributes :81-90 or from the samples :91-121, ``get_norm_stats`` :205-233).

File layout (one group per sample, named by its seed):
    <seed>/data/input   [T, X, c_in]      <seed>/data/target  [T, X, c_out]
    <seed>/grid/x       [X]               <seed>/grid/t       [T] (some simulators store T + 1)
    <seed>/const/<name> [1]               (optional, ``use_theta``)
    file attributes     inp_mean, inp_std, tar_mean, tar_std, inp_min, inp_max, tar_min, tar_max

A *store* is anything with that mapping interface: an ``h5py.File`` when h5py is installed, or ``NpzStore`` -- the same
tree flattened into one ``.npz`` (keys ``"<seed>/data/input"``, attributes under ``"__attrs__/<name>"``), which needs no
HDF5 library on the GPU box; ``store_to_npz`` converts.  The dataset classes keep the reference's names, constructor
arguments, return tuples and -- because training masks are drawn from torch's global generator -- the reference's exact
order of random draws, so a seeded run sees the same masks.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import DataLoader, Dataset

from .mcedm import DotDict

_STAT_ATTRS = ("inp_mean", "inp_std", "tar_mean", "tar_std", "inp_min", "inp_max", "tar_min", "tar_max")


class _Node:
    """A group of an NpzStore: children addressed by name, leaves are numpy arrays (``[:]`` works on both)."""

    def __init__(self, arrays: Dict[str, np.ndarray], prefix: str):
        self._a, self._p = arrays, prefix

    def __getitem__(self, name):
        key = f"{self._p}{name}"
        if key in self._a:
            return self._a[key]
        if any(k.startswith(key + "/") for k in self._a):
            return _Node(self._a, key + "/")
        raise KeyError(key)

    def keys(self):
        n = len(self._p)
        return sorted({k[n:].split("/", 1)[0] for k in self._a if k.startswith(self._p) and not k.startswith("__attrs__/")})

    def __contains__(self, name):
        return name in self.keys()


class NpzStore(_Node):
    """The HDF5 tree flattened into one .npz (read fully into memory; the SWE / Darcy sets are a few hundred MB)."""

    def __init__(self, path_or_arrays):
        arrays = dict(np.load(path_or_arrays)) if isinstance(path_or_arrays, (str, os.PathLike)) else dict(path_or_arrays)
        super().__init__(arrays, "")
        self.attrs = {k[len("__attrs__/"):]: v for k, v in arrays.items() if k.startswith("__attrs__/")}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def close(self):
        pass


def open_store(path):
    """``h5py.File(path, 'r')`` for .h5 / .hdf5 files (h5py must be importable), ``NpzStore`` for .npz."""
    if isinstance(path, (NpzStore,)) or hasattr(path, "keys") and hasattr(path, "attrs"):
        return path
    if str(path).endswith(".npz"):
        return NpzStore(path)
    try:
        import h5py
    except ImportError as e:
        raise RuntimeError(f"{path}: reading HDF5 needs h5py, which is not installed here; convert the file once with "
                           "mcedm_amd.data.store_to_npz on a machine that has it and pass the .npz") from e
    return h5py.File(path, "r")


def store_to_npz(store, out_path: str) -> None:
    """Flatten any store (e.g. an open h5py.File) into the .npz form NpzStore reads."""
    flat = {}

    def walk(node, prefix):
        for k in node.keys():
            child = node[k]
            if hasattr(child, "keys"):
                walk(child, f"{prefix}{k}/")
            else:
                flat[f"{prefix}{k}"] = np.asarray(child[:] if hasattr(child, "__getitem__") else child)
    walk(store, "")
    for k, v in dict(store.attrs).items():
        flat[f"__attrs__/{k}"] = np.asarray(v)
    np.savez_compressed(out_path, **flat)


def _bilinear(field: torch.Tensor, scale: float) -> torch.Tensor:
    """[T, X, C] -> interpolate over the two grid axes exactly as the reference routes it through F.interpolate
    (h5_dataset.py:146-148): permute to [1, C, X, T], bilinear, align_corners False."""
    return F.interpolate(field.permute(2, 1, 0).unsqueeze(0), scale_factor=scale, mode="bilinear",
                         align_corners=False).squeeze(0).permute(2, 1, 0)


class HDF5Dataset(Dataset):
    """datamodules/h5_dataset.py:14-186: one item = (input, dx | x | x_grid, dt | t | t_grid, target)."""

    def __init__(self, datapath, return_abs_coords: bool, return_grid: bool, input_mean, input_std, target_mean, target_std,
                 norm_x: bool = False, norm_t: bool = False, norm_input: bool = True, norm_target: bool = True,
                 flip_xy: bool = False, use_theta: bool = False, use_tar_ic: bool = False, dtype=torch.float32,
                 down_factor: int = 1, down_interp: bool = True):
        super().__init__()
        self.dtype, self.datapath = dtype, datapath
        self.return_abs_coords, self.return_grid = return_abs_coords, return_grid
        as_t = lambda v: v if torch.is_tensor(v) else torch.tensor(v, dtype=dtype)     # noqa: E731
        self.input_mean, self.input_std = as_t(input_mean), as_t(input_std)
        self.target_mean, self.target_std = as_t(target_mean), as_t(target_std)
        self.norm_x, self.norm_t, self.norm_input, self.norm_target = norm_x, norm_t, norm_input, norm_target
        self.flip_xy, self.use_theta, self.use_tar_ic = flip_xy, use_theta, use_tar_ic
        self.down_factor, self.down_interp = down_factor, down_interp
        self._store = None
        with self._open() as f:
            self.data_list = np.array(sorted(f.keys()))

    def _open(self):
        # .npz stores are parsed once and kept; HDF5 files are re-opened per access like the reference (worker safety)
        if str(self.datapath).endswith(".npz") or not isinstance(self.datapath, (str, os.PathLike)):
            if self._store is None:
                self._store = open_store(self.datapath)
            return self._store
        return open_store(self.datapath)

    def __len__(self):
        return len(self.data_list)

    def __getitem__(self, idx: int):
        with self._open() as f:
            grp = f[self.data_list[idx]]
            inp = torch.tensor(np.asarray(grp["data"]["input"][:]), dtype=self.dtype)           # [T, X, c]
            target = torch.tensor(np.asarray(grp["data"]["target"][:]), dtype=self.dtype)
            if self.norm_input:
                inp = (inp - self.input_mean) / self.input_std
            if self.norm_target:
                target = (target - self.target_mean) / self.target_std
            if self.flip_xy:
                inp, target = target, inp.clone()
            if self.use_theta:
                names = list(grp["const"].keys())
                theta = torch.ones(inp.shape[0], inp.shape[1], len(names), dtype=self.dtype)
                for i, c in enumerate(names):
                    theta[..., i] = torch.tensor(np.asarray(grp["const"][c])[0], dtype=self.dtype)
                inp = torch.cat([inp, theta], dim=-1)
            if self.use_tar_ic:
                inp = torch.cat([inp, target[0:1].repeat(inp.shape[0], 1, 1)], dim=-1)
            x = torch.tensor(np.asarray(grp["grid"]["x"][:]), dtype=self.dtype)
            t = torch.tensor(np.asarray(grp["grid"]["t"][:]), dtype=self.dtype)
        if len(t) > len(inp):
            t = t[:-1]
        if self.norm_x:
            x = (x - x.min()) / (x.max() - x.min())
        if self.norm_t:
            t = (t - t.min()) / (t.max() - t.min())
        if self.down_factor > 1:
            each = 2 ** (self.down_factor - 1)
            if self.down_interp:             # sub-sample, then interpolate back to the full grid (:143-154)
                inp = _bilinear(inp[::each, ::each], each)
                target = _bilinear(target[::each, ::each], each)
            else:                            # feed the smaller resolution (:155-167)
                inp = _bilinear(inp, 1 / each)
                lin = lambda v: F.interpolate(v[None, None], scale_factor=1 / each, mode="linear", align_corners=False)[0, 0]  # noqa: E731
                x, t = lin(x), lin(t)
                target = _bilinear(target, 1 / each)
        if self.return_abs_coords:
            if self.return_grid:
                t_grid, x_grid = torch.meshgrid(t, x, indexing="ij")
                return inp, t_grid.unsqueeze(-1), x_grid.unsqueeze(-1), target
            return inp, x, t, target
        return inp, torch.diff(x)[0], torch.diff(t)[0], target


def _task_masks(inp, target):
    """The two evaluation tasks (1 = missing): 'u' = h observed / u missing, 'h' = the converse (h5_dataset.py:244-253)."""
    zi, oi, zt, ot = torch.zeros_like(inp), torch.ones_like(inp), torch.zeros_like(target), torch.ones_like(target)
    return {"u": torch.cat([zi, ot], dim=-1), "h": torch.cat([oi, zt], dim=-1)}


class HDF5MaskDataset(HDF5Dataset):
    """h5_dataset.py:189-261: adds the task mask; training draws ONE uniform number per item (:235)."""

    def __init__(self, *args, is_train: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.is_train = is_train

    def sample_mask(self, inp, target):
        if not self.is_train:
            return _task_masks(inp, target)
        if torch.rand(1) > 0.5:
            return torch.cat([torch.zeros_like(inp), torch.ones_like(target)], dim=-1)
        return torch.cat([torch.ones_like(inp), torch.zeros_like(target)], dim=-1)

    def __getitem__(self, idx: int):
        inp, dx, dt, target = super().__getitem__(idx)
        return inp, dx, dt, target, self.sample_mask(inp, target)


def _variable_mask(inp, target, p_target: float, p_input: float):
    """Which variable is hidden entirely: one uniform draw (h5_dataset.py:310-324 / 442-456)."""
    var = torch.rand(1)
    hide_inp, hide_tar = bool(p_target < var <= p_input), bool(var <= p_target)
    return torch.cat([torch.full_like(inp, hide_inp, dtype=torch.bool), torch.full_like(target, hide_tar, dtype=torch.bool)], dim=-1)


class HDF5TimeMaskDataset(HDF5MaskDataset):
    """h5_dataset.py:264-393: training hides a variable and everything after a random time per variable; evaluation
    optionally adds the half-horizon forecasting tasks (:346-391)."""

    def __init__(self, *args, add_time_masks: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.add_time_masks = add_time_masks

    def get_train_mask(self, inp, target):
        c = inp.shape[-1]
        mask_var = _variable_mask(inp, target, 0.4, 0.8)
        res = inp.shape[0]
        t1 = res // 2 + torch.randint(res // 2 + 1, (1,))
        t2 = res // 2 + torch.randint(res // 2 + 1, (1,))
        late = torch.ones_like(mask_var, dtype=torch.bool)
        late[:t1, :, :c] = False
        late[:t2, :, c:] = False
        return (mask_var | late).float()

    def sample_mask(self, inp, target):
        if self.is_train:
            return self.get_train_mask(inp, target)
        masks = _task_masks(inp, target)
        if self.add_time_masks:
            half = int(0.5 * inp.shape[0])

            def after(n, base_inp, base_tar, cut_inp, cut_tar):
                mi, mt = torch.full_like(inp, base_inp), torch.full_like(target, base_tar)
                if cut_inp:
                    mi[n:] = 1
                if cut_tar:
                    mt[n:] = 1
                return torch.cat([mi, mt], dim=-1)
            masks = {"hu": after(half, 0.0, 0.0, True, True), "u": after(half, 0.0, 1.0, True, False),
                     "h": after(half, 1.0, 0.0, False, True)}
        return masks


class HDF5SparseMaskDataset(HDF5MaskDataset):
    """h5_dataset.py:396-548: training observes a variable on a random sub-lattice up to a random time; evaluation
    optionally observes every 4th point only (:510-546)."""

    def __init__(self, *args, add_res_masks: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.add_res_masks = add_res_masks

    def get_train_mask(self, inp, target):
        c = inp.shape[-1]
        mask_var = _variable_mask(inp, target, 0.33, 0.66)
        r1 = torch.randint(3, ()) + 1
        r2 = torch.randint(3, ()) + 1
        e1, e2 = 2 ** (r1 - 1), 2 ** (r2 - 1)
        res = inp.shape[0]
        t1 = res // 2 + r1 * torch.randint(res // 2 ** (r1 - 1) // 2 + 1, (1,))
        t2 = res // 2 + r2 * torch.randint(res // 2 ** (r2 - 1) // 2 + 1, (1,))
        hidden = torch.ones_like(mask_var, dtype=torch.bool)
        hidden[:t1:e1, ::e1, :c] = False
        hidden[:t2:e2, ::e2, c:] = False
        return (mask_var | hidden).float()

    def sample_mask(self, inp, target):
        if self.is_train:
            return self.get_train_mask(inp, target)
        if not self.add_res_masks:
            return _task_masks(inp, target)
        mi, mt = torch.ones_like(inp), torch.ones_like(target)
        mi[::4, ::4] = 0
        mu = torch.cat([mi, torch.ones_like(target)], dim=-1)
        mt[::4, ::4] = 0
        mh = torch.cat([torch.ones_like(inp), mt], dim=-1)
        return {"u": mu, "h": mh}


def norm_stats_from_store(store, const_norm_stats: bool = True):
    """pl_datamodule.py:76-121: (mean_std, min_max) as fp32 tensors, from the file attributes or from the samples."""
    f = open_store(store)
    if const_norm_stats:
        vals = [torch.tensor(np.asarray(f.attrs[k]), dtype=torch.float32) for k in _STAT_ATTRS]
        return vals[:4], vals[4:]
    inputs, targets = [], []
    for key in f.keys():
        inputs.append(np.asarray(f[key]["data"]["input"][:]))
        targets.append(np.asarray(f[key]["data"]["target"][:]))
    inputs = torch.tensor(np.stack(inputs, axis=0), dtype=torch.float32).squeeze(dim=-1)
    targets = torch.tensor(np.stack(targets, axis=0), dtype=torch.float32).squeeze(dim=-1)
    mean_std = [inputs.mean(dim=0), inputs.std(dim=0), targets.mean(dim=0), targets.std(dim=0)]
    min_max = [inputs.min(dim=0)[0], inputs.max(dim=0)[0], targets.min(dim=0)[0], targets.max(dim=0)[0]]
    return mean_std, min_max


class HDF5MaskDatamodule:
    """datamodules/pl_datamodule.py:221-317 without the Lightning base class: the three splits, their loaders and
    ``get_norm_stats()`` (what ``PlMcedm.setup('fit')`` reads; ``down_factor`` / ``down_interp`` are what ``test_step`` reads).
    ``dataset_cls`` / ``dataset_kwargs`` select the time-mask or sparse-mask variants."""

    eps = 1e-8

    def __init__(self, train_path, val_path, test_path, return_abs_coords=False, return_grid=False, norm_x=False, norm_t=False,
                 norm_input=True, norm_target=True, flip_xy=False, const_norm_stats=True, use_theta=False, use_tar_ic=False,
                 num_workers=0, batch_size=32, test_batch_size=None, down_factor=1, down_interp=True,
                 dataset_cls=HDF5MaskDataset, dataset_kwargs: Optional[dict] = None):
        self.paths = dict(train=train_path, val=val_path, test=test_path)
        self.common = dict(return_abs_coords=return_abs_coords, return_grid=return_grid, norm_x=norm_x, norm_t=norm_t,
                           norm_input=norm_input, norm_target=norm_target, flip_xy=flip_xy, use_theta=use_theta,
                           use_tar_ic=use_tar_ic, dtype=torch.float32)
        self.norm_input, self.norm_target, self.flip_xy = norm_input, norm_target, flip_xy
        self.batch_size, self.test_batch_size = batch_size, (test_batch_size or batch_size)
        self.num_workers, self.down_factor, self.down_interp = num_workers, down_factor, down_interp
        self.dataset_cls, self.dataset_kwargs = dataset_cls, dict(dataset_kwargs or {})
        (im, istd, tm, tstd), (imin, imax, tmin, tmax) = norm_stats_from_store(train_path, const_norm_stats)
        self.input_mean, self.input_std, self.target_mean, self.target_std = im, istd + self.eps, tm, tstd + self.eps
        self.input_min, self.input_min_max = imin, imax - imin + self.eps
        self.target_min, self.target_min_max = tmin, tmax - tmin + self.eps

    def _make(self, split, **extra):
        return self.dataset_cls(self.paths[split], input_mean=self.input_mean, input_std=self.input_std,
                                target_mean=self.target_mean, target_std=self.target_std, **self.common, **self.dataset_kwargs,
                                **extra)

    def setup(self, stage=None):
        self.train_dataset = self._make("train", is_train=True)
        self.val_dataset = self._make("val", down_factor=self.down_factor, down_interp=self.down_interp)
        self.test_dataset = self._make("test", down_factor=self.down_factor, down_interp=self.down_interp)

    def train_dataloader(self):
        return DataLoader(self.train_dataset, batch_size=self.batch_size, num_workers=self.num_workers, shuffle=True, pin_memory=True)

    def val_dataloader(self):
        return DataLoader(self.val_dataset, batch_size=self.batch_size, num_workers=self.num_workers, shuffle=False, pin_memory=True)

    def test_dataloader(self):
        return DataLoader(self.test_dataset, batch_size=self.test_batch_size, num_workers=self.num_workers, shuffle=False, pin_memory=True)

    def get_norm_stats(self):
        a, b = ("target", "input") if self.flip_xy else ("input", "target")
        return DotDict({f"norm_{b}": self.norm_target, f"{b}_mean": self.target_mean, f"{b}_std": self.target_std,
                        f"{b}_min": self.target_min, f"{b}_min_max": self.target_min_max,
                        f"norm_{a}": self.norm_input, f"{a}_mean": self.input_mean, f"{a}_std": self.input_std,
                        f"{a}_min": self.input_min, f"{a}_min_max": self.input_min_max})
