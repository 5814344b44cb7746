"""Golden vectors for the data side of the hot path, made by RUNNING THE REFERENCE's dataset
classes (datamodules/h5_dataset.py) on a synthetic in-memory HDF5 tree.  h5py is not installed in the build container: the
reference module is imported with an in-memory ``h5py`` whose ``File`` serves nested dicts (the same kind of stand-in as the
``pytorch_lightning`` one in make_golden.py; no reference code is modified).  Stores outputs only: items and masks for every
dataset class in train / eval mode under a fixed torch seed, and the statistics both ways.

    python oracle/make_golden_data.py        # rewrites tests/golden/data.npz
"""
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get("MCEDM_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import fixtures as fx

_FILES = {}


class _Group(dict):
    attrs = {}


class _File:
    def __init__(self, path, mode="r"):
        self._g = _FILES[path]
        self.attrs = self._g.attrs

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def keys(self):
        return self._g.keys()

    def __getitem__(self, k):
        return self._g[k]

    def close(self):
        pass


h5 = types.ModuleType("h5py")
h5.File = _File
sys.modules["h5py"] = h5

from datamodules import h5_dataset as ref          # noqa: E402  (reference)

tree, attrs = fx.data_tree()
g = _Group({k: v for k, v in tree.items()})
g.attrs = attrs
_FILES["mem.h5"] = g
(im, istd, tm, tstd) = (attrs["inp_mean"], attrs["inp_std"], attrs["tar_mean"], attrs["tar_std"])
out = {}


def record(tag, item):
    inp, dx, dt, target = item[:4]
    out[f"{tag}/inp"], out[f"{tag}/target"] = inp.numpy(), target.numpy()
    out[f"{tag}/dx"], out[f"{tag}/dt"] = np.asarray(dx), np.asarray(dt)
    if len(item) > 4:
        m = item[4]
        if isinstance(m, dict):
            out[f"{tag}/mask_keys"] = np.array(list(m.keys()))
            for k, v in m.items():
                out[f"{tag}/mask_{k}"] = v.numpy()
        else:
            out[f"{tag}/mask"] = m.numpy()


for tag, (cls_name, kwargs) in fx.DATA_CASES.items():
    cls = getattr(ref, cls_name)
    ds = cls("mem.h5", input_mean=im, input_std=istd, target_mean=tm, target_std=tstd, **kwargs)
    assert len(ds) == len(tree)
    torch.manual_seed(fx.DATA_SEED)
    for idx in range(len(ds)):
        record(f"{tag}/{idx}", ds[idx])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "data.npz"), **out)
print(f"wrote tests/golden/data.npz: {len(out)} arrays from {len(fx.DATA_CASES)} dataset configurations")
