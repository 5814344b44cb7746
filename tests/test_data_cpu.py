"""CPU: the data side of the hot path.  mcedm_amd.data's dataset classes against the outputs of the
reference's own classes (tests/golden/data.npz, oracle/make_golden_data.py) under the same torch seed -- items, grids,
down-sampling policies and every train / evaluation mask policy -- plus the statistics, the .npz store round trip, the
loader -> LightningModule batch contract, and reference-layout checkpoints."""
import io

import numpy as np
import pytest
import torch

import mcedm_amd  # noqa: F401
from mcedm_amd import checkpoint as ck
from mcedm_amd import data as D
from oracle import fixtures as fx


@pytest.fixture(scope="module")
def store():
    return D.NpzStore(fx.data_tree_flat())


@pytest.mark.parametrize("tag", list(fx.DATA_CASES))
def test_dataset_items_and_masks_match_reference(golden, store, tag):
    g = golden("data.npz")
    cls_name, kwargs = fx.DATA_CASES[tag]
    (im, istd, tm, tstd), _ = D.norm_stats_from_store(store, const_norm_stats=True)
    ds = getattr(D, cls_name)(store, input_mean=im, input_std=istd, target_mean=tm, target_std=tstd, **kwargs)
    assert len(ds) == 5
    torch.manual_seed(fx.DATA_SEED)                       # the mask policies draw from torch's global generator
    for idx in range(len(ds)):
        item = ds[idx]
        pre = f"{tag}/{idx}"
        for name, val in zip(("inp", "dx", "dt", "target"), item[:4]):
            ref = g[f"{pre}/{name}"]
            got = np.asarray(val)
            assert got.shape == ref.shape and got.dtype == ref.dtype, (pre, name, got.shape, ref.shape)
            np.testing.assert_array_equal(got, ref, err_msg=f"{pre}/{name}")
        if len(item) > 4:
            m = item[4]
            if isinstance(m, dict):
                assert list(m.keys()) == list(g[f"{pre}/mask_keys"])
                for k, v in m.items():
                    np.testing.assert_array_equal(v.numpy(), g[f"{pre}/mask_{k}"], err_msg=f"{pre}/mask_{k}")
            else:
                np.testing.assert_array_equal(m.numpy(), g[f"{pre}/mask"], err_msg=f"{pre}/mask")


def test_statistics_both_ways_and_npz_round_trip(store, tmp_path):
    tree, attrs = fx.data_tree()
    (im, istd, tm, tstd), (imin, imax, tmin, tmax) = D.norm_stats_from_store(store, True)
    assert float(im) == np.float32(attrs["inp_mean"]) and float(tmax) == np.float32(attrs["tar_max"])
    (pm, ps, qm, qs), _ = D.norm_stats_from_store(store, False)       # per-location statistics over the samples
    allin = torch.tensor(np.stack([v["data"]["input"] for v in tree.values()]), dtype=torch.float32).squeeze(-1)
    assert tuple(pm.shape) == (16, 16) and torch.equal(pm, allin.mean(0)) and torch.equal(ps, allin.std(0))
    path = tmp_path / "set.npz"
    D.store_to_npz(store, str(path))
    again = D.NpzStore(str(path))
    assert sorted(again.keys()) == sorted(store.keys())
    np.testing.assert_array_equal(again["1000"]["data"]["input"][:], store["1000"]["data"]["input"][:])
    assert float(again.attrs["inp_std"]) == float(store.attrs["inp_std"])
    with pytest.raises(RuntimeError, match="h5py"):
        D.open_store(str(tmp_path / "missing.h5"))


def test_datamodule_batches_feed_the_lightning_module_contract(store):
    """loader -> (h, t_grid, x_grid, u, mask) exactly as PlMcedm.training_step / test_step unpack it (mcedm.py:255, 344)."""
    dm = D.HDF5MaskDatamodule(store, store, store, return_abs_coords=True, return_grid=True, batch_size=2, down_factor=2,
                              dataset_cls=D.HDF5TimeMaskDataset, dataset_kwargs=dict(add_time_masks=True))
    dm.setup()
    st = dm.get_norm_stats()
    assert float(st["input_std"]) > 0 and set(st) >= {"input_mean", "input_std", "target_mean", "target_std", "input_min_max"}
    h, tg, xg, u, mask = next(iter(dm.train_dataloader()))
    assert tuple(h.shape) == (2, 16, 16, 1) and tuple(u.shape) == (2, 16, 16, 1) and tuple(mask.shape) == (2, 16, 16, 2)
    assert tuple(tg.shape) == (2, 16, 16, 1) and set(np.unique(mask.numpy())) <= {0.0, 1.0}
    h, tg, xg, u, masks = next(iter(dm.test_dataloader()))
    assert list(masks) == ["hu", "u", "h"] and tuple(masks["u"].shape) == (2, 16, 16, 2)
    assert dm.down_factor == 2 and dm.down_interp is True          # read by test_step (mcedm.py:347)
    flipped = D.HDF5MaskDatamodule(store, store, store, flip_xy=True).get_norm_stats()
    assert float(flipped["target_mean"]) == float(st["input_mean"])


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _wrap(d):
    return AttrDict({k: _wrap(v) for k, v in d.items()}) if isinstance(d, dict) else d


def test_reference_layout_checkpoint_round_trip():
    """A Lightning-style .ckpt with the reference's state_dict keys (model.*, ema_model.ma_model.*, resample_filter buffers,
    normaliser buffers) loads into the drop-in module with strict=True."""
    from mcedm_amd.mcedm import PlMcedm
    from oracle import mcedm_oracle as orc
    from tests.test_hip_module import hparams
    cfg = fx.CFG_P
    P = orc.make_params(cfg, 3)
    sd = {}
    for n, v in P.items():
        sd[f"model.{n}"] = v
        sd[f"ema_model.ma_model.{n}"] = v * 0.5
    for net in ("model.", "ema_model.ma_model."):
        for key in ("enc.64x64_down", "enc.32x32_down", "dec.64x64_up", "dec.128x128_up"):
            for conv in ("conv0", "skip"):
                sd[f"{net}{key}.{conv}.resample_filter"] = torch.full((1, 1, 2, 2), 0.25)
    sd.update({"normalizer_input.subtract": torch.tensor(1.4), "normalizer_input.divide": torch.tensor(0.2),
               "normalizer_target.subtract": torch.tensor(0.0), "normalizer_target.divide": torch.tensor(0.5)})
    assert len(sd) == 412

    buf = io.BytesIO()
    torch.save({"state_dict": sd, "epoch": 41, "global_step": 1300, "pytorch-lightning_version": "1.8.0"}, buf)
    buf.seek(0)
    m = PlMcedm(hparams(cfg))
    meta = ck.load_reference_checkpoint(m, buf, strict=True)
    assert meta["epoch"] == 41 and meta["global_step"] == 1300
    got = m.state_dict()
    assert sorted(got) == sorted(sd)
    for k in ("model.dec.32x32_in0.qkv.weight", "ema_model.ma_model.out_conv.bias", "normalizer_input.divide"):
        assert torch.equal(got[k], sd[k]), k
    out = io.BytesIO()
    ck.save_checkpoint(m, out, epoch=42)
    out.seek(0)
    again = ck.read_checkpoint(out)
    assert again["epoch"] == 42 and sorted(again["state_dict"]) == sorted(sd)
    # a checkpoint that does not match is refused under strict loading
    bad = dict(sd)
    bad.pop("model.out_conv.bias")
    b2 = io.BytesIO()
    torch.save({"state_dict": bad}, b2)
    b2.seek(0)
    with pytest.raises(RuntimeError, match="does not match"):
        ck.load_reference_checkpoint(PlMcedm(hparams(cfg)), b2, strict=True)


def test_tolerant_unpickler_skips_unknown_classes(tmp_path):
    import sys
    import types
    mod = types.ModuleType("vanishing_pkg")

    class Cfg:
        def __init__(self):
            self.lr = 2e-4
    Cfg.__module__, Cfg.__qualname__ = "vanishing_pkg", "Cfg"
    mod.Cfg = Cfg
    sys.modules["vanishing_pkg"] = mod
    path = tmp_path / "last.ckpt"
    torch.save({"state_dict": {"w": torch.ones(3)}, "hyper_parameters": Cfg(), "epoch": 7}, str(path))
    del sys.modules["vanishing_pkg"]                         # the class can no longer be imported
    c = ck.read_checkpoint(str(path))
    assert c["epoch"] == 7 and torch.equal(c["state_dict"]["w"], torch.ones(3)) and c["hyper_parameters"] is not None


class _Boom:
    """A pickle whose REDUCE would call an importable function with side effects."""

    def __init__(self, path):
        self.path = path

    def __reduce__(self):
        import os
        return (os.mkdir, (self.path,))


def test_checkpoint_loading_never_calls_importable_globals(tmp_path):
    """A hostile .ckpt (hyper_parameters reducing to os.mkdir(...)) must load its tensors WITHOUT executing the call: the
    fallback unpickler is an allow-list, importable globals included (round-2 advisor finding)."""
    marker = tmp_path / "pwned"
    path = tmp_path / "evil.ckpt"
    torch.save({"state_dict": {"w": torch.arange(4.0)}, "hyper_parameters": _Boom(str(marker)), "epoch": 3}, str(path))
    c = ck.read_checkpoint(str(path))
    assert not marker.exists(), "unpickling executed os.mkdir"
    assert c["epoch"] == 3 and torch.equal(c["state_dict"]["w"], torch.arange(4.0))
    assert isinstance(c["hyper_parameters"], ck._Opaque)
    # numpy payloads (Lightning stores some counters as numpy scalars) still load
    import numpy as np
    p2 = tmp_path / "np.ckpt"
    torch.save({"state_dict": {"w": torch.ones(2)}, "best": np.float64(0.25), "arr": np.arange(3)}, str(p2))
    c2 = ck.read_checkpoint(str(p2))
    assert float(c2["best"]) == 0.25 and list(c2["arr"]) == [0, 1, 2]
