"""Lightning checkpoints of the reference: ``configs/callbacks/callbacks_ddim.yaml:1-10`` writes
``checkpoints/last.ckpt`` (``ModelCheckpoint(save_last=True)``), ``run.py:68-72`` / ``eval_model.py:39,77`` resume / evaluate from
it.  A ``.ckpt`` is a ``torch.save``-d dict whose ``"state_dict"`` holds ``model.*``, ``ema_model.ma_model.*`` (incl. the
``resample_filter`` buffers) and the ``normalizer_*`` buffers; the drop-in modules keep those keys and shapes, so a reference
checkpoint loads with ``strict=True``.

The other entries of a Lightning checkpoint (``hyper_parameters`` -- an OmegaConf object --, optimizer states, callbacks)
may need packages this image does not have, and a pickle can name ANY importable callable: ``read_checkpoint`` therefore
unpickles through an allow-list (tensors, storages, dtypes, numpy arrays, plain containers) and turns every other global
into an inert placeholder.
"""
from __future__ import annotations

import io
import pickle
from typing import Any, Dict

import torch


class _Opaque:
    """Inert stand-in for an object whose class is not on the allow-list (e.g. omegaconf.DictConfig, a callback, or a
    hostile reducer): it swallows whatever the pickle stream does to it (construction, state, items) and does nothing."""

    args, kwargs, state = (), {}, None     # class-level defaults: pickle may build instances through __new__ alone

    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k

    @property
    def items(self):
        return self.__dict__.setdefault("_items", [])

    def __setstate__(self, state):
        self.state = state

    def __call__(self, *a, **k):          # some reducers call the reconstructed object
        return self

    def __setitem__(self, key, value):    # SETITEM(S) of dict subclasses
        self.items.append((key, value))

    def append(self, value):              # APPEND(S) of list subclasses
        self.items.append(value)

    def extend(self, values):
        self.items.extend(values)

    def add(self, value):                 # ADDITEMS of set subclasses
        self.items.append(value)


# Globals a tensor checkpoint legitimately needs.  Everything else -- importable or not -- is replaced by an inert
# placeholder, so unpickling never resolves (and REDUCE never calls) os.system, builtins.eval, subprocess.Popen, ...
_ALLOWED = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy", "ndarray"), ("numpy", "dtype"),
    ("_codecs", "encode"),                       # how pickle protocol 2 (torch.save's) spells a bytes literal
    ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "slice"), ("builtins", "range"), ("builtins", "complex"),
}


def _allowed(module: str, name: str) -> bool:
    if (module, name) in _ALLOWED:
        return True
    if module == "torch" and (name.endswith("Storage") or isinstance(getattr(torch, name, None), torch.dtype)):
        return True
    return module.startswith("numpy") and name in ("float32", "float64", "int32", "int64", "bool_", "uint8", "int8", "int16",
                                                    "float16", "complex64", "complex128")


class _RestrictedUnpickler(pickle.Unpickler):
    """Allow-list unpickler: tensors, storages, dtypes, numpy arrays and plain containers are rebuilt; every other global
    (hyper_parameters' OmegaConf objects, callbacks, optimizer classes, or anything hostile) becomes an ``_Opaque``."""

    def find_class(self, module, name):
        if _allowed(module, name):
            return super().find_class(module, name)
        return type(f"{module}.{name}".replace(".", "_"), (_Opaque,), {})


class _RestrictedPickle:
    """The ``pickle_module`` interface torch.load expects."""
    __name__ = "mcedm_restricted_pickle"
    Unpickler = _RestrictedUnpickler

    @staticmethod
    def load(f, **kw):
        return _RestrictedUnpickler(f, **kw).load()


def read_checkpoint(path_or_buffer) -> Dict[str, Any]:
    """Load a Lightning ``.ckpt`` (or any ``torch.save`` dict) to CPU.  ``weights_only=True`` first; a real Lightning file
    fails that (its ``hyper_parameters`` entry holds OmegaConf objects), and is then read by the allow-list unpickler above,
    which never imports or calls a global outside ``_ALLOWED``: loading an untrusted file cannot execute code."""
    try:
        return torch.load(path_or_buffer, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError:
        if hasattr(path_or_buffer, "seek"):
            path_or_buffer.seek(0)
        return torch.load(path_or_buffer, map_location="cpu", weights_only=False, pickle_module=_RestrictedPickle)


def state_dict_of(ckpt: Dict[str, Any]) -> Dict[str, torch.Tensor]:
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    return {k: v for k, v in sd.items() if torch.is_tensor(v)}


def load_reference_checkpoint(module: torch.nn.Module, path_or_buffer, strict: bool = True) -> Dict[str, Any]:
    """``trainer.fit/test(..., ckpt_path=<dir>/checkpoints/last.ckpt)`` without Lightning: restores ``model.*``,
    ``ema_model.ma_model.*`` and the normaliser buffers into a drop-in module (``PlMcedm`` / ``PlCondEdm`` / ``PlDdim``) and
    returns the bookkeeping entries (``epoch``, ``global_step``)."""
    ckpt = read_checkpoint(path_or_buffer)
    sd = state_dict_of(ckpt)
    for name in ("normalizer_input", "normalizer_target"):        # the reference stores scalar or per-channel statistics
        norm = getattr(module, name, None)
        if norm is not None:
            for buf in ("subtract", "divide"):
                key = f"{name}.{buf}"
                if key in sd and tuple(getattr(norm, buf).shape) != tuple(sd[key].shape):
                    setattr(norm, buf, torch.zeros_like(sd[key]))
    missing, unexpected = module.load_state_dict(sd, strict=False)
    if strict and (missing or unexpected):
        raise RuntimeError(f"checkpoint does not match the module: missing {list(missing)[:5]}, unexpected {list(unexpected)[:5]}")
    for net in (getattr(module, "model", None), getattr(getattr(module, "ema_model", None), "ma_model", None)):
        if net is not None and hasattr(net, "invalidate_packed"):
            net.invalidate_packed()
    return {k: ckpt.get(k) for k in ("epoch", "global_step", "pytorch-lightning_version") if isinstance(ckpt, dict) and k in ckpt}


def save_checkpoint(module: torch.nn.Module, path_or_buffer, epoch: int = 0, global_step: int = 0) -> None:
    """Write the subset of a Lightning checkpoint the reference's resume / eval paths read back (state_dict + counters)."""
    torch.save({"state_dict": {k: v.detach().cpu() for k, v in module.state_dict().items()}, "epoch": int(epoch),
                "global_step": int(global_step), "pytorch-lightning_version": "1.8.0"}, path_or_buffer)
