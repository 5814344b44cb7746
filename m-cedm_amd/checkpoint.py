"""Lightning checkpoints of the reference: ``configs/callbacks/callbacks_ddim.yaml:1-10`` writes
``checkpoints/last.ckpt`` (``ModelCheckpoint(save_last=True)``), ``run.py:68-72`` / ``eval_model.py:39,77`` resume / evaluate from
it.  A ``.ckpt`` is a ``torch.save``-d dict whose ``"state_dict"`` holds ``model.*``, ``ema_model.ma_model.*`` (incl. the
``resample_filter`` buffers) and the ``normalizer_*`` buffers; the drop-in modules keep those keys and shapes, so a reference
checkpoint loads with ``strict=True``.

The other entries of a Lightning checkpoint (``hyper_parameters`` -- an OmegaConf object --, optimizer states, callbacks)
may need packages this image does not have: ``read_checkpoint`` therefore unpickles with stand-ins for unknown classes
and returns only tensors, numbers and plain containers.
"""
from __future__ import annotations

import io
import pickle
from typing import Any, Dict

import torch


class _Opaque:
    """Stand-in for an object whose class is not importable here (e.g. omegaconf.DictConfig)."""

    def __init__(self, *a, **k):
        self.args, self.kwargs, self.state = a, k, None

    def __setstate__(self, state):
        self.state = state

    def __call__(self, *a, **k):          # some reducers call the reconstructed object
        return self


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            return type(f"{module}.{name}".replace(".", "_"), (_Opaque,), {})


class _TolerantPickle:
    """The ``pickle_module`` interface torch.load expects."""
    __name__ = "mcedm_tolerant_pickle"
    Unpickler = _TolerantUnpickler

    @staticmethod
    def load(f, **kw):
        return _TolerantUnpickler(f, **kw).load()


def read_checkpoint(path_or_buffer) -> Dict[str, Any]:
    """Load a Lightning ``.ckpt`` (or any ``torch.save`` dict) to CPU.  Unknown classes become opaque placeholders."""
    try:
        return torch.load(path_or_buffer, map_location="cpu", weights_only=True)
    except Exception:
        if hasattr(path_or_buffer, "seek"):
            path_or_buffer.seek(0)
        return torch.load(path_or_buffer, map_location="cpu", weights_only=False, pickle_module=_TolerantPickle)


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
