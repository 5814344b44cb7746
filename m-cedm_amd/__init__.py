"""mcedm_amd -- MI355X-native (gfx950) EDM training / Heun-sampling hot path of katehai/m-cedm.

``lib``        ctypes binding of libmcedm_hip.so (HIP kernels + C ABI, see include/mcedm_hip.h)
``adm_blocks`` drop-in ``DhariwalUNet`` (same constructor, state_dict keys and forward signature)
``mcedm``      drop-in ``PlMcedm`` (training_step / model_precond / get_denoised / sample_edm)
"""
from . import lib  # noqa: F401

__all__ = ["lib"]
