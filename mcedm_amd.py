"""Import shim: the package directory is named ``m-cedm_amd`` (not a Python identifier), so this
module loads it under the importable name ``mcedm_amd``:  ``import mcedm_amd; mcedm_amd.lib.load()``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "m-cedm_amd")
_spec = importlib.util.spec_from_file_location("mcedm_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mcedm_amd"] = _mod
_spec.loader.exec_module(_mod)
