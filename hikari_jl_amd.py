"""Import shim: the package directory is named `hikari.jl_amd` (not a valid Python identifier), so this
module loads it under the importable name `hikari_jl_amd`."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hikari.jl_amd")
_spec = importlib.util.spec_from_file_location("hikari_jl_amd", os.path.join(_d, "__init__.py"),
                                               submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["hikari_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
