"""Import shim: the package directory is `ode-rl_amd/` (not a valid Python identifier), this module
makes it importable as `ode_rl_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ode-rl_amd")
_spec = importlib.util.spec_from_file_location("ode_rl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ode_rl_amd"] = _mod
_spec.loader.exec_module(_mod)
