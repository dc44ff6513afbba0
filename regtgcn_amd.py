"""Import shim: makes the package directory ``regt-gcn_amd/`` importable as ``regtgcn_amd``."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg_dir = os.path.join(_here, "regt-gcn_amd")
_spec = importlib.util.spec_from_file_location("regtgcn_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["regtgcn_amd"] = _mod
_spec.loader.exec_module(_mod)
