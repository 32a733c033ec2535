"""Import shim: the package directory is `water-sandbox_amd/` (hyphenated, as the repository
layout prescribes); this module loads it under the importable name `water_sandbox_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "water-sandbox_amd")
_spec = importlib.util.spec_from_file_location(
    "water_sandbox_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["water_sandbox_amd"] = _mod
_spec.loader.exec_module(_mod)
