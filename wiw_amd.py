"""Import alias: `import wiw_amd` loads the package that lives in `world-in-world_amd/`
(the directory name required by the project layout is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "world-in-world_amd")
_spec = importlib.util.spec_from_file_location(
    "wiw_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["wiw_amd"] = _mod
_spec.loader.exec_module(_mod)
