"""Import shim: exposes the package in `sentinel2-landcover-classification_amd/` as `s2lc_amd`."""
import importlib.util
import sys
from pathlib import Path

_dir = Path(__file__).resolve().parent / "sentinel2-landcover-classification_amd"
_spec = importlib.util.spec_from_file_location("s2lc_amd", _dir / "__init__.py", submodule_search_locations=[str(_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["s2lc_amd"] = _mod
_spec.loader.exec_module(_mod)
