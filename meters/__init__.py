"""Import shim: makes the repo's product package, which lives in the directory literally named
``meters.lv2_amd/`` (the reference is x42/meters.lv2), importable as ``meters.lv2_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "meters.lv2_amd")
_spec = importlib.util.spec_from_file_location(
    "meters.lv2_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
lv2_amd = importlib.util.module_from_spec(_spec)
sys.modules["meters.lv2_amd"] = lv2_amd
_spec.loader.exec_module(lv2_amd)
