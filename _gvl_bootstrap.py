"""Registers the hyphen-named package directory `grounded-video-llm_amd/` as the importable
module `grounded_video_llm_amd` (a directory name with '-' cannot be imported directly)."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "grounded-video-llm_amd")
_NAME = "grounded_video_llm_amd"


def ensure():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


ensure()
