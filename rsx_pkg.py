"""Imports the package directory `libdeflate-rsx_amd/` (its name is not a valid
Python identifier) under the module name `libdeflate_rsx_amd`."""
import importlib.util
import os
import sys

_NAME = "libdeflate_rsx_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.dirname(os.path.abspath(__file__))
    pkg = os.path.join(root, "libdeflate-rsx_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg, "__init__.py"),
                                                  submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
