"""Import shim: exposes the package directory `intra-and-inter-human-relation-network-for-mpee_amd/` as module `i2r_amd`."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        "intra-and-inter-human-relation-network-for-mpee_amd")
_spec = importlib.util.spec_from_file_location(
    "i2r_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["i2r_amd"] = _mod
_spec.loader.exec_module(_mod)
