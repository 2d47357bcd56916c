"""Import alias for the package directory ``micro-aes_amd/`` (a hyphen is not a
legal module name): ``import micro_aes_amd`` loads that directory as a package
under this name, sub-modules included (``micro_aes_amd.sharding``)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro-aes_amd")
_spec = importlib.util.spec_from_file_location(
    "micro_aes_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["micro_aes_amd"] = _mod
_spec.loader.exec_module(_mod)
