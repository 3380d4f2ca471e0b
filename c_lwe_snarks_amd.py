"""Importable alias for the package directory ``c-lwe-snarks_amd/`` (a hyphen cannot be imported by name).

`import c_lwe_snarks_amd` executes this file, which loads ``c-lwe-snarks_amd/__init__.py`` as the package
``c_lwe_snarks_amd`` and puts that in ``sys.modules`` in its own place.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c-lwe-snarks_amd")
_spec = importlib.util.spec_from_file_location(
    "c_lwe_snarks_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["c_lwe_snarks_amd"] = _mod
_spec.loader.exec_module(_mod)
