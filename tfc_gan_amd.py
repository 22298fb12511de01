"""Importable alias of the package directory ``tfc-gan_amd/`` (a hyphen cannot appear in a Python module name).

``import tfc_gan_amd`` executes tfc-gan_amd/__init__.py as the package ``tfc_gan_amd``.
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tfc-gan_amd")
_spec = importlib.util.spec_from_file_location("tfc_gan_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tfc_gan_amd"] = _mod
_spec.loader.exec_module(_mod)
