"""Import alias: the package directory is ``relax-vqa_amd/`` (a hyphen is not
importable), so ``import relax_vqa_amd`` lands here and is redirected to it."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "relax-vqa_amd")
_spec = _ilu.spec_from_file_location("relax_vqa_amd", _os.path.join(_pkg_dir, "__init__.py"),
                                     submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["relax_vqa_amd"] = _mod
_spec.loader.exec_module(_mod)
