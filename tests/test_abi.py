"""CPU checks of the drop-in boundary: the C-ABI library builds/loads here (no GPU) and exports every symbol
that include/relax_hip.h declares; the ctypes table covers the header one to one."""
import ctypes
import os
import re

import pytest

import relax_vqa_amd  # noqa: F401
from relax_vqa_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "relax_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(relax_[a-z0-9_]+)\s*\(", text)))


def test_header_and_ctypes_table_agree():
    assert sorted(_lib.PROTOTYPES) == _header_symbols()


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib, name), f"librelax_hip.so does not export {name}"
    assert lib.relax_abi_version() == 1


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from relax_vqa_amd.engine import RelaxEngine
    with pytest.raises(RuntimeError):
        RelaxEngine(0)
