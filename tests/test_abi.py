"""CPU: libgoenv.so loads and exports every symbol include/goenv.h declares; pure host helpers."""
import ctypes
import os
import re

import pytest

from common import ROOT, load_golden


def header_symbols():
    src = open(os.path.join(ROOT, "include", "goenv.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(goenv_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_list_agree():
    from elf_amd import goenv
    assert header_symbols() == goenv.exported_symbols()


def test_library_exports_every_declared_symbol():
    from elf_amd import goenv
    assert os.path.exists(goenv.LIB_PATH), "build libgoenv.so first (__graft_entry__.build())"
    lib = ctypes.CDLL(goenv.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert goenv.load_library().goenv_version() == 4


def test_d4_host_helpers_match_reference_tables():
    from elf_amd import goenv
    g = load_golden("d4.json")
    for code in range(8):
        assert [goenv.aug_action(code, a) for a in range(361)] == g["coord2action"][code]
        assert [goenv.aug_action_inv(code, a) for a in range(361)] == g["action2coord"][code]
        assert goenv.aug_action_inv(code, -1) == -1


def test_missing_library_fails_loudly(tmp_path):
    from elf_amd import goenv
    saved = goenv._lib
    goenv._lib = None
    try:
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            goenv.load_library(str(tmp_path / "nope.so"))
    finally:
        goenv._lib = saved
