"""The C-ABI library builds, loads, and exports every symbol include/matchglue.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    import __graft_entry__ as ge

    return ge._load_build().build(verbose=False)


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "matchglue.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from gluefactory_amd import _native

    assert declared_symbols() == sorted(_native.EXPORTS)


def test_library_exports_every_declared_symbol(libpath):
    lib = ctypes.CDLL(libpath)
    for sym in declared_symbols():
        assert hasattr(lib, sym), sym
    want = int(re.search(r"#define\s+MG_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "matchglue.h")).read()).group(1))
    assert lib.mg_abi_version() == want


def test_struct_layouts_match_header():
    from gluefactory_amd import _native

    assert ctypes.sizeof(_native.SpConf) == 24  # 5 x int32 + float
    assert ctypes.sizeof(_native.LgConf) == 12  # 2 x int32 + float


def test_argument_errors_are_reported_not_thrown(libpath):
    lib = ctypes.CDLL(libpath)
    lib.mg_last_error.restype = ctypes.c_char_p
    assert lib.mg_set_weight(None, 0, b"x", None, None, 0) == -1
    assert b"bad argument" in lib.mg_last_error()
    assert lib.mg_destroy(None) == 0
