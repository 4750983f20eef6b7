"""The C-ABI shared library loads without a GPU and exports every symbol that
include/scaml_gp.h declares; argument validation happens before any HIP call."""
import ctypes
import os
import re

from scamlgp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "scaml_gp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(scaml_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    declared = _declared_symbols()
    assert declared, "no declarations parsed from include/scaml_gp.h"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(dll, name), f"{name} declared in scaml_gp.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def test_version_and_limits():
    assert _lib.lib.scaml_version() >= 100
    assert _lib.lib.scaml_fit_max_n() >= 256
    assert _lib.lib.scaml_fit_max_d(256) >= 8
    assert _lib.lib.scaml_fit_max_d(32) >= _lib.lib.scaml_fit_max_d(256)


def test_bad_arguments_are_rejected_without_touching_the_gpu():
    f = _lib.lib.scaml_gp_fit_fused_f64
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    # NULL X
    assert f(None, one, one, None, None, 1, 8, 2, 0, None, None, None, None, None, one, None, None, 0, None) == _lib.E_BADARG
    # NULL info
    assert f(one, one, one, None, None, 1, 8, 2, 0, None, None, None, None, None, None, None, None, 0, None) == _lib.E_BADARG
    # STORE_L without L
    assert f(one, one, one, None, None, 1, 8, 2, 0, None, None, None, None, None, one, None, None, _lib.FIT_STORE_L, None) == _lib.E_BADARG
    # unknown kernel kind
    assert f(one, one, one, None, None, 1, 8, 2, 7, None, None, None, None, None, one, None, None, 0, None) == _lib.E_BADARG
    # N too large for the register-resident kernel
    assert f(one, one, one, None, None, 1, 100000, 2, 0, None, None, None, None, None, one, None, None, 0, None) == _lib.E_TOOLARGE
    # D too large for the LDS budget
    assert f(one, one, one, None, None, 1, 256, 5000, 0, None, None, None, None, None, one, None, None, 0, None) == _lib.E_TOOLARGE
    # empty stack is a no-op
    assert f(one, one, one, None, None, 0, 8, 2, 0, None, None, None, None, None, one, None, None, 0, None) == 0
