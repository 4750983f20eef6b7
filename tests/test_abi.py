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


def test_round3_entry_points_validate_before_any_hip_call():
    """scaml_target_mll_f64 / scaml_target_fit_f64 / scaml_posterior_linv_grad_f64 / scaml_target_posterior_grad_f64 /
    scaml_linv_batched_lower_f64: argument checks and size limits answer without a GPU."""
    L = _lib.lib
    one = ctypes.c_void_p(16)   # never dereferenced: validation fails first
    spec = (ctypes.c_double * 19)(1e-4, 1e2, 1e-4, 1e2, 1e-8, 1e-2, 2, 0.5, 1.5, 2, -2.0, 3.0, 2, -8.0, 2.0, 1, 1.0, 1.0, 1e-10)
    assert L.scaml_target_fit_max_d() >= 8
    assert L.scaml_target_fit_max_n(32, 6) >= 128 and L.scaml_target_fit_max_n(32, 6) < 4096
    assert L.scaml_target_fit_max_n(32, 99) == 0 and L.scaml_target_fit_max_n(0, 6) == 0
    assert L.scaml_target_fit_workspace_doubles(3, 32, 6, 10) == 3 * 26 * 40
    f = L.scaml_target_mll_f64
    # NULL spec / NULL grad / NULL info
    assert f(one, one, one, one, 0.0, 1.0, None, one, 1, 8, 3, 2, 0, one, one, one, None, None) == _lib.E_BADARG
    assert f(one, one, one, one, 0.0, 1.0, spec, one, 1, 8, 3, 2, 0, one, None, one, None, None) == _lib.E_BADARG
    assert f(one, one, one, one, 0.0, 1.0, spec, one, 1, 8, 3, 2, 0, one, one, None, None, None) == _lib.E_BADARG
    # s_all must be positive; kernel kind; D and n beyond the kernel
    assert f(one, one, one, one, 0.0, 0.0, spec, one, 1, 8, 3, 2, 0, one, one, one, None, None) == _lib.E_BADARG
    assert f(one, one, one, one, 0.0, 1.0, spec, one, 1, 8, 3, 2, 5, one, one, one, None, None) == _lib.E_BADARG
    assert f(one, one, one, one, 0.0, 1.0, spec, one, 1, 8, 3, 17, 0, one, one, one, None, None) == _lib.E_TOOLARGE
    assert f(one, one, one, one, 0.0, 1.0, spec, one, 1, 1000, 3, 2, 0, one, one, one, None, None) == _lib.E_TOOLARGE
    # an inverted interval / an unknown prior kind in the constraint block
    bad = (ctypes.c_double * 19)(*spec)
    bad[0], bad[1] = 1.0, 0.5
    assert f(one, one, one, one, 0.0, 1.0, bad, one, 1, 8, 3, 2, 0, one, one, one, None, None) == _lib.E_BADARG
    bad = (ctypes.c_double * 19)(*spec)
    bad[6] = 7
    assert f(one, one, one, one, 0.0, 1.0, bad, one, 1, 8, 3, 2, 0, one, one, one, None, None) == _lib.E_BADARG
    # no start points: a no-op
    assert f(one, one, one, one, 0.0, 1.0, spec, one, 0, 8, 3, 2, 0, one, one, one, None, None) == 0
    g = L.scaml_target_fit_f64
    # workspace too small / history out of range
    assert g(one, one, one, one, 0.0, 1.0, spec, one, 2, 8, 3, 2, 0, 50, 10, 1e-5, 1e-9, one, one, None, None, one, 10, None) == _lib.E_BADARG
    assert g(one, one, one, one, 0.0, 1.0, spec, one, 2, 8, 3, 2, 0, 50, 99, 1e-5, 1e-9, one, one, None, None, one, 10 ** 6, None) == _lib.E_BADARG
    h = L.scaml_posterior_linv_grad_f64
    assert h(one, one, one, one, one, one, None, None, None, one, 2, 64, 3, 4, 16, 0, one, one, one, 0, None) == _lib.E_TOOLARGE   # D > 15
    assert h(one, None, one, one, one, one, None, None, None, one, 2, 64, 3, 4, 3, 0, one, one, one, 0, None) == _lib.E_BADARG     # Ma > 0 without Xa
    assert h(one, one, one, one, one, one, None, None, None, one, 2, 64, 3, 200, 3, 0, one, one, one, 0, None) == _lib.E_TOOLARGE  # Ma > 96
    t = L.scaml_target_posterior_grad_f64
    assert t(one, None, one, one, one, one, one, one, 1.0, None, 4, 3, 2, 0, one, one, None) == _lib.E_BADARG                      # NULL mu_g
    assert t(one, one, one, one, one, one, one, one, 0.0, None, 4, 3, 2, 0, one, one, None) == _lib.E_BADARG                       # s_all
    assert t(one, one, one, one, one, one, one, one, 1.0, None, 4, 0, 2, 0, one, one, None) == 0                                   # no query points
    assert L.scaml_linv_batched_lower_f64(None, one, None, 1, 16, one, None) == _lib.E_BADARG
    assert L.scaml_linv_batched_lower_f64(one, one, None, 1, 100000, one, None) == _lib.E_TOOLARGE
