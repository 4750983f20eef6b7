"""CPU checks of the target-GP fit kernel's ARITHMETIC (csrc/gp_target_fit.hip) through a single-threaded host build of the
same source (tests/host_emul/target_fit_emul.cpp: test infrastructure, never part of the library): objective and analytic
gradient against torch autograd through the oracle's target_train_mll (scamlgp/model.py:360-363, 376-383; utils.py:171-177),
and the in-kernel L-BFGS against scipy L-BFGS-B on the oracle objective.  The parallel execution of the kernel (barriers, wave
reductions) is what tests/test_target_fit_gpu.py covers on the MI355X."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import scipy.optimize
import torch

torch.set_num_threads(1)   # (many small oracle ops: threads only get in each other's way)

from tests._target_problem import TARGET_SPEC, make_target_problem, oracle_mll_and_grad, pack_lower, raw_start

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "scalable-meta-learning-with-gaussian-processes_amd", "csrc")


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("emul") / "target_fit_emul.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I", CSRC,
                    os.path.join(ROOT, "tests", "host_emul", "target_fit_emul.cpp"), "-o", so], check=True)
    lib = ctypes.CDLL(so)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    lib.emul_target_fit.restype = ctypes.c_int
    lib.emul_target_fit.argtypes = [dp, dp, dp, dp, ctypes.c_double, ctypes.c_double, dp, dp] + [ctypes.c_int] * 8 + [
        ctypes.c_double, ctypes.c_double, dp, dp, ip, dp, ip]
    return lib


def _call(lib, prob, z, mode, max_iter=200, history=10):
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)
    B, P = z.shape
    arr = lambda t: np.ascontiguousarray(t.numpy() if isinstance(t, torch.Tensor) else t, dtype=np.float64)   # noqa: E731
    mt = arr(prob["source_means"].transpose(0, 1).contiguous())
    cp = arr(pack_lower(prob["source_covs"]))
    X, y, spec, zz = arr(prob["X"]), arr(prob["y"]), np.array(TARGET_SPEC, dtype=np.float64), arr(z.clone())
    value, grad = np.zeros(B), np.zeros((B, P))
    info, jit, stats = np.zeros(B, dtype=np.int32), np.zeros(B), np.zeros((B, 4), dtype=np.int32)
    P_ = lambda a: a.ctypes.data_as(dp)   # noqa: E731
    rc = lib.emul_target_fit(P_(mt), P_(cp), P_(X), P_(y), prob["m_all"], prob["s_all"], P_(spec), P_(zz), B, prob["n"], prob["T"], prob["D"],
                             prob["kind"], mode, max_iter, history, 1e-5, 2.2e-9, P_(value), P_(grad), info.ctypes.data_as(ip), P_(jit),
                             stats.ctypes.data_as(ip))
    assert rc == 0
    return dict(value=value, grad=grad, info=info, jitter=jit, stats=stats, z=zz)


@pytest.mark.parametrize("n,T,D,kind", [(1, 3, 2, 0), (7, 3, 2, 1), (20, 5, 6, 1), (33, 4, 3, 0)])
def test_objective_and_gradient_match_oracle_autograd(emul, n, T, D, kind):
    prob = make_target_problem(n, T, D, kind, seed=n)
    z = raw_start(D, T, seed=n, B=2)
    out = _call(emul, prob, z, mode=0)
    assert not out["info"].any()
    for b in range(2):
        val, g = oracle_mll_and_grad(prob, z[b])
        np.testing.assert_allclose(out["value"][b], float(val), rtol=1e-9)
        np.testing.assert_allclose(out["grad"][b], g.numpy(), rtol=1e-6, atol=1e-9)


def test_inkernel_lbfgs_reaches_scipy_optimum(emul):
    n, T, D, kind = 12, 4, 2, 1
    prob = make_target_problem(n, T, D, kind, seed=5)
    z0 = raw_start(D, T, seed=5, B=2)
    out = _call(emul, prob, z0, mode=1)
    assert (out["stats"][:, 2] != 4).all()

    def fun(zv):
        val, g = oracle_mll_and_grad(prob, torch.from_numpy(zv))
        return -float(val), -g.numpy()

    bounds = [(None, None)] * (D + 2) + [(1e-10, None)] * T
    for b in range(2):
        ref = scipy.optimize.minimize(fun, z0[b].numpy(), jac=True, method="L-BFGS-B", bounds=bounds, options=dict(maxiter=200))
        # the kernel's optimum is at least as good as scipy's (to 1e-3 of the objective), and its reported value is the oracle's there
        assert -out["value"][b] <= ref.fun + 1e-3 * max(1.0, abs(ref.fun)), (out["value"][b], ref.fun, out["stats"][b])
        val, _ = oracle_mll_and_grad(prob, torch.from_numpy(out["z"][b]))
        np.testing.assert_allclose(out["value"][b], float(val), rtol=1e-8)
        assert (out["z"][b][D + 2:] >= 1e-10).all()
