"""The blocked fit for 256 < N <= 512 points per task (scaml_gp_fit_blocked_f64): BASELINE configs[4]'s source tasks
(scamlgp/benchmarking/configurations/hartmann6_ablation_num_points_per_task.py:17-18).  Every test runs twice: through the 2 x 2
sequence of launches (csrc/gp_fit_blocked.hip, one CU per task) and through the one-launch kernel that gives a task several CUs
(csrc/gp_fit_coop.hip; what small stacks take by default -- tests/test_coop_fit_gpu.py has the tests specific to it).  Parity against the oracle
(1e-4 alpha, 1e-3 MLL; the kernels are far inside), against the composition of library launches it replaces, the argument
contract of the C entry point, the jitter rounds on the device, and stream capture (no host synchronisation inside)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import gp_oracle as O
from scamlgp_amd import _lib, ops, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["sequence of launches", "one launch"])
def blocked_path(request, device):
    was = _lib.lib.scaml_debug_blocked_fit_path(1 if request.param == "sequence of launches" else 2)
    yield request.param
    _lib.lib.scaml_debug_blocked_fit_path(was)


def _stack(T, N, D, seed):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    y = torch.sin(3.0 * X.sum(-1)) + 0.1 * torch.randn(T, N, dtype=torch.float64, generator=g)
    y = (y - y.mean(-1, keepdim=True)) / y.std(-1, keepdim=True)
    theta = torch.cat([0.4 + torch.rand(T, D, dtype=torch.float64, generator=g), 0.5 + torch.rand(T, 1, dtype=torch.float64, generator=g),
                       1e-3 + 1e-2 * torch.rand(T, 1, dtype=torch.float64, generator=g)], 1)
    return X, y, theta


def _composed(*a, **k):
    ops._FORCE_COMPOSED_TWO_BLOCK = True
    try:
        return ops.gp_fit_fused(*a, **k)
    finally:
        ops._FORCE_COMPOSED_TWO_BLOCK = False


@pytest.mark.parametrize("kind", [O.KIND_RBF, O.KIND_MATERN52])
@pytest.mark.parametrize("N,D", [(272, 3), (320, 12), (448, 6), (512, 6), (512, 1)])
def test_blocked_fit_matches_oracle_and_composition(N, D, kind, device):
    T = 3
    X, y, theta = _stack(T, N, D, 7 * N + D + kind)
    Xd, yd, thd = X.to(device), y.to(device), theta.to(device)
    assert N % 16 == 0 and D <= _lib.lib.scaml_fit_blocked_max_d()          # -> the blocked entry point runs
    out = ops.gp_fit_fused(Xd, yd, thd, kind)
    ref = O.gp_fit_stack_loop(X, y, theta, kind)
    assert not out["info"].cpu().any() and out["jitter"].cpu().tolist() == ref["jitter"].tolist()
    torch.testing.assert_close(out["L"].cpu(), ref["L"], rtol=1e-7, atol=1e-9)          # (strict upper triangle: zeros on both sides)
    torch.testing.assert_close(out["alpha"].cpu(), ref["alpha"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["quad"].cpu(), ref["quad"], rtol=1e-7, atol=1e-8)
    torch.testing.assert_close(out["logdet"].cpu(), ref["logdet"], rtol=1e-9, atol=1e-8)
    torch.testing.assert_close(out["mll"].cpu(), ref["mll"], rtol=1e-3, atol=1e-9)
    comp = _composed(Xd, yd, thd, kind)
    torch.testing.assert_close(out["L"], comp["L"], rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(out["alpha"], comp["alpha"], rtol=1e-8, atol=1e-10)
    torch.testing.assert_close(out["Linv_diag"], comp["Linv_diag"], rtol=1e-8, atol=1e-10)
    # the inverted diagonal blocks really are the inverses of L's diagonal blocks (what the posterior / gradient kernels use)
    Lc = out["L"].cpu()
    for b in (0, 15, 16, N // 16 - 1):
        blk = Lc[:, 16 * b:16 * b + 16, 16 * b:16 * b + 16]
        torch.testing.assert_close(out["Linv_diag"][:, b].cpu() @ blk, torch.eye(16, dtype=torch.float64).expand(T, 16, 16), rtol=0, atol=1e-9)


def test_blocked_fit_is_reproducible(device):
    """Repeated launches: the factor, its inverted diagonal blocks and the scalars are identical bit for bit (no atomics in the
    strip solve, the Schur complement or the finish); alpha inherits the last-bit freedom of the fused fit's own back-substitution
    (LDS floating-point atomics in its tail) and is held to 1e-12."""
    T, N, D, kind = 5, 512, 6, O.KIND_MATERN52
    X, y, theta = (t.to(device) for t in _stack(T, N, D, 77))
    first = ops.gp_fit_fused(X, y, theta, kind)
    for _ in range(5):
        again = ops.gp_fit_fused(X, y, theta, kind)
        for k in ("L", "mll", "quad", "logdet", "Linv_diag"):
            assert torch.equal(first[k], again[k]), k
        torch.testing.assert_close(again["alpha"], first["alpha"], rtol=1e-12, atol=1e-12 * float(first["alpha"].abs().max()))
    sub = ops.gp_fit_fused(X[1:3].contiguous(), y[1:3].contiguous(), theta[1:3].contiguous(), kind)
    assert torch.equal(sub["L"], first["L"][1:3]) and torch.equal(sub["mll"], first["mll"][1:3])


def test_blocked_fit_upper_triangle_is_left_alone_without_zero_upper(device):
    T, N, D, kind = 2, 512, 4, O.KIND_MATERN52
    X, y, theta = _stack(T, N, D, 5)
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, zero_upper=False)
    ref = O.gp_fit_stack_loop(X, y, theta, kind)
    torch.testing.assert_close(torch.tril(out["L"].cpu()), ref["L"], rtol=1e-7, atol=1e-9)
    torch.testing.assert_close(out["alpha"].cpu(), ref["alpha"], rtol=1e-4, atol=1e-6)


def test_blocked_fit_ragged_tasks(device):
    T, N, D, kind = 6, 512, 5, O.KIND_RBF
    X, y, theta = _stack(T, N, D, 21)
    n = torch.tensor([512, 511, 273, 257, 256, 17], dtype=torch.int32)
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, n_points=n.to(device))
    assert not out["info"].cpu().any()
    for t in range(T):
        k = int(n[t])
        ref = O.gp_fit(X[t, :k], y[t, :k], theta[t], kind)
        torch.testing.assert_close(out["L"][t, :k, :k].cpu(), ref["L"], rtol=1e-7, atol=1e-9)
        torch.testing.assert_close(out["alpha"][t, :k].cpu(), ref["alpha"], rtol=1e-4, atol=1e-6)
        assert float(out["alpha"][t, k:].abs().sum()) == 0.0
        torch.testing.assert_close(out["mll"][t].cpu(), ref["mll"], rtol=1e-3, atol=1e-9)
        torch.testing.assert_close(out["logdet"][t].cpu(), ref["logdet"], rtol=1e-9, atol=1e-8)


def test_blocked_fit_jitter_rounds_on_the_device(device):
    """One jitter value for the WHOLE matrix of a failing task, only for failing tasks, escalating 1e-8, 1e-7, 1e-6
    (psd_safe_cholesky); a failure in the second block restarts the first block too; a hopeless task ends with info > 0."""
    T, N, D, kind = 5, 512, 3, O.KIND_RBF
    X, y, theta = _stack(T, N, D, 3)
    X[1, 300:330] = X[1, :30]          # duplicates across the blocks: the Schur complement is what fails
    theta[1, D + 1] = -2e-9
    X[2, 10:40] = X[2, 100:130]        # duplicates inside block 1
    theta[2, D + 1] = -5e-8
    X[3, 400:430] = X[3, 440:470]      # duplicates inside block 2 only, short lengthscales: block 1 is well conditioned
    theta[3, :D] = 0.05
    theta[3, D + 1] = -1e-9
    theta[4, D + 1] = -1.0             # indefinite whatever the ladder adds
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind)
    ref = O.gp_fit_stack_loop(X[:4], y[:4], theta[:4], kind)
    assert out["jitter"][:4].cpu().tolist() == ref["jitter"].tolist()
    assert ref["jitter"].tolist()[0] == 0.0 and min(ref["jitter"].tolist()[1:]) > 0.0
    info = out["info"].cpu()
    assert info[:4].tolist() == [0, 0, 0, 0] and int(info[4]) > 0
    assert bool(torch.isnan(out["mll"][4])) and bool(torch.isfinite(out["mll"][:4]).all())
    torch.testing.assert_close(out["logdet"][:4].cpu(), ref["logdet"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out["alpha"][0].cpu(), ref["alpha"][0], rtol=1e-4, atol=1e-6)
    one = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind, retry=False)
    assert (one["info"].cpu() > 0).tolist() == [False, True, True, True, True]
    # info counts pivots over the full matrix: task 3 breaks down inside the second block
    assert int(one["info"][3]) > 256 and 0 < int(one["info"][2]) <= 256


def test_blocked_fit_is_stream_capturable(device):
    """No host synchronisation, no allocation-dependent control flow inside: the whole sequence replays from a HIP graph."""
    T, N, D, kind = 4, 512, 6, O.KIND_MATERN52
    d = synthetic.hartmann6_task_stack(T, N, seed=3)
    ys, _, _ = synthetic.standardize_rows(d["Y"])
    theta = np.concatenate([np.full((T, D), 0.6), np.ones((T, 1)), np.full((T, 1), 1e-2)], 1)
    X, y, th = (torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in (d["X"], ys, theta))
    eager = ops.gp_fit_fused(X, y, th, kind)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.gp_fit_fused(X, y, th, kind)          # warm-up on the side stream
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = ops.gp_fit_fused(X, y, th, kind)
    y2 = y.flip(1).contiguous()
    y.copy_(y2)                                   # new targets in the captured buffers
    g.replay()
    torch.cuda.synchronize()
    ref = ops.gp_fit_fused(X, y2, th, kind)
    torch.testing.assert_close(out["alpha"], ref["alpha"], rtol=1e-11, atol=1e-12 * float(ref["alpha"].abs().max()))   # (last-bit freedom of the fit's tail)
    assert torch.equal(out["mll"], ref["mll"]) and torch.equal(out["L"], ref["L"])
    assert float((out["mll"] - eager["mll"]).abs().max()) > 0.0


def test_blocked_entry_point_argument_contract(device):
    lib = _lib.lib
    assert lib.scaml_fit_blocked_max_n() == 512 and lib.scaml_fit_blocked_max_d() >= 16
    assert lib.scaml_gp_fit_blocked_workspace_bytes(4, 512) > 4 * 256 * 256 * 8 and lib.scaml_gp_fit_blocked_workspace_bytes(4, 256) == 0
    T, N, D = 2, 512, 3
    X, y, theta = (t.to(device) for t in _stack(T, N, D, 1))
    L = torch.empty(T, N, N, dtype=torch.float64, device=device)
    alpha = torch.empty(T, N, dtype=torch.float64, device=device)
    info = torch.empty(T, dtype=torch.int32, device=device)
    W = torch.empty(T, N // 16, 16, 16, dtype=torch.float64, device=device)
    nbytes = lib.scaml_gp_fit_blocked_workspace_bytes(T, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)

    def call(N_=N, D_=D, L_=L, W_=W, ws_=ws, nb=nbytes, kind=1):
        return lib.scaml_gp_fit_blocked_f64(X.data_ptr(), y.data_ptr(), theta.data_ptr(), None, None, T, N_, D_, kind, L_.data_ptr() if L_ is not None else None,
                                            alpha.data_ptr(), None, None, None, info.data_ptr(), None, W_.data_ptr() if W_ is not None else None, 1,
                                            ws_.data_ptr() if ws_ is not None else None, nb, None)

    assert call() == 0
    torch.cuda.synchronize()
    assert not info.cpu().any()
    assert call(N_=256) == _lib.E_TOOLARGE and call(N_=500) == _lib.E_TOOLARGE and call(N_=528) == _lib.E_TOOLARGE
    assert call(D_=lib.scaml_fit_blocked_max_d() + 1) == _lib.E_TOOLARGE
    assert call(L_=None) == _lib.E_BADARG and call(W_=None) == _lib.E_BADARG and call(ws_=None) == _lib.E_BADARG
    assert call(nb=nbytes - 1) == _lib.E_BADARG and call(kind=7) == _lib.E_BADARG


def test_dimensions_beyond_the_blocked_limit_take_the_composed_path(device):
    T, N, kind = 2, 272, O.KIND_RBF
    D = _lib.lib.scaml_fit_blocked_max_d() + 3
    X, y, theta = _stack(T, N, D, 9)
    theta[:, :D] *= 3.0
    out = ops.gp_fit_fused(X.to(device), y.to(device), theta.to(device), kind)
    ref = O.gp_fit_stack_loop(X, y, theta, kind)
    assert not out["info"].cpu().any()
    torch.testing.assert_close(out["alpha"].cpu(), ref["alpha"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out["mll"].cpu(), ref["mll"], rtol=1e-3, atol=1e-9)


@pytest.mark.parametrize("T", [11, 32])
def test_blocked_fit_full_stack_task_groups(T, device):
    """BASELINE configs[4]'s source shape at full stack (T = 32, N = 512, D = 6, Matern) and a T that is not a multiple of 8: more
    than one task group per XCD in the strip solve (task = (idx / parts) * 8 + xcd), switched-off padding workgroups, the
    320-workgroup Schur grid.  One ragged task, one task that needs the jitter ladder.  Against the composed two-block path on
    every task, against the oracle on three, and through size-independent properties on all (L L^T = K + (noise + jitter) I,
    (K + noise I) alpha = y)."""
    N, D, kind = 512, 6, O.KIND_MATERN52
    X, y, theta = _stack(T, N, D, 1000 + T)
    n = torch.full((T,), N, dtype=torch.int32)
    n[T - 2] = 389                                  # ragged: second block partly filled
    X[3, 300:320] = X[3, 20:40]                     # duplicates across the two blocks ...
    theta[3, D + 1] = -3e-9                         # ... under a slightly negative diagonal: needs the ladder
    Xd, yd, thd, nd = X.to(device), y.to(device), theta.to(device), n.to(device)
    out = ops.gp_fit_fused(Xd, yd, thd, kind, n_points=nd)
    comp = _composed(Xd, yd, thd, kind, n_points=nd)
    assert not out["info"].cpu().any() and not comp["info"].cpu().any()
    assert out["jitter"].cpu().tolist() == comp["jitter"].cpu().tolist() and float(out["jitter"][3]) > 0.0
    torch.testing.assert_close(out["L"], comp["L"], rtol=1e-8, atol=1e-10)
    well = [t for t in range(T) if t != 3]     # (task 3 sits on a diagonal of ~1e-8 under duplicated points: condition ~1e10, alpha is
    torch.testing.assert_close(out["alpha"][well], comp["alpha"][well], rtol=1e-6, atol=1e-8)   # held by the residual check below)
    torch.testing.assert_close(out["mll"][well], comp["mll"][well], rtol=1e-9, atol=1e-12)
    torch.testing.assert_close(out["mll"][3], comp["mll"][3], rtol=1e-5, atol=0)
    for t in (0, T - 2):
        k = int(n[t])
        ref = O.gp_fit(X[t, :k], y[t, :k], theta[t], kind)
        torch.testing.assert_close(out["L"][t, :k, :k].cpu(), ref["L"], rtol=1e-6, atol=1e-8)
        torch.testing.assert_close(out["alpha"][t, :k].cpu(), ref["alpha"], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(out["mll"][t].cpu(), ref["mll"], rtol=1e-3, atol=1e-9)
    # properties on every task, on the device
    K = ops.kernel_matrix(Xd, thd, kind, add_noise=True)
    L = torch.tril(out["L"])
    for t in range(T):
        k = int(n[t])
        Kt = K[t, :k, :k] + float(out["jitter"][t]) * torch.eye(k, dtype=torch.float64, device=device)
        Lt = L[t, :k, :k]
        assert float((Lt @ Lt.T - Kt).abs().max()) <= 1e-10 * float(Kt.abs().max())
        r = Kt @ out["alpha"][t, :k] - yd[t, :k]
        assert float(r.abs().max()) <= 1e-6 * float(yd[t, :k].abs().max())
        torch.testing.assert_close(out["logdet"][t], 2.0 * torch.log(torch.diagonal(Lt)).sum(), rtol=1e-10, atol=1e-8)


def test_blocked_fit_ragged_with_uninitialised_outputs_through_the_c_abi(device):
    """include/scaml_gp.h: rows / columns >= n_t of L are never written and may hold anything.  A caller that hands over L and
    alpha buffers full of NaN bit patterns (what a C caller's uninitialised memory may be) gets finite results in the valid
    part, through the raw entry point."""
    T, N, D, kind = 4, 512, 4, O.KIND_RBF
    X, y, theta = _stack(T, N, D, 31)
    n = torch.tensor([512, 300, 256, 100], dtype=torch.int32)
    Xd, yd, thd, nd = X.to(device), y.to(device), theta.to(device), n.to(device)
    f64 = dict(dtype=torch.float64, device=device)
    L = torch.full((T, N, N), float("nan"), **f64)
    alpha = torch.full((T, N), float("nan"), **f64)
    quad, logdet, mll, jit = (torch.empty(T, **f64) for _ in range(4))
    info = torch.empty(T, dtype=torch.int32, device=device)
    linv = torch.empty(T, N // 16, 16, 16, **f64)
    nbytes = int(_lib.lib.scaml_gp_fit_blocked_workspace_bytes(T, N))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    rc = _lib.lib.scaml_gp_fit_blocked_f64(Xd.data_ptr(), yd.data_ptr(), thd.data_ptr(), nd.data_ptr(), None, T, N, D, kind, L.data_ptr(),
                                           alpha.data_ptr(), quad.data_ptr(), logdet.data_ptr(), mll.data_ptr(), info.data_ptr(), jit.data_ptr(),
                                           linv.data_ptr(), _lib.FIT_STORE_L, ws.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert not info.cpu().any()
    for t in range(T):
        k = int(n[t])
        ref = O.gp_fit(X[t, :k], y[t, :k], theta[t], kind)
        torch.testing.assert_close(torch.tril(L[t, :k, :k]).cpu(), ref["L"], rtol=1e-7, atol=1e-9)
        torch.testing.assert_close(alpha[t, :k].cpu(), ref["alpha"], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(mll[t].cpu(), ref["mll"], rtol=1e-3, atol=1e-9)
