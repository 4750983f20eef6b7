"""The fit beyond 256 points per task with several CUs per task in ONE launch (csrc/gp_fit_coop.hip behind scaml_gp_fit_blocked_f64):
what the blocked entry point takes by default while the stack leaves CUs idle (T <= #CUs / 3, or / 2 for N <= 320).  The tests of
tests/test_blocked_fit_gpu.py (oracle parity, ragged tasks, jitter ladder, argument contract, stream capture) run through it as
well; here is what is specific to the cooperating workgroups: the result does not depend on how many workgroups share a task nor
on the kind of hand-off stores, the path choice by shape, failures raised by different workgroups of a task, and the
size-independent properties at the full configs[4] stack."""
import pytest
import torch

from oracle import gp_oracle as O
from scamlgp_amd import _lib, ops

pytestmark = pytest.mark.gpu


def _stack(T, N, D, seed):
    g = torch.Generator().manual_seed(seed)
    X = torch.rand(T, N, D, dtype=torch.float64, generator=g)
    y = torch.sin(3.0 * X.sum(-1)) + 0.1 * torch.randn(T, N, dtype=torch.float64, generator=g)
    y = (y - y.mean(-1, keepdim=True)) / y.std(-1, keepdim=True)
    theta = torch.cat([0.4 + torch.rand(T, D, dtype=torch.float64, generator=g), 0.5 + torch.rand(T, 1, dtype=torch.float64, generator=g),
                       1e-3 + 1e-2 * torch.rand(T, 1, dtype=torch.float64, generator=g)], 1)
    return X, y, theta


@pytest.fixture
def one_launch(device):
    was = _lib.lib.scaml_debug_blocked_fit_path(2)
    yield
    _lib.lib.scaml_debug_blocked_fit_path(was)


def _took():
    return _lib.lib.scaml_debug_blocked_fit_path(-1)   # 1: sequence of launches, 2: one launch


def test_path_choice_by_shape(device):
    """Default: the one-launch kernel while a task can have three workgroups (two for N <= 320), the sequence of launches beyond."""
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    kind = O.KIND_MATERN52
    for T, N, want in [(2, 512, 2), (cus // 3, 512, 2), (cus // 2, 512, 1), (cus // 2, 320, 2), (cus // 2 + 8, 320, 1)]:
        X, y, theta = (t.to(device) for t in _stack(T, N, 3, T + N))
        out = ops.gp_fit_fused(X, y, theta, kind)
        assert _took() == want, (T, N)
        assert not out["info"].cpu().any()


@pytest.mark.parametrize("N,D,kind", [(512, 6, O.KIND_MATERN52), (400, 3, O.KIND_RBF)])
def test_result_does_not_depend_on_the_number_of_workgroups_per_task(one_launch, device, N, D, kind):
    """The same three tasks inside stacks of 3, 40, 70 and 120 tasks -- 8, 6, 3 and 2 workgroups per task on a 256-CU device: every tile
    has one writer and a fixed summation order, so factor, inverted diagonal blocks, alpha and the scalars are identical bit for bit."""
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    X, y, theta = _stack(120, N, D, 11)
    ref = None
    for T in (3, 40, 70, 120):
        if T > cus:
            continue
        out = ops.gp_fit_fused(X[:T].to(device), y[:T].to(device), theta[:T].to(device), kind)
        assert _took() == 2 and not out["info"].cpu().any()
        got = {k: out[k][:3].clone() for k in ("L", "alpha", "mll", "quad", "logdet", "Linv_diag")}
        if ref is None:
            ref = got
            chk = O.gp_fit_stack_loop(X[:3], y[:3], theta[:3], kind)
            torch.testing.assert_close(got["L"].cpu(), chk["L"], rtol=1e-7, atol=1e-9)
            torch.testing.assert_close(got["alpha"].cpu(), chk["alpha"], rtol=1e-4, atol=1e-6)
            torch.testing.assert_close(got["mll"].cpu(), chk["mll"], rtol=1e-3, atol=1e-9)
        else:
            for k in ref:
                assert torch.equal(ref[k], got[k]), (T, k)


def test_write_through_and_plain_hand_off_stores_agree(one_launch, device):
    """Payload stores are plain when all workgroups of a task report the same XCD, write-through (sc1) otherwise; forcing the
    write-through form must not change a bit."""
    T, N, D, kind = 9, 512, 5, O.KIND_MATERN52
    X, y, theta = (t.to(device) for t in _stack(T, N, D, 5))
    a = ops.gp_fit_fused(X, y, theta, kind)
    was = _lib.lib.scaml_debug_coop_far(1)
    try:
        b = ops.gp_fit_fused(X, y, theta, kind)
    finally:
        _lib.lib.scaml_debug_coop_far(was)
    for k in ("L", "alpha", "mll", "Linv_diag", "info", "jitter"):
        assert torch.equal(a[k], b[k]), k


def test_failures_in_different_workgroups_and_attempts(one_launch, device):
    """Breakdowns placed in block columns that belong to different workgroups of the task (column j -> part j mod 8), each task needing a
    different rung of psd_safe_cholesky's ladder; a hopeless task among them.  Jitter, status and factor against the oracle; a
    second run must give the same bits (an abandoned attempt leaves nothing behind)."""
    T, N, D, kind = 6, 512, 3, O.KIND_RBF
    X, y, theta = _stack(T, N, D, 42)
    theta[:5, :D] = 0.05                # short lengthscales: well conditioned but for the duplicated rows, whose pivots are the diagonal term
    X[0, 40:60] = X[0, 5:25]            # first breakdown in block column 1 (rows 32 .. 63): part 1
    theta[0, D + 1] = -2e-9
    X[1, 200:230] = X[1, 100:130]       # block column 6: part 6
    theta[1, D + 1] = -5e-8
    X[2, 480:500] = X[2, 300:320]       # block column 15: the last part, the one that also finishes the task
    theta[2, D + 1] = -3e-7
    theta[3, D + 1] = -1.0              # hopeless
    X[4, 300:330] = X[4, :30]           # (block column 9: part 1 again, second column of that workgroup)
    theta[4, D + 1] = -2e-9
    Xd, yd, thd = X.to(device), y.to(device), theta.to(device)
    out = ops.gp_fit_fused(Xd, yd, thd, kind)
    assert _took() == 2
    keep = [0, 1, 2, 4, 5]
    ref = O.gp_fit_stack_loop(X[keep], y[keep], theta[keep], kind)
    assert out["jitter"][keep].cpu().tolist() == ref["jitter"].tolist()
    assert ref["jitter"].tolist()[:3] == [1e-8, 1e-7, 1e-6] and ref["jitter"].tolist()[4] == 0.0
    info = out["info"].cpu()
    assert info[keep].tolist() == [0] * 5 and int(info[3]) > 0 and bool(torch.isnan(out["mll"][3]))
    torch.testing.assert_close(out["logdet"][keep].cpu(), ref["logdet"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out["L"][5].cpu(), ref["L"][4], rtol=1e-7, atol=1e-9)
    again = ops.gp_fit_fused(Xd, yd, thd, kind)
    for k in ("L", "alpha", "jitter", "info"):
        assert torch.equal(out[k][keep], again[k][keep]), k
    one = ops.gp_fit_fused(Xd, yd, thd, kind, retry=False)
    assert (one["info"].cpu() > 0).tolist() == [True, True, True, True, True, False]
    # info = the first pivot that is not positive, counted over the whole matrix: the same one the sequence of launches reports,
    # the first duplicated row (block columns 1, 6, 15 and 9: parts 1, 6, 7 and 1)
    _lib.lib.scaml_debug_blocked_fit_path(1)
    seq = ops.gp_fit_fused(Xd, yd, thd, kind, retry=False)
    _lib.lib.scaml_debug_blocked_fit_path(2)
    assert one["info"].cpu().tolist() == seq["info"].cpu().tolist()
    i = one["info"].cpu().tolist()
    assert [i[0], i[1], i[2], i[4]] == [41, 201, 481, 301]   # the first duplicated row of each task


def test_full_stack_properties(device):
    """BASELINE configs[4] at full stack (T = 32, N = 512, D = 6, Matern) through the default path (one launch, eight workgroups per
    task): L L^T = K + noise I and (K + noise I) alpha = y on every task, the oracle on three."""
    T, N, D, kind = 32, 512, 6, O.KIND_MATERN52
    X, y, theta = _stack(T, N, D, 2024)
    Xd, yd, thd = X.to(device), y.to(device), theta.to(device)
    out = ops.gp_fit_fused(Xd, yd, thd, kind)
    assert _took() == 2 and not out["info"].cpu().any()
    K = ops.kernel_matrix(Xd, thd, kind, add_noise=True)
    L = out["L"]
    torch.testing.assert_close(L @ L.transpose(-1, -2), K, rtol=1e-10, atol=1e-11)
    torch.testing.assert_close((K @ out["alpha"].unsqueeze(-1)).squeeze(-1), yd, rtol=1e-7, atol=1e-8)
    assert float(torch.triu(L, 1).abs().max()) == 0.0
    for t in (0, 13, 31):
        ref = O.gp_fit(X[t], y[t], theta[t], kind)
        torch.testing.assert_close(L[t].cpu(), ref["L"], rtol=1e-7, atol=1e-9)
        torch.testing.assert_close(out["alpha"][t].cpu(), ref["alpha"], rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(out["mll"][t].cpu(), ref["mll"], rtol=1e-3, atol=1e-9)


def test_tiny_and_empty_tasks_agree_with_the_sequence_of_launches(device):
    """n_t = 0, 1, 31, 33 next to a full task: rows / columns past n_t are an identity block that is never written (the one-launch
    kernel's buffer descriptor of L ends at row n_t); both paths give the same factor, alpha and scalars (mll = 0 for the empty task)."""
    T, N, D, kind = 5, 512, 3, O.KIND_MATERN52
    X, y, theta = (t.to(device) for t in _stack(T, N, D, 77))
    n = torch.tensor([0, 1, 31, 33, 512], dtype=torch.int32, device=device)
    res = {}
    was = _lib.lib.scaml_debug_blocked_fit_path(0)
    try:
        for path in (1, 2):
            _lib.lib.scaml_debug_blocked_fit_path(path)
            res[path] = ops.gp_fit_fused(X, y, theta, kind, n_points=n)
            assert _took() == path and not res[path]["info"].cpu().any()
    finally:
        _lib.lib.scaml_debug_blocked_fit_path(was)
    assert float(res[2]["mll"][0]) == 0.0
    for k, tol in (("L", 1e-11), ("alpha", 1e-8), ("mll", 1e-10), ("logdet", 1e-9)):
        torch.testing.assert_close(res[2][k], res[1][k], rtol=tol, atol=tol)
    assert float(res[2]["alpha"][1, 1:].abs().sum()) == 0.0 and float(res[2]["L"][2, 31:, :].abs().sum()) == 0.0
