"""Known answers for the synthetic objectives (values from the reference's
tests/benchmarking/functions_test.py:7-51, asserted there to 4 decimals)."""
import math

import numpy as np

from scamlgp_amd import synthetic as S


def test_branin_global_minimum():
    assert abs(float(S.branin(-math.pi, 12.275)) - 0.397887) < 1e-4
    assert abs(float(S.branin(math.pi, 2.275)) - 0.397887) < 1e-4


def test_hartmann3_global_minimum():
    x = np.array([[0.114614, 0.555649, 0.852547]])
    assert abs(float(S.hartmann3(x)[0]) + 3.86278) < 1e-4


def test_hartmann6_global_minimum():
    x = np.array([[0.20169, 0.150011, 0.476874, 0.275332, 0.311652, 0.6573]])
    assert abs(float(S.hartmann6(x)[0]) + 3.32237) < 1e-4


def test_task_stacks_are_seeded_and_shaped():
    a = S.branin_task_stack(3, 10, seed=5)
    b = S.branin_task_stack(3, 10, seed=5)
    c = S.branin_task_stack(3, 10, seed=6)
    assert a["X"].shape == (3, 10, 2) and a["Y"].shape == (3, 10)
    np.testing.assert_array_equal(a["Y"], b["Y"])
    assert not np.allclose(a["Y"], c["Y"])
    for k, (lo, hi) in zip(range(6), S.BRANIN_PARAM_RANGES.values()):
        assert np.all((a["params"][:, k] >= lo) & (a["params"][:, k] <= hi))
    h = S.hartmann6_task_stack(2, 7, seed=1)
    assert h["X"].shape == (2, 7, 6) and h["params"].shape == (2, 4)
    f = S.smooth_field_task_stack(2, 9, 8)
    assert f["X"].shape == (2, 9, 8) and np.isfinite(f["Y"]).all()


def test_standardize_rows_matches_botorch_semantics():
    Y = np.array([[1.0, 2.0, 4.0], [3.0, 3.0, 3.0]])
    Ys, m, s = S.standardize_rows(Y)
    np.testing.assert_allclose(m, [7 / 3, 3.0])
    np.testing.assert_allclose(s, [np.std([1, 2, 4], ddof=1), 1.0])
    np.testing.assert_allclose(Ys[1], 0.0)


def test_quadratic_family_and_hartmann3_stack_and_sobol_design():
    # functions/quadratic.py:31: f = (a (x + b))^2 + c; minimum c at x = -b
    assert float(S.quadratic(-0.3, a=1.2, b=0.3, c=-0.5)) == -0.5
    np.testing.assert_allclose(S.quadratic(np.array([0.0, 1.0]), 2.0, 0.5, 1.0), [2.0, 10.0])
    q = S.quadratic_task_stack(4, 16, seed=2)
    assert q["X"].shape == (4, 16, 1) and q["params"].shape == (4, 3)
    for k, (lo, hi) in enumerate(S.QUADRATIC_PARAM_RANGES.values()):
        assert np.all((q["params"][:, k] >= lo) & (q["params"][:, k] <= hi))
    x = -1.0 + 2.0 * q["X"][1, :, 0]
    np.testing.assert_allclose(q["Y"][1], (q["params"][1, 0] * (x + q["params"][1, 1])) ** 2 + q["params"][1, 2])
    h = S.hartmann3_task_stack(3, 8, seed=1, design="sobol")
    assert h["X"].shape == (3, 8, 3) and h["params"].shape == (3, 4) and np.all((h["X"] >= 0) & (h["X"] < 1))
    # a scrambled Sobol design is seeded and fills the cube more evenly than its size in random points would
    a = S.unit_cube_design(1, 64, 2, np.random.default_rng(0), "sobol")
    b = S.unit_cube_design(1, 64, 2, np.random.default_rng(0), "sobol")
    np.testing.assert_array_equal(a, b)
    counts = np.histogram2d(a[0, :, 0], a[0, :, 1], bins=4, range=[[0, 1], [0, 1]])[0]
    assert counts.min() == counts.max() == 4   # 64 points: exactly 4 in each of the 16 cells
