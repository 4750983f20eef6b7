"""CPU tests of the study / regret format (mirror of scamlgp/benchmarking/plotting.py:21-53 and
local_runner.py:79-84, 188-201)."""
import json
import warnings

import pytest

from scamlgp_amd import results


def test_compute_regrets_running_minimum_and_sign():
    vals = [{"loss": 3.0}, {"loss": 5.0}, {"loss": 1.5}, {"loss": 2.0}]
    assert results.compute_regrets("loss", 1.0, vals) == [2.0, 2.0, 0.5, 0.5]
    up = [{"acc": 0.2}, {"acc": 0.9}, {"acc": 0.5}]
    assert results.compute_regrets("acc", 1.0, up, greater_is_better=True) == pytest.approx([0.8, 0.1, 0.1])


def test_negative_regret_warns():
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r = results.compute_regrets("loss", 1.0, [{"loss": 0.9}])
    assert r == pytest.approx([-0.1]) and any("negative regret" in str(x.message) for x in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        results.compute_regrets("loss", 1.0, [{"loss": 1.0 - 1e-9}])
    assert not w


def test_study_record_round_trip(tmp_path):
    X = [[0.1, 0.2], [0.3, 0.4], [0.5, 0.6]]
    study = results.study_record(X, [2.0, 1.0, 1.5], [1.8, 1.1, 1.4], seed=7, optimum=0.4)
    assert study["seed"] == 7 and len(study["evaluations"]) == 3
    assert study["evaluations"][1]["configuration"] == {"x0": 0.3, "x1": 0.4}
    assert set(study["evaluations"][0]["objectives"]) == {"loss (noisy)", "loss (noise free)"}
    assert results.regrets_of_study(study) == pytest.approx([1.4, 0.7, 0.7])
    assert results.regrets_of_study(study, noise_free=False) == pytest.approx([1.6, 0.6, 0.6])
    path = results.write_study(str(tmp_path), "branin_scamlgp", {"n_studies": 1, "optimizer": "ScaMLGP"}, study)
    loaded = json.load(open(path))
    assert loaded["experiment_key"] == "branin_scamlgp" and loaded["studies"][0]["optimum"] == 0.4
    assert path.endswith(".json") and "_7_" in path
