"""Study records and regret curves in the reference's result format.

  compute_regrets     scamlgp/benchmarking/plotting.py:21-53  (running minimum of loss - optimum;
                      warns on regrets below -1e-6)
  study_record        scamlgp/benchmarking/local_runner.py:79-84  (optimum / objectives / evaluations / seed)
  write_study         scamlgp/benchmarking/local_runner.py:188-201 (one JSON file per study:
                      <experiment_key>_<seed>_<config hash>.json with a one-element "studies" list)

so that BO runs of `scamlgp_amd.bo.ScaMLGPBOLoop` can be compared with the reference's regret plots.
Plain Python: nothing here touches the GPU path.
"""
from __future__ import annotations

import hashlib
import itertools
import json
import os
import time
import warnings
from typing import Any, Dict, Iterable, List, Optional, Sequence


def compute_regrets(objective_name: str, optimum: float, objective_values: Sequence[Dict[str, float]],
                    greater_is_better: bool = False) -> List[float]:
    """Simple regret curve (what scamlgp/benchmarking/plotting.py:21-53 plots): the instantaneous regret
    sign * (value_t - optimum) of every evaluation, then its cumulative minimum.  A regret below -1e-6 (the tabulated
    optimum of some benchmarks is itself an estimate) is reported with a warning, as the reference does."""
    sign = -1.0 if greater_is_better else 1.0
    inst = [sign * ovs[objective_name] - sign * optimum for ovs in objective_values]
    for r in inst:
        if r < -1e-6:
            warnings.warn(f"A negative regret was detected. The regret value was {r}.", Warning)
    return list(itertools.accumulate(inst, min))


def study_record(X: Iterable[Sequence[float]], Y_noisy: Iterable[float], Y_noise_free: Optional[Iterable[float]] = None, *,
                 seed: int, optimum: Optional[float], objective_name: str = "loss", greater_is_better: bool = False,
                 parameter_names: Optional[Sequence[str]] = None) -> Dict[str, Any]:
    """One study in the reference's layout: evaluations carry the configuration and both the noisy and the
    noise-free objective ("<name> (noisy)" / "<name> (noise free)")."""
    evaluations = []
    nf = list(Y_noise_free) if Y_noise_free is not None else None
    for i, (x, y) in enumerate(zip(X, Y_noisy)):
        x = [float(v) for v in x]
        names = list(parameter_names) if parameter_names is not None else [f"x{j}" for j in range(len(x))]
        objectives = {f"{objective_name} (noisy)": float(y)}
        if nf is not None:
            objectives[f"{objective_name} (noise free)"] = float(nf[i])
        evaluations.append({"configuration": dict(zip(names, x)), "objectives": objectives, "optimizer_info": {}, "user_info": {},
                            "settings": {}, "context": None, "constraints": None})
    return {"optimum": optimum, "objectives": [{"name": objective_name, "greater_is_better": greater_is_better}],
            "evaluations": evaluations, "seed": int(seed)}


def regrets_of_study(study: Dict[str, Any], noise_free: bool = True) -> List[float]:
    obj = study["objectives"][0]
    key = f"{obj['name']} ({'noise free' if noise_free else 'noisy'})"
    return compute_regrets(key, study["optimum"], [e["objectives"] for e in study["evaluations"]], obj["greater_is_better"])


def config_hash_of(experiment_config: Dict[str, Any], short: bool = False) -> str:
    """SHA-256 of the JSON text of the parsed experiment configuration, optionally cut to 7 characters
    (scamlgp/benchmarking/experiment_config_utils.py:95-100: same digest, so result file names line up)."""
    digest = hashlib.sha256(json.dumps(experiment_config).encode()).hexdigest()
    return digest[:7] if short else digest


def write_study(output_dir: str, experiment_key: str, experiment_config: Dict[str, Any], study: Dict[str, Any],
                experiment_module: str = "scamlgp_amd") -> str:
    """<output_dir>/<experiment_key>_<seed>_<config hash>.json, one study per file."""
    config_hash = config_hash_of(experiment_config)
    results = dict(experiment_config=experiment_config, experiment_module=experiment_module, experiment_key=experiment_key,
                   timestamp=time.time(), studies=[study])
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, f"{experiment_key}_{study['seed']}_{config_hash}.json")
    with open(path, "w", encoding="UTF-8") as fh:
        json.dump(results, fh)
    return path
