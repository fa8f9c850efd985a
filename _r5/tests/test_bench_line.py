"""bench.py's final stdout line: strict JSON, under 4 KB, the contract's keys, scalars only per leg (round 4's
29 KB line was not parsed by the driver).  Fed with a full result of the shape main() builds -- round 4's own, re-nested."""
import importlib.util
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _canned():
    with open(os.path.join(ROOT, "profiles", "r04_default_bench_line.json")) as f:
        o = json.load(f)
    legs = {}
    for k in ("biglm", "lattice_beam13", "lattice_beam15_no_determinizer", "lattice_beam15", "lattice_beam15_pipelined"):
        legs[k] = o.pop(k)
    sp = o["service_point"]
    legs["service_point_7000_200"] = {k: v for k, v in sp.items() if not k.endswith("_at_7000_200")}
    legs["calibrated_7000_200"] = sp["calibrated_workload_at_7000_200"]
    legs["headline_at_7000_200"] = sp["headline_workload_at_7000_200"]
    legs["reference_default_limits"] = o.pop("reference_default_limits")
    legs["biglm"]["cpu_baseline"] = dict(o["cpu_baseline"], determinizer_ms_per_lattice=np.float64(41.25))
    legs["broken"] = {"error": "TimeoutExpired(" + "x" * 500 + ")", "stderr_tail": "y" * 600}
    o["legs"] = legs
    o["roofline"]["profiled_step_ms"] = np.float32(19.5)
    o["config"]["mean_active_tokens_per_frame"] = float("nan")   # must not reach the line as NaN
    return o


def _strict(line):
    def no_constants(x):
        raise ValueError("non-finite constant %s in the bench line" % x)

    return json.loads(line, parse_constant=no_constants)


def test_line_is_small_strict_json():
    b = _bench()
    line = b.summary_line(_canned(), "bench_detail.json")
    assert "\n" not in line
    assert len(line.encode()) < 4096
    d = _strict(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms", "whole_path_frac",
              "profiled_step_ms"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-5
    for k in ("value", "unit", "cores", "kind", "single_thread_value", "cpu_model", "sample"):
        assert k in d["cpu_baseline"], k
    scalar = (int, float, str, bool, type(None))
    for part in ("config", "roofline", "cpu_baseline"):
        assert all(isinstance(v, scalar) for v in d[part].values()), part
    assert set(d["legs"]) >= {"biglm", "lattice_beam15", "lattice_beam15_pipelined", "calibrated_7000_200"}
    for name, leg in d["legs"].items():
        assert all(isinstance(v, scalar) for v in leg.values()), name
    assert d["legs"]["biglm"]["cpu_baseline_value"] > 0 and d["legs"]["biglm"]["parity"] == "8/8"
    assert d["legs"]["calibrated_7000_200"]["bit_identical"] == "112/128"
    assert "error" in d["legs"]["broken"] and len(d["legs"]["broken"]["error"]) <= 120
    assert "mean_active_tokens_per_frame" not in d["config"]   # NaN dropped, not printed


def test_line_shrinks_instead_of_overflowing():
    b = _bench()
    o = _canned()
    for i in range(40):   # many more legs than the real run has
        o["legs"]["extra_%d" % i] = dict(o["legs"]["lattice_beam15"])
    line = b.summary_line(o, "bench_detail.json")
    assert len(line.encode()) < 4096
    _strict(line)
    o["legs"].update({"filler_%d" % i: dict(o["legs"]["lattice_beam15"]) for i in range(400)})
    line = b.summary_line(o, None)   # (never refuses to print: drops legs from the line, says how many)
    assert len(line.encode()) < 4096
    d = _strict(line)
    assert d["legs_dropped_from_line"] > 0 and "biglm" in d["legs"] and "roofline" in d and "cpu_baseline" in d


def test_detail_file_is_strict_json(tmp_path):
    b = _bench()
    p = str(tmp_path / "detail.json")
    assert b.write_detail(_canned(), p) == p
    with open(p) as f:
        d = _strict(f.read())
    assert d["legs"]["biglm"]["roofline"]["kernel"].endswith("_kernel")
    assert d["config"]["mean_active_tokens_per_frame"] is None
