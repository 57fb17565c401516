"""The line bench.py prints must stay parseable by the driver: ONE compact JSON object below 4 KB carrying the contract's
fields, `roofline` and `cpu_baseline`, whatever the side legs grow to (round 5's 21 KB line left BENCH_r05.parsed null)."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _full_objects():
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_bench_default_run.json"))):
        try:
            obj = json.load(open(fn))
        except ValueError:
            continue
        if isinstance(obj, dict) and "metric" in obj:
            yield os.path.basename(fn), obj


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


@pytest.mark.parametrize("name,full", list(_full_objects()))
def test_compact_line_of_every_committed_full_run(name, full):
    b = _bench()
    line = json.dumps(b.compact(full, "gpurun_out/bench_full.json"), separators=(",", ":"))
    assert len(line) < 4096, (name, len(line))
    assert "\n" not in line
    back = json.loads(line)
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == pytest.approx(full["value"], rel=1e-5)
    assert back["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-5)
    assert "workload" in back["config"] and "model" not in back["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"], k
    assert back["roofline"]["frac"] == pytest.approx(back["roofline"]["achieved"] / back["roofline"]["peak"], rel=1e-4)
    if "cpu_baseline" in full:
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"], k


def test_compact_line_stays_bounded_when_the_side_legs_grow():
    b = _bench()
    name, full = max(_full_objects(), key=lambda t: len(json.dumps(t[1])))
    fat = json.loads(json.dumps(full))
    # forty more side legs with prose, long workload strings, a long per-rank list
    for i in range(40):
        fat.setdefault("streamed_rows", {})["m%d" % (20000 + i)] = {
            "workload": "x" * 400, "fits_per_s": 1.0e6 + i, "ms_per_step": 1.0,
            "roofline": {"kernel": "k" * 300, "bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": None}}
    fat["config"]["workload"] = "w" * 2000
    fat["value_definition"] = "v" * 2000
    line = json.dumps(b.compact(fat, "gpurun_out/bench_full.json"), separators=(",", ":"))
    assert len(line) < 4096
    back = json.loads(line)
    for k in CONTRACT + ("roofline",):
        assert k in back
    if "cpu_baseline" in fat:
        assert "cpu_baseline" in back


def test_emit_writes_the_full_object_beside_the_line(tmp_path, monkeypatch):
    b = _bench()
    name, full = next(iter(_full_objects()))
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    line = b.emit(full)
    assert len(line) < 4096
    back = json.loads(line)
    side = json.load(open(os.path.join(str(tmp_path), back["extra_file"])))
    assert side["value"] == full["value"]
    assert json.load(open(os.path.join(str(tmp_path), "bench_full.json")))["metric"] == full["metric"]
