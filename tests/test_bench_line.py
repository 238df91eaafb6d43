"""The bench line the driver parses stays small and well-formed (a 25.6 KB line was not parsed).

compact_line() is fed recorded FULL results (profiles/r5_bench_c3_final.json — the very record that broke the driver's parser —,
the FAUST line and a two-rank line) and must give one JSON line under bench.LINE_LIMIT that round-trips through json.loads and
keeps every key the contract and SURVEY.md §8(d) name."""
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _full(name):
    with open(os.path.join(ROOT, "profiles", name)) as fh:
        txt = fh.read().strip()
    return json.loads(txt.splitlines()[-1] if txt.count("\n") and not txt.lstrip().startswith("{\n") else txt)


@pytest.mark.parametrize("name", ["r5_bench_c3_final.json", "r4_bench_c3_final.json", "r5_bench_c3_graph.json",
                                  "r5_bench_gpus2_gloo_one_device.json", "r5_bench_faust_n1.json"])
def test_recorded_results_give_a_small_line(name):
    full = _full(name)
    line = json.dumps(bench.compact_line(full, "bench_detail.json"), separators=(",", ":"))
    assert len(line) < bench.LINE_LIMIT, len(line)
    back = json.loads(line)
    for k in CONTRACT:
        assert k in back, k
    assert back["value"] == pytest.approx(full["value"], rel=1e-3) and back["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-3)
    assert back["config"]["workload"] and "model" not in back["config"]
    if full.get("roofline"):
        roof = back["roofline"]
        assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
        assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], rel=2e-3)
        assert roof["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-3)
        assert set(roof["frac_by_convention"]) == {"product_plus_epilogue_operands", "product_bytes_only", "measured_hbm_traffic"}
    if full.get("cpu_baseline"):
        assert set(back["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    # only scalars (or one level of scalars) under `secondary`: the tables live in the detail file
    for k, v in (back.get("secondary") or {}).items():
        if isinstance(v, dict):
            assert all(not isinstance(x, (dict, list)) for x in v.values()), k
        else:
            assert not isinstance(v, list), k
    assert back["detail"] == "bench_detail.json"


def test_emit_writes_the_detail_file_and_one_line(tmp_path, monkeypatch):
    full = _full("r5_bench_c3_final.json")
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r, w = os.pipe()
    bench.emit(full, w)
    os.close(w)
    with os.fdopen(r) as fh:
        txt = fh.read()
    assert txt.endswith("\n") and txt.count("\n") == 1 and len(txt) < bench.LINE_LIMIT
    line = json.loads(txt)
    with open(tmp_path / line["detail"]) as fh:
        assert json.load(fh) == full


def test_an_oversized_secondary_is_dropped_not_printed(tmp_path, monkeypatch):
    full = _full("r5_bench_c3_final.json")
    full["secondary"]["frac_min_packed_by_order"] = {f"order{i}": 0.5 for i in range(600)}
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    r, w = os.pipe()
    bench.emit(full, w)
    os.close(w)
    with os.fdopen(r) as fh:
        txt = fh.read()
    assert len(txt) < bench.LINE_LIMIT and "dropped" in json.loads(txt)["secondary"]
