"""bench.py's ONE stdout line (round 6): the driver parses the last stdout line; round 5's was 19.8 KB of notes, dispatch texts and
whole per-config records and did not parse (BENCH_r05.json: parsed = null).  The line is now a summary built by
bench.compact_line from the full record -- strict JSON (no NaN / Infinity), under LINE_CAP bytes whatever the record holds --
and the full record goes to stderr and bench_full.json.  CPU only: the canned record is round 5's last full line."""
import glob
import json
import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def bench():
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        import bench as b
    finally:
        sys.argv = argv
    return b


def _canned():
    path = os.path.join(ROOT, "profiles", "r05_bench", "bench_default_line_d.json")
    return json.loads(open(path).read().strip().splitlines()[-1])


def _dump(bench, rec):
    text = json.dumps(bench.compact_line(rec), allow_nan=False, separators=(",", ":"))
    assert "\n" not in text
    return text


def test_default_line_is_compact_strict_json_and_keeps_the_contract(bench):
    full = _canned()
    assert len(json.dumps(full)) > 15000          # the record that did not parse
    text = _dump(bench, full)
    assert len(text) < bench.LINE_TARGET < bench.LINE_CAP == 8192
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["vs_baseline"] is None and line["config"]["workload"].startswith("c3: 10000000x768")
    assert "model" not in line["config"]
    rl = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "min_bytes", "traffic_over_min", "kernel_ms", "kernel"):
        assert key in rl, key
    assert rl["bound"] == "hbm" and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-4
    assert abs(rl["frac"] - full["roofline"]["frac"]) < 1e-5 and abs(line["value"] / full["value"] - 1) < 1e-5
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"] and cb["parity_ok"] is True
    # summaries only for the other configurations
    for name in ("c2", "refbench", "c4_shard_1rank_rccl", "c5", "refbench_from_parquet"):
        s = line["configs"][name]
        assert s["parity_ok"] is True and "dispatch" not in s and "config" not in s
        assert len(json.dumps(s)) < 400
    assert line["secondary_mixture"]["roofline"]["frac"] == pytest.approx(full["secondary_mixture"]["roofline"]["min_bytes_frac"], rel=1e-5)
    assert line["single_query"]["p50_us"] == pytest.approx(full["single_query"]["p50_us"], rel=1e-5)
    assert line["index_build"]["roofline"]["bound"] == "mfma"
    # nothing that reads like prose survived
    assert not any(k.endswith("note") or k.endswith("_label") or k == "dispatch" for k in json.dumps(line).replace('"', " ").split())


def test_line_survives_non_finite_numbers_oversized_strings_and_failed_configs(bench):
    full = _canned()
    full["roofline"]["traffic"] = float("nan")
    full["roofline"]["traffic_over_min"] = float("inf")
    full["single_query"]["p99_us"] = float("-inf")
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["configs"]["c5"] = {"error": "x" * 5000}
    full["configs"]["c2"]["roofline"]["min_bytes_frac"] = float("nan")
    full["configs"].update({f"extra{i}": dict(full["configs"]["refbench"]) for i in range(12)})
    text = _dump(bench, full)
    assert len(text) < bench.LINE_CAP
    line = json.loads(text)
    assert line["roofline"]["traffic"] is None and len(line["config"]["workload"]) <= 160
    assert "error" in line["configs"]["c5"] and len(line["configs"]["c5"]["error"]) <= 160
    # far beyond anything a run produces: the optional sections are shed, the contract keys stay
    full["configs"].update({f"more{i}": dict(full["configs"]["refbench"]) for i in range(400)})
    text = _dump(bench, full)
    assert len(text) < bench.LINE_CAP
    line = json.loads(text)
    assert "dropped" in line["configs"] and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


def test_every_committed_round5_record_compacts(bench):
    seen = 0
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[45]_bench", "*.json"))):
        try:
            rec = json.loads(open(path).read().strip().splitlines()[-1])
        except (ValueError, IndexError):
            continue
        if not isinstance(rec, dict) or "metric" not in rec:
            continue
        text = _dump(bench, rec)
        assert len(text) < bench.LINE_CAP, path
        line = json.loads(text)
        assert line["metric"] == rec["metric"] and math.isclose(line["value"], rec["value"], rel_tol=1e-5), path
        seen += 1
    assert seen >= 5


def test_eight_rank_line_fits(bench):
    """The --gpus 8 line: eight ranks' step times, row-group ranges, bases and loader figures ride on it."""
    full = _canned()
    full.update({"n_gpus": 8, "per_rank_ms_per_step": {"min": 2.1, "max": 2.4, "ranks": [2.1 + 0.04 * i for i in range(8)], "note": "n" * 300},
                 "exchange": {"ranks": 8, "backend": "RCCL", "ms_per_step": 0.05, "share_of_step": 0.02, "bytes_per_rank_per_step": 81920,
                              "collective": "c" * 300},
                 "replicas": {"value": 7.0e7, "unit": "queries/s", "ms_per_step": 0.11, "note": "n" * 300}})
    full["config"].update({"shards": 8, "row_groups": 33, "row_group_ranges": [[4 * i, 4 * i + 4] for i in range(8)],
                           "row_bases": [12500000 * i for i in range(8)], "shard_rows": [12500000] * 8})
    full["per_rank"] = {"load_s": [1.5] * 8, "build_s": [0.13] * 8, "loader_GBps": [25.0] * 8, "loader_path": "data pages walked"}
    text = _dump(bench, full)
    assert len(text) < bench.LINE_CAP
    line = json.loads(text)
    assert len(line["per_rank_ms_per_step"]["ranks"]) == 8 and line["exchange"]["ranks"] == 8 and len(line["config"]["row_bases"]) == 8


def test_a_line_always_comes_out(bench, monkeypatch):
    """Whatever goes wrong while the summary is built -- an exception, a summary beyond the cap -- the run still prints the contract
    keys with `roofline` and `cpu_baseline`."""
    full = _canned()
    monkeypatch.setattr(bench, "compact_line", lambda r: (_ for _ in ()).throw(KeyError("boom")))
    line = json.loads(bench.line_text(full))
    assert line["metric"] == full["metric"] and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["kind"] == "port"
    monkeypatch.setattr(bench, "compact_line", lambda r: {"x": "y" * 20000})
    text = bench.line_text(full)
    assert len(text) < bench.LINE_CAP and json.loads(text)["config"]["workload"].startswith("c3")
