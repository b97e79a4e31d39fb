"""bench.py's stdout contract: ONE JSON line of at most 6 KB that carries the contract's scalars,
`roofline`, `cpu_baseline` and `config.workload`; everything else goes to bench_extra.json. The input here is the
largest full record a run has produced (round 5's 21.7 KB line, profiles/r05zz_bench_steps20.json)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full_record():
    with open(os.path.join(ROOT, "profiles", "r05zz_bench_steps20.json")) as fh:
        return json.load(fh)


def test_line_is_small_and_complete():
    line = bench.compact_line(_full_record())
    assert "\n" not in line
    assert len(line.encode()) < bench.LINE_LIMIT <= 6144
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "extra"):
        assert k in d, k
    assert d["config"]["workload"].startswith("Llama-2-7B int4")
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["kind"] in ("port", "reference") and c["cores"] >= 1 and len(c["sample"]) <= 120
    assert d["parity"]["greedy_tokens_equal"] is True and d["parity"]["decode_logits_max_abs"] < 1e-3
    assert 0.0 < d["prefill"]["mfma_frac"] < 1.0


def test_line_never_exceeds_the_limit_whatever_the_record_holds():
    rec = _full_record()
    rec["configs_summary"] = "x" * 5000
    rec["note"] = "y" * 5000
    line = bench.compact_line(rec)
    assert len(line.encode()) <= bench.LINE_LIMIT
    d = json.loads(line)
    assert "roofline" in d and "cpu_baseline" in d and d["value"] > 0


def test_emit_writes_the_full_record_beside_the_line(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rec = _full_record()
    bench.emit(rec)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0])["extra"] == bench.EXTRA_FILE
    with open(tmp_path / bench.EXTRA_FILE) as fh:
        full = json.load(fh)
    assert len(full["extra_configs"]) == len(rec["extra_configs"]) and "launch_modes" in full


def test_traffic_comes_from_the_latest_counter_pass(tmp_path, monkeypatch):
    """Visit names sort like spreadsheet columns (r06k < r06z < r06aa < r06ab), not as plain strings."""
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    for name, b in (("r05z", 1.0), ("r06k", 2.0), ("r06ab", 3.0), ("r06aa", 4.0)):
        with open(tmp_path / "profiles" / (name + "_pmc_traffic.json"), "w") as fh:
            json.dump({"hbm_bytes_per_launch": b, "command": "c"}, fh)
    t = bench.read_traffic()
    assert t["source"] == "profiles/r06ab_pmc_traffic.json" and t["bytes"] == 3.0 and t["measured_in_this_run"] is False
