"""CPU: the committed bench line (profiles/r2_bench_line.json, produced by `python bench.py` on the GPU box) carries
every field the bench contract names, with consistent arithmetic."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_contract():
    d = json.load(open(os.path.join(ROOT, "profiles", "r2_bench_line.json")))
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["metric"].split(" (")[0] in base["metric"] and "workload" in d["config"] and "model" not in d["config"]
    frames = d["n_gpus"] * d["config"]["frames_per_gpu"] * d["steps"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == d["unit"]
    k1 = d["roofline_planesweep"]
    assert k1["bound"] == "hbm" and abs(k1["frac"] - k1["achieved"] / k1["peak"]) < 1e-9
    assert abs(k1["achieved"] - k1["algorithmic_bytes_per_launch"] / k1["avg_launch_ms"] / 1e6) < 1e-6 * k1["achieved"]
    assert k1["launch_ms"]["n"] >= 50 and k1["launch_ms"]["p10"] <= k1["launch_ms"]["median"] <= k1["launch_ms"]["p90"]
    assert k1["burst_frac"] > 0 and abs(k1["burst_frac"] - k1["algorithmic_bytes_per_launch"] / k1["burst_avg_launch_ms"] / 1e6 / k1["peak"]) < 1e-9
    # round 2: per-step HIP-event percentiles, rank spread, secondary workloads (never the headline), CPU model in the sample
    st = d["step_ms"]
    assert st["n"] == d["steps"] >= 50 and st["p10"] <= st["median"] <= st["p90"]
    assert d["per_rank_frames_per_s"]["min"] <= d["per_rank_frames_per_s"]["max"]
    for k, dt in (("f16", "f16"), ("config4", "f32")):
        assert d[k]["unit"] == "frames/s" and d[k]["value"] > 0 and dt in d[k]["dtype"] and d[k]["value"] != d["value"]
    if "train" in d:                                                       # added late in round 2
        assert d["train"]["unit"] == "samples/s" and d["train"]["value"] > 0 and d["train"]["ms_per_step"] > 0
    assert " on " in c["sample"]                                           # "... threads (...) on <CPU model>"
