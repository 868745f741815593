"""bench.py's JSON line (the driver's contract): the metric string is BASELINE.json's, and the line committed under profiles/
for the headline configuration carries every field the contract names, with roofline and cpu_baseline objects that are
internally consistent."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_metric_is_the_baseline_metric():
    bench = _bench_module()
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert bench.METRIC == baseline["metric"]
    assert "1024x1024" in bench.WORKLOADS["c2"][0] and bench.WORKLOADS["c2"][1:] == ((1024, 1024), 1024, 16)  # configs[1]
    assert bench.HBM_PEAK_GBPS == 8000.0


def test_committed_headline_line_follows_the_contract():
    line = json.load(open(os.path.join(ROOT, "profiles", "r01d_bench_c2_1gpu.json")))
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], typ), key
    assert line["vs_baseline"] is None and line["n_gpus"] == 1 and line["unit"] == "Msamples/s" and line["dtype"] == "f32"
    assert "workload" in line["config"] and "model" not in line["config"]
    samples = 1024 * 1024 * 1024
    assert abs(line["value"] - samples * line["steps"] / (line["ms_per_step"] * line["steps"] * 1e-3) / 1e6) < 1e-6 * line["value"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = algorithmic bytes per launch / the kernel's HIP-event duration
    assert abs(r["achieved"] - r["algorithmic_bytes_per_sample"] * samples / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or 0 < r["traffic"] < r["algorithmic_bytes_per_sample"] * samples  # PMC traffic: far below the algorithmic bytes
    assert r["kernel"].startswith("lrd::megapath_kernel<")
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Msamples/s" and "oracle" in c["sample"]
    # the rocprofv3 kernel trace committed next to it agrees with the HIP-event time of the bench line
    prof = json.load(open(os.path.join(ROOT, "profiles", "r01d_c2_1024spp.json")))
    assert abs(prof["kernel_ms_mean"] - r["kernel_ms"]) < 0.01 * r["kernel_ms"]
