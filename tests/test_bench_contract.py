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
    line = json.load(open(os.path.join(ROOT, "profiles", "archive", "r02j_bench_c2_1gpu.json")))
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                     ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[key], typ), key
    assert line["vs_baseline"] is None and line["n_gpus"] == 1 and line["unit"] == "Msamples/s" and line["dtype"] == "f32"
    assert "workload" in line["config"] and "model" not in line["config"]
    samples = 1024 * 1024 * 1024
    assert abs(line["value"] - samples * line["steps"] / (line["ms_per_step"] * line["steps"] * 1e-3) / 1e6) < 1e-6 * line["value"]
    r = line["roofline"]
    # achieved / frac: the MEASURED HBM traffic (PMC passes of the same run) over the kernel's HIP-event duration -- a fraction of the
    # peak that can be one; the algorithmic figure of SURVEY 8(d) is reported next to it under its own name
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["frac"] < 1.0 and r["traffic_source"].startswith("live")
    assert abs(r["achieved"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["algorithmic_gbps"] - r["algorithmic_bytes_per_sample"] * samples / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["algorithmic_gbps"]
    assert 0 < r["traffic"] < r["algorithmic_bytes_per_sample"] * samples  # PMC traffic: far below the algorithmic bytes
    assert r["kernel"].startswith("lrd::megapath_kernel<")
    v = r["valu"]
    assert 0.3 < v["issue_frac_if_all_full_rate"] < v["issue_frac_if_all_quarter_rate"] < 1.2 and v["wave_instr_per_sample"] > 100
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Msamples/s" and "oracle" in c["sample"]
    # the other BASELINE configs ride in the same line; C1 carries the CPU leg of BASELINE configs[0] (the whole configuration)
    extra = {e["workload"].split(",")[0].split(" (")[0]: e for e in line["extra_configs"]}
    assert set(extra) == {"Cornell Box", "Bedroom-class", "Kitchen-class"}
    c1 = extra["Cornell Box"]
    assert c1["spp_timed"] == 64 and "512x512 at 64 spp" in c1["cpu_baseline"]["sample"] and c1["value"] > 100 * c1["cpu_baseline"]["value"]


def test_committed_line_times_the_reference_code_beside_the_kernel():
    """BASELINE configs[0] (Cornell 512x512, 64 spp) rides in extra_configs with BOTH CPU legs: the oracle on all cores ("port") and
    the reference's own MegaPath code through oracle/_ref on one thread ("reference")."""
    line = json.load(open(os.path.join(ROOT, "profiles", "archive", "r02j_bench_c2_1gpu.json")))
    c1 = line["extra_configs"][0]
    assert "Cornell" in c1["workload"] and c1["cpu_baseline"]["kind"] == "port" and c1["cpu_baseline"]["cores"] >= 1
    ref = c1["cpu_reference"]
    assert ref["kind"] == "reference" and ref["cores"] == 1 and ref["unit"] == "Msamples/s" and 0.01 < ref["value"] < c1["value"]


def test_product_package_never_imports_the_oracle():
    """oracle/ is the checker: nothing under luisarender_amd/ (Python or C++/HIP) may import, include, link or dlopen it."""
    import re
    pkg = os.path.join(ROOT, "luisarender_amd")
    bad = []
    for base, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".cpp", ".h", ".hip", ".inl")):
                continue
            text = open(os.path.join(base, f), errors="replace").read()
            if re.search(r"^\s*(from|import)\s+oracle\b|#include\s*[\"<][^\">]*oracle|liboracle|libref\.so|dlopen\([^)]*oracle", text, re.M):
                bad.append(os.path.join(base, f))
    assert not bad, bad
    mk = open(os.path.join(ROOT, "Makefile")).read()
    # the product libraries' link lines never name the oracle
    for line in mk.splitlines():
        if ("liblrhip.so" in line or "liblrhost.so" in line) and "-o" in line:
            assert "oracle" not in line, line


def test_round3_line_carries_parity_path_statistics_and_every_baseline_config():
    """the line of the round-3 state (profiles/archive/r03zb_bench_c2_1gpu.json, the driver's default command)"""
    line = json.load(open(os.path.join(ROOT, "profiles", "archive", "r03zb_bench_c2_1gpu.json")))
    assert line["n_gpus"] == 1 and line["value"] > 850 and line["roofline"]["traffic_source"].startswith("live") and 0.05 < line["roofline"]["frac"] < 0.5
    p = line["parity"]
    assert p["finite"] and p["samples"] == 1024 * 1024 * p["spp"] and p["rel_l1"] < 2e-2 and p["rmse_over_mean"] < 0.5 and abs(p["mean_bias"]) < 1e-3 and p["flip"] < 0.1
    assert line["rays_per_s"] > 1e9 and 2.0 < line["mean_path_length"] < 6.0 and len(line["source_hash"]) == 16
    extra = [(e["workload"].split(",")[0].split(" (")[0], e["sampler"]) for e in line["extra_configs"]]
    assert extra == [("Cornell Box", "Independent"), ("Bedroom-class", "Independent"), ("Camera-class", "Independent"), ("Kitchen-class", "Independent"),
                     ("Contemporary Bathroom-class", "PaddedSobol")]
    c1 = line["extra_configs"][0]
    assert c1["parity"]["rel_l1"] < 1e-4 and c1["cpu_reference"]["kind"] == "reference"  # full C1 against the CPU leg's frame


def test_round5_line_measures_its_roofline_block_at_the_timed_configuration():
    """the line of the round-5 state (profiles/r05_final_bench_c2_1gpu.json, the driver's default command) -- the
    roofline block measured at the timed 1024 spp and reproducible from its own fields, the low-discrepancy line with a parity of its own."""
    line = json.load(open(os.path.join(ROOT, "profiles", "r05_final_bench_c2_1gpu.json")))
    assert line["n_gpus"] == 1 and line["value"] > 1000 and line["config"]["spp"] == 1024 and "scheduler_override" not in line["config"]
    r = line["roofline"]
    assert r["kernel"] == "lrd::megapool_kernel<4096u>" and "1024 spp" in r["traffic_source"] and 0.2 < r["frac"] < 0.4
    assert abs(r["achieved"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    v = r["valu"]
    assert abs(v["issue_frac"] - v["wave_instr_per_launch"] * v["cycles_per_wave_instr"] / v["simd_cycles_per_launch"]) < 1e-9
    assert abs(v["simd_cycles_per_launch"] - v["simds"] * r["kernel_ms"] * 1e-3 * v["shader_clock_hz"]) < 1e-3 * v["simd_cycles_per_launch"]
    assert 0.9 < v["pmc_busy"] < 1.25 and v["time_elasticity_to_valu_instructions"] == 0.42 and 0.3 < v["pmc_wait_any_over_wave_cycles"] < 0.7
    assert "1024 spp" in r["lanes"]["note"] and r["lanes"]["trace"] > 0.9 and r["lanes"]["trace_starved"] < 0.01
    assert line["path_statistics"]["spp"] == 1024 and line["path_statistics"]["pool_state_bytes_per_sample"] > 200
    extra = {(e["workload"].split(",")[0].split(" (")[0], e["sampler"]): e for e in line["extra_configs"]}
    sobol = extra["Contemporary Bathroom-class", "PaddedSobol"]
    assert sobol["spp_timed"] == 1024 and sobol["value"] > 880 and sobol["kernel"] == "lrd::megapool_kernel<4098u>"
    assert sobol["parity"]["rel_l1"] < 1e-2 and sobol["parity"]["finite"] and sobol["cpu_baseline"]["kind"] == "port"
    for key, floor in ((("Cornell Box", "Independent"), 3800), (("Bedroom-class", "Independent"), 1000), (("Camera-class", "Independent"), 1000), (("Kitchen-class", "Independent"), 550)):
        assert extra[key]["value"] > floor and extra[key]["parity"]["finite"], key


def test_round6_line_carries_a_roofline_block_for_every_configuration():
    """the line of the round-6 state (profiles/r06_final_bench_c2_1gpu.json, the driver's default command) -- one
    `roofline` block per configuration, each recomputable with a calculator from the raw counter totals it carries (`pmc.counters`), priced per
    kernel from the dynamic class mix; `bound` says what DESIGN.md section 5 says; the reference's own code beside the HEADLINE number; the
    PaddedSobol line on a kernel compiled for that sampler."""
    line = json.load(open(os.path.join(ROOT, "profiles", "r06_final_bench_c2_1gpu.json")))
    assert line["n_gpus"] == 1 and line["value"] > 1050 and line["config"]["spp"] == 1024 and "scheduler_override" not in line["config"]
    ref = line["cpu_reference"]
    assert ref["kind"] == "reference" and ref["cores"] >= 1 and ref["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    blocks = [("headline", line["roofline"])] + [(e["workload"].split(",")[0], e["roofline"]) for e in line["extra_configs"]]
    assert len(blocks) == 6
    for name, r in blocks:
        assert r["bound"] == "issue + latency at 4 waves per SIMD" and r["unit"] == "GB/s" and r["peak"] == 8000.0, name
        assert "pmc_busy" not in r.get("valu", {}), name  # (an in-flight sum, not a fraction of a roof: out of the block since round 6)
        c, p = r["pmc"]["counters"], r["pmc"]
        traffic = (c["FETCH_SIZE"] * 2048.0 + c["WRITE_SIZE"] * 1024.0) / p["samples"] * p["timed_launch_samples"]
        assert abs(traffic - r["traffic"]) < 1e-9 * traffic, name
        assert abs(r["frac"] - traffic / (r["kernel_ms"] * 1e-3) / 8e12) < 1e-9, name
        v = r["valu"]
        assert abs(v["wave_instr_per_sample"] - c["SQ_INSTS_VALU"] / p["samples"]) < 1e-9 * v["wave_instr_per_sample"], name
        assert 2.4 < v["cycles_per_wave_instr"] < 4.2 and abs(sum(v["valu_mix"].values()) - 1.0) < 1e-9, name
        assert abs(v["issue_frac"] - v["wave_instr_per_launch"] * v["cycles_per_wave_instr"] / v["simd_cycles_per_launch"]) < 1e-9, name
        w = r["waves"]
        assert 0.9 < w["waiting_at_waitcnt"] + w["issue_stalled"] + w["instruction_in_flight"] < 1.1, name
        assert 0.0 < r["frac"] < 0.6 and r["lanes"]["trace"] > 0.5, name
    extra = {(e["workload"].split(",")[0].split(" (")[0], e["sampler"]): e for e in line["extra_configs"]}
    sobol = extra["Contemporary Bathroom-class", "PaddedSobol"]
    assert sobol["kernel"] == "lrd::megapool_kernel<20482u>" and sobol["value"] > 950 and sobol["parity"]["rel_l1"] < 1e-2 and sobol["parity"]["finite"]
    assert extra["Camera-class", "Independent"]["kernel"] == "lrd::megapool_kernel<12308u>" and extra["Camera-class", "Independent"]["value"] > 1150
    for key, floor in ((("Cornell Box", "Independent"), 3800), (("Bedroom-class", "Independent"), 1050), (("Kitchen-class", "Independent"), 590)):
        assert extra[key]["value"] > floor and extra[key]["parity"]["finite"], key


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_multi_gpu_path_runs_as_the_driver_launches_it(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` with N = 1 and the collective forced: the code the
    driver's 2 / 4 / 8-GPU runs take (process group, C-ABI communicator, lrhip_film_reduce inside the timed region, the reduced film
    checked against a 1-GPU render) on a one-GPU box.  The line must carry the multi_gpu record with a passed film check."""
    import subprocess
    import sys
    env = dict(os.environ, LR_BENCH_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "c1", "--spp", "8",
           "--no-cpu-baseline", "--no-pmc", "--no-stats", "--extra-spp", "4"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "lrhip_film_reduce" in line["config"]["collective"], line["config"]
    m = line["multi_gpu"]
    assert m["reduced_film_equals_1gpu_render"] is True and m["reduce_ms"] >= 0.0 and m["reduce_bytes"] == 512 * 512 * 16, m
    # round 5: a multi-GPU run also carries the configurations BASELINE.json names for 8 GPUs -- C4 (3840 x 2160) and C5
    # (wavefront mode) -- each sharded, reduced and checked like the headline frame
    extra = {e["workload"].split("-class")[0]: e for e in line["extra_configs"]}
    assert set(extra) == {"Camera", "Kitchen"}, list(extra)
    for name, res in (("Camera", (3840, 2160)), ("Kitchen", (1280, 720))):
        e = extra[name]
        assert e["value"] > 0 and e["spp_timed"] == 4 and "lrhip_film_reduce" in e["collective"], e
        assert e["multi_gpu"]["reduced_film_equals_1gpu_render"] is True and e["multi_gpu"]["checked_sample_counts_ok"] is True, e["multi_gpu"]
        assert e["multi_gpu"]["reduce_bytes"] == res[0] * res[1] * 16 and e["multi_gpu"]["rccl_saw_all_ranks"] is True, e["multi_gpu"]
    assert extra["Kitchen"]["kernel"].startswith("lrd::megapool_kernel<5")  # the wavefront camera pass of the pool family
