#!/usr/bin/env python3
"""bench.py — Msamples/s of the MI355X megakernel path tracer on BASELINE.json's configs.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

A "step" is one full pass of the hot path over one frame: every pixel of the workload's
resolution x its spp through the megakernel (film accumulate included), scene already resident
in HBM.  N = 1 runs configs[1] ("Contemporary Bathroom"-class: 1024 x 1024, 1024 spp, depth 16;
the real asset is an external download, so the seeded procedural stand-in of SURVEY §8(d) is
used).  For N > 1 the same frame is sharded by screen tile over the ranks (strong scaling) and
the float4 film is sum-reduced to rank 0 over RCCL inside the timed region.

Rank 0 prints ONE JSON line with the driver's contract plus
  "roofline":     against the 8 TB/s HBM3E peak.  `traffic` = HBM bytes per launch from the PMC counters (FETCH_SIZE x 2 +
                  WRITE_SIZE, MI355X_MICROARCH.md), collected LIVE by this run: rocprofv3 --pmc passes of this same script on
                  the same workload AT THE TIMED SPP (round 5; round 4 measured 64 spp and scaled);
                  `achieved` = traffic / kernel duration and `frac` = achieved / peak -- the MEASURED HBM fraction (round 1
                  reported the algorithmic figure here, which exceeds the peak for this kernel: most of the canonical-BVH2
                  bytes are served by L2 / LDS).  The algorithmic figure of SURVEY 8(d) stays, under its own name:
                  `algorithmic_gbps` = algorithmic bytes per launch / kernel duration.  `valu` holds the third PMC pass
                  (wave-level VALU instructions) against the issue rates calibrated on the box by tools/valu_peak.hip.
  "cpu_baseline": the CPU oracle ("port": reference-faithful restatement, not the reference binary; it is pinned to the
                  reference's own code, tests/test_oracle_vs_ref.py) timed on the host cores on a bounded sample of the workload.
  "parity":       the frame the CPU leg just rendered (the oracle, pinned to the reference's own code) against the SAME samples
                  rendered by the shipped kernel, outside the timed region: rel-L1, per-pixel RMSE, mean bias, a FLIP-class error.
  "rays_per_s", "mean_path_length": from the device counters of a short counting render of the same frame.
  "extra_configs": short runs of the other BASELINE configs in the same line -- C1 (with its full CPU leg = configs[0], its parity and
                  "cpu_reference": the reference's OWN MegaPath code, oracle/_ref, on one host thread), C3, C4, C5 -- and C2 with the
                  sampler + filter the reference's converter writes for the README scenes (a low-discrepancy sampler + Gaussian r = 1).
  N > 1 adds "multi_gpu": the collective's own time, per-rank kernel times, and a check of the reduced film against a 1-GPU render.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (description, resolution, spp, depth)
    "c2": ("Contemporary Bathroom-class (procedural stand-in, ~600k tris), 1024x1024, 1024spp, depth 16", (1024, 1024), 1024, 16),
    "c1": ("Cornell Box, 512x512, 64spp, depth 8", (512, 512), 64, 8),
    # parity-test configurations; benchable on request (--workload), never the default line
    "c3": ("Bedroom-class (procedural stand-in: window openings, 10% glass, image environment), 1280x720, 4096spp, depth 16", (1280, 720), 4096, 16),
    "c4": ("Camera-class (procedural stand-in: ~1M tris, Disney/Plastic/Matte on 8 2k images, thin lens, env), 3840x2160, 1024spp, depth 16", (3840, 2160), 1024, 16),
    "c5": ("Kitchen-class (procedural stand-in: full surface closure set), 1280x720, 65536spp, depth 16", (1280, 720), 65536, 16),
}
# C5's 60.4 G samples take minutes per frame; throughput of the Independent sampler is spp-invariant, so the bench
# times this many spp of the same frame unless --spp says otherwise and says so in config.note
BENCH_SPP_CAP = {"c5": 2048}
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Algorithmic bytes per sample (SURVEY §8d formula on the oracle's canonical-BVH2 counters).  Measured live by
# the cpu_baseline leg at N = 1 (and reported from that measurement); at N > 1 no oracle runs, so the N = 1
# value of the same seeded workload is used (profiles/archive/r01_bench_c2_1gpu.json).
# (all five measured by the cpu_baseline leg of profiles/archive/r01d_bench_c*_1gpu.json)
ALGORITHMIC_BYTES_PER_SAMPLE = {"c2": 15104.1, "c1": 3909.0, "c3": 11125.0, "c4": 8323.0, "c5": 13974.0}
# VALU issue rates calibrated on the box (tools/valu_peak.hip -> profiles/archive/r02_valu_peak.json): cycles a SIMD needs per wave64
# instruction with >= 2 waves resident: v_fma_f32 / v_add_u32 2.3-2.6, v_max_f32 / v_cvt_f32_ubyte / v_pk_fma_f32 4.1-4.3
VALU_CYCLES_PER_WAVE_INSTR = (2.4, 4.2)
# ... and the kernel's own mix priced with that table: tools/isa_census.py on the traversal loop OF THE KERNEL THAT RUNS (round 5: the pool
# kernel <4096>; where two thirds of the instructions are issued) gives 828 issue cycles for 253 VALU instructions
# (profiles/r05_isa_census_pool.txt) = 3.27 cycles per instruction -- the figure round 3's census of the one-path kernel gave, too (807 / 247)
VALU_CYCLES_PER_WAVE_INSTR_MIX = 3.27
# Round 6: the mix is priced PER KERNEL from what ran: a third PMC pass counts the wave-level VALU instructions by class (SQ_INSTS_VALU_ADD_F32 /
# MUL_F32 / FMA_F32 -> 2.4 cycles, TRANS_F32 -> 8.2, CVT -> 4.2); what the hardware does not classify further (integer, compare, select, min / max,
# moves) is priced at the mean of those opcodes in the pool kernel's census: (53 full x 2.4 + 120 half x 4.2 + 10 v_cndmask_e32 x 2.0) / 183 = 3.55
# (profiles/r06_isa_census_pool.txt).  The static 3.27 above stays as the fallback where the class counters are missing.
VALU_CLASS_CYCLES = {"SQ_INSTS_VALU_ADD_F32": 2.4, "SQ_INSTS_VALU_MUL_F32": 2.4, "SQ_INSTS_VALU_FMA_F32": 2.4, "SQ_INSTS_VALU_TRANS_F32": 8.2, "SQ_INSTS_VALU_CVT": 4.2}
VALU_OTHER_CYCLES = 3.55
# the PMC passes of one run (rocprofv3 --pmc, counters only + --kernel-trace): SQ holds 8 counters per pass, the TCC 4 (FETCH_SIZE takes 3, WRITE_SIZE 2:
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), so the two byte counters ride in different passes and the SQ counters fill the rest
PMC_PASSES = (
    ("fetch", ["FETCH_SIZE", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "SQ_INSTS_SALU"]),
    ("write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INST_LEVEL_VMEM", "SQ_INSTS_LDS", "SQ_INST_LEVEL_LDS", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_VALU"]),
    ("mix", ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64"]),
)
# what a pass falls back to when the box refuses the combination (one block per pass, as round 5 ran them)
PMC_FALLBACK = {"fetch": [["FETCH_SIZE"], ["SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVES", "SQ_INSTS_SALU"]],
                "write": [["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INST_LEVEL_VMEM", "SQ_INSTS_LDS", "SQ_INST_LEVEL_LDS", "SQ_INSTS_SMEM"]]}
# spp of the extra configurations' counter passes (their frames are timed at the configurations' own spp; three more passes of those would
# take minutes): per-sample counters of the same frame at fewer samples per pixel, stated in every block's traffic_source
PMC_SPP = {"c1": 64, "c3": 512, "c4": 128, "c5": 512, "c2": 1024, "c2:PaddedSobol": 512}
SHADER_CLOCK_HZ = 2.4e9
L2_PEAK_GBPS = 34500.0  # aggregate L2 bandwidth, MI355X_MICROARCH.md (4 MiB per XCD, 32 MiB aggregate, ~34.5 TB/s)
METRIC = "Msamples/s (+ fraction of HBM roofline) at fixed SPP, 1/2/4/8 GPU"


SCENE_FILES = {}  # (workload, sampler, spp) -> the generated scene description (the PMC child processes load it instead of generating it again)


def build_scene(workload: str, tmpdir: str, spp_override: int | None, sampler: str = "Independent", scene_file: str | None = None):
    from luisarender_amd import Scene
    from luisarender_amd.scenes import cornell_box, generate_room_scene
    desc, res, spp, depth = WORKLOADS[workload]
    spp = spp_override or min(spp, BENCH_SPP_CAP.get(workload, spp))
    if scene_file:
        scene = Scene.load(scene_file)
    elif workload == "c1":
        scene = Scene.from_string(cornell_box(resolution=res[0], spp=spp, depth=depth, sampler=sampler))
    else:
        from luisarender_amd.scenes import generate_bedroom_scene, generate_camera_scene, generate_kitchen_scene
        gen = {"c2": generate_room_scene, "c3": generate_bedroom_scene, "c4": generate_camera_scene, "c5": generate_kitchen_scene}[workload]
        sub = os.path.join(tmpdir, f"{workload}_{sampler}_{spp}")  # (one directory per description: the generators write fixed file names)
        SCENE_FILES[(workload, sampler)] = gen(sub, resolution=res, spp=spp, depth=depth, sampler=sampler)
        scene = Scene.load(SCENE_FILES[(workload, sampler)])
    return scene, desc, res, spp


def cpu_baseline(scene, res, budget_s: float, full_spp: int | None = None):
    """Oracle on all host cores over a bounded sample: whole frame at 1..n spp until ~budget_s (full_spp: the whole configuration)."""
    from oracle.check import Oracle, algorithmic_bytes
    cores = os.cpu_count() or 1
    oracle = Oracle(scene)
    # calibrate on a strip, then size the sample
    t0 = time.perf_counter()
    _, c0 = oracle.render(0, 1, rect=(0, res[1] // 2 - 16, res[0], res[1] // 2 + 16), threads=cores)
    rate = c0["paths"] / max(time.perf_counter() - t0, 1e-6)
    spp = int(max(1, min(full_spp or 16, budget_s * rate / (res[0] * res[1]))))
    t0 = time.perf_counter()
    film, counters = oracle.render(0, spp, threads=cores)
    dt = time.perf_counter() - t0
    return {
        "value": counters["paths"] / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"CPU oracle (reference-faithful restatement, not the reference binary), full frame {res[0]}x{res[1]} at {spp} spp "
                  f"= {counters['paths']} paths in {dt:.1f} s",
    }, algorithmic_bytes(counters) / counters["paths"], counters, (oracle.convert(film), spp)


def device_parity(scene, local_rank, cpu_frame, spp, scheduler="auto"):
    """The frame the CPU leg just rendered against the SAME samples [0, spp) on the device, shipped kernel, outside any timed region
    (VERDICT r02 1c).  FLIP on the central 512 x 512 of larger frames (the numpy restatement takes seconds per megapixel)."""
    import numpy as np
    from luisarender_amd.render import MegaPathRenderer
    from oracle import image_metrics as M
    r = MegaPathRenderer(local_rank)
    if scheduler != "auto":
        r.set_scheduler(scheduler == "pool")
    r.upload(scene)
    r.render(0, spp, counters=False, sync=True)
    gpu = r.download(converted=True)
    variant = r.last_variant()
    r.close()
    out = M.summary(gpu[..., :3], cpu_frame[..., :3], with_flip=False)
    h, w = gpu.shape[:2]
    y0, x0 = max(0, (h - 512) // 2), max(0, (w - 512) // 2)
    out["flip"] = M.flip(gpu[y0:y0 + 512, x0:x0 + 512, :3], cpu_frame[y0:y0 + 512, x0:x0 + 512, :3])
    # ... and against the oracle on the DEVICE'S OWN GEOMETRY (oracle_bvh.h baked-geometry mode: the fp32 world-space triangles the host bakes,
    # not the reference's object-space ones), central window, same samples: what is left is the kernel's arithmetic alone
    try:
        from oracle.check import Oracle
        rect = (x0, y0, min(x0 + 512, w), min(y0 + 512, h))
        baked = Oracle(scene, bake_instances=True)
        sub, _ = baked.render(0, spp, rect=rect)
        ref = baked.convert(sub)[rect[1]:rect[3], rect[0]:rect[2], :3]
        dev = gpu[rect[1]:rect[3], rect[0]:rect[2], :3]
        obj = cpu_frame[rect[1]:rect[3], rect[0]:rect[2], :3]
        out["parity_same_geometry"] = {"rel_l1": float(np.abs(dev - ref).sum() / max(np.abs(ref).sum(), 1e-30)),
                                       "bias": float((dev.mean() - ref.mean()) / max(ref.mean(), 1e-30)), "window": list(rect), "spp": spp,
                                       # the scene's own sensitivity: the oracle against itself across the two geometry modes (same algorithm, same samples,
                                       # hit points that differ in their last bit) -- the device cannot be closer to either than they are to each other
                                       "oracle_vs_oracle_rel_l1": float(np.abs(ref - obj).sum() / max(np.abs(obj).sum(), 1e-30)),
                                       "device_vs_object_space_oracle_rel_l1_same_window": float(np.abs(dev - obj).sum() / max(np.abs(obj).sum(), 1e-30)),
                                       "against": "the CPU oracle intersecting the fp32 world-space triangles the host bakes for the device (test infrastructure), same samples"}
    except Exception as e:  # noqa: BLE001
        out["parity_same_geometry"] = {"error": str(e)}
    out.update({"samples": int(h * w * spp), "spp": spp, "kernel": kernel_name(variant), "finite": bool(np.isfinite(gpu).all()),
                "against": "the CPU oracle's frame of the same samples (the oracle is bit-equal to the reference's own code: tests/test_oracle_vs_ref.py, "
                           "tests/test_ref_golden.py); rmse = per-pixel L2 of the converted linear RGB, flip = LDR-FLIP restated in oracle/image_metrics.py "
                           "(clip + sRGB), on the central 512 x 512"})
    return out


def path_statistics(scene, local_rank, spp=64, scheduler="auto"):
    """rays per sample, mean path length and lane use from the device counters of a counting render (the COUNT twin of the kernel) of the
    frame AT THE SPP THAT IS TIMED (round 5: lane use and the drain of a launch depend on spp; round 4 measured them at 64)"""
    from luisarender_amd.render import MegaPathRenderer
    r = MegaPathRenderer(local_rank)
    if scheduler != "auto":
        r.set_scheduler(scheduler == "pool")
    r.upload(scene)
    r.render(0, spp, counters=True, sync=True)
    c = r.counters()
    r_variant = r.last_variant()
    r.close()
    paths = max(c["paths"], 1)
    # what the kernel itself asks the memory system for, per sample: 64 B per BVH packet, 48 B per triangle test, one 128-byte shading
    # record per surface hit, and per light sample the light's triangle (alias entry 8 B + shading-point gather 128 B)
    requested = (64.0 * c["nodes_visited"] + 48.0 * c["tris_tested"] + 128.0 * c["surface_hits"] + 136.0 * c["nee_samples"]) / paths
    pool_state = 0.0
    if r_variant & 4096:  # a pool kernel: every shaded vertex reads and writes three 16-byte quads of its context's record, a path one more at its start and its end
        pool_state = (96.0 * c["closest_rays"] + 32.0 * c["paths"]) / paths
        requested += pool_state
    wave_cycles = max(c["wave_cycles"], 1)
    return {"pool_state_bytes_per_sample": pool_state, "wave_life": {"traversal_loop": c["trace_cycles"] / wave_cycles, "shading_block": c["shade_cycles"] / wave_cycles},
            "rays_per_sample": (c["closest_rays"] + c["shadow_rays"]) / paths, "closest_rays_per_sample": c["closest_rays"] / paths,
            "shadow_rays_per_sample": c["shadow_rays"] / paths, "mean_path_length": c["path_length_sum"] / paths,
            "nodes_per_ray": c["nodes_visited"] / max(c["closest_rays"] + c["shadow_rays"], 1), "spp": spp,
            "nodes_per_sample": c["nodes_visited"] / paths, "tris_per_sample": c["tris_tested"] / paths,
            "surface_hits_per_sample": c["surface_hits"] / paths, "nee_samples_per_sample": c["nee_samples"] / paths,
            "requested_bytes_per_sample": requested,
            "lanes": {"trace": c["trace_steps_busy"] / max(c["trace_steps"], 1), "trace_starved": c["trace_steps_starved"] / max(c["trace_steps"], 1),
                      "shade": c["shade_busy"] / max(c["shade_calls"], 1),
                      "note": "lane utilisation from the counting twin of the kernel at %d spp: trace = lane-steps of the traversal loop with a ray in flight / all lane-steps; "
                              "shade = lanes with a hit to shade / lanes of a shading batch; trace_starved = lane-steps idle because the work item had no sample left" % spp}}


def kernel_name(variant: int) -> str:
    """symbol of the megakernel variant lrhip_last_variant names (include/lrhip.h LRHIP_FEAT_*): the pool kernels are another template"""
    return f"lrd::megapool_kernel<{variant}u>" if variant & 4096 else f"lrd::megapath_kernel<{variant}u>"


def source_hash() -> str:
    """identifies the build a profile belongs to: the device + BVH-builder sources (there is no .git on the GPU box)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "luisarender_amd", "csrc", "hip", "*")) + [os.path.join(ROOT, "luisarender_amd", "csrc", "host", "accel.cpp"), os.path.join(ROOT, "Makefile")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def reference_leg(workload, res, spp, sampler, budget_s):
    """The REFERENCE'S OWN MegaPath::Li (oracle/_ref = /root/reference/src compiled in place on the scalar LuisaCompute stand-in) on the bench's
    own description of `workload` -- C2 in its InlineMesh form: the same 600 k instanced triangles, transforms and materials, the form the
    reference's parser loads without an OBJ importer -- a pixel grid of the frame for a bounded time, ONE PROCESS PER HOST THREAD (the shim's
    dispatch is serial; libref.so brings its own operator new, hence child processes).  None where oracle/_ref was not built."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref.so")):
        return None
    procs = max(1, min(os.cpu_count() or 1, 64))
    if workload == "c1":
        procs = 1  # (configs[0] keeps round 5's one-thread figure: comparable across rounds)
        make = "from luisarender_amd.scenes import cornell_box; text = cornell_box(resolution=%d, spp=%d, depth=%d, sampler=%r)" % (res[0], spp, WORKLOADS[workload][3], sampler)
    else:
        make = ("import tempfile; from luisarender_amd.scenes import generate_room_scene; d = tempfile.mkdtemp(prefix='lr_ref_'); "
                "text = open(generate_room_scene(d, resolution=(%d, %d), spp=%d, depth=%d, sampler=%r, inline_meshes=True)).read()" % (res[0], res[1], spp, WORKLOADS[workload][3], sampler))
    code = ("import json, sys; sys.path.insert(0, %r); from oracle.check import reference_rate; %s; "
            "print(json.dumps(reference_rate(text, %f, first_sample=int(sys.argv[1]))))" % (ROOT, make, budget_s))
    children = [subprocess.Popen([sys.executable, "-c", code, str(i * 4096)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(procs)]
    results = []
    for c in children:
        try:
            text, _ = c.communicate(timeout=budget_s + 120.0)
            results.append(json.loads(text.strip().splitlines()[-1]))
        except Exception:  # noqa: BLE001
            c.kill()
    results = [r for r in results if r]
    if not results:
        return None
    out = dict(results[0])
    out["value"] = sum(r["value"] for r in results)
    out["cores"] = len(results)
    out["sample"] = results[0]["sample"].replace(", 1 thread", f", in each of {len(results)} processes (one per host thread, disjoint sample numbers); value = their sum")
    return out


def kernel_names_of(variant: int):
    """substrings of the kernel symbols a configuration's frame launches: the megakernel of `variant` (lrhip_last_variant), and in wavefront
    mode (the mask holds closure bits the lean kernel does not: lrhip.hip) camera pass + continuation pass + the heavy-closure kernels"""
    if variant & 1024:  # kFeatWf
        camera = variant & ~(16 | 32 | 64 | 512)
        return [f"_kernel<{camera}u>", f"_kernel<{camera | 2048}u>", "heavy_kernel<"]
    return [f"_kernel<{variant}u>"]


def pmc_child(spec: str, local_rank: int):
    """--pmc-child: renders every configuration of `spec` once (workload:spp:sampler:scheduler:scene file, comma separated) -- the process the
    counter passes of live_pmc profile.  Nothing is timed or printed."""
    from luisarender_amd.render import MegaPathRenderer
    with tempfile.TemporaryDirectory(prefix="lr_pmc_child_") as tmp:
        for item in spec.split(","):
            workload, spp, sampler, scheduler, scene_file = item.split(":", 4)
            scene, _, _, _ = build_scene(workload, tmp, None, sampler, scene_file or None)
            r = MegaPathRenderer(local_rank)
            if scheduler != "auto":
                r.set_scheduler(scheduler == "pool")
            r.upload(scene)
            r.render(0, int(spp), sync=True)
            r.close()


def live_pmc(configs, timeout: float = 420.0):
    """HBM traffic, VALU instructions by class, wait / issue cycles and memory-instruction levels per sample of every configuration's kernels,
    measured NOW: rocprofv3 --pmc passes (counters only, with --kernel-trace, as the pool allows) of this script in --pmc-child mode, which
    renders each configuration once at its counter spp.  configs: [{"key", "workload", "spp", "sampler", "variant", "scheduler"}].
    Returns {key: {...}} (a key is missing where nothing of its kernels was seen)."""
    import shutil
    import sqlite3
    import subprocess
    if shutil.which("rocprofv3") is None:
        return {}
    # already under a profiler (someone runs `rocprofv3 ... -- python bench.py`): no nested passes
    if any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return {}
    spec = ",".join(f"{c['workload']}:{c['spp']}:{c['sampler']}:{c.get('scheduler', 'auto')}:{SCENE_FILES.get((c['workload'], c['sampler']), '')}" for c in configs)
    totals = {}  # kernel name -> counter -> total over its dispatches
    errors = []
    with tempfile.TemporaryDirectory(prefix="lr_pmc_") as d:
        env = dict(os.environ, TMPDIR="/tmp")

        def one_pass(name, pmc):
            cmd = ["rocprofv3", "--pmc", *pmc, "--kernel-trace", "-d", os.path.join(d, name), "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), "--pmc-child", spec]
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(os.path.join(d, name)) for f in fs if f.endswith(".db")]
            db = sqlite3.connect(dbs[0])
            launches = dict(db.execute("select name, count(*) from kernels group by name"))
            for kname, ns in db.execute("select name, sum(end - start) from kernels group by name"):
                totals.setdefault(kname, {})["kernel_ns:" + name.split("_")[0]] = float(ns)  # (this pass's own kernel time: what its cycle counters are read against)
            seen = 0
            # (one row per dispatch and counter, as round 5 read them: the mean over a kernel's dispatches x its launches = the frame's total)
            for kname, cname, mean in db.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
                totals.setdefault(kname, {})[cname] = mean * launches.get(kname, 1)
                seen += 1
            if seen == 0:
                raise RuntimeError("no counter rows")

        for name, pmc in PMC_PASSES:
            try:
                one_pass(name, pmc)
            except Exception as e:  # a refused combination / no profiler on this box: the smaller passes of round 5, then report what was measured
                errors.append(f"{name}: {type(e).__name__}")
                for i, part in enumerate(PMC_FALLBACK.get(name, [])):
                    try:
                        one_pass(f"{name}_{i}", part)
                    except Exception as e2:  # noqa: BLE001
                        errors.append(f"{name}_{i}: {type(e2).__name__}")
    out = {}
    for c in configs:
        desc, res, _, _ = WORKLOADS[c["workload"]]
        samples = res[0] * res[1] * c["spp"]
        names = kernel_names_of(c["variant"])
        counters, per_kernel = {}, {}
        for kname, cs in totals.items():
            if any(n in kname for n in names):
                short = kname.split("(")[0].replace("void ", "")
                per_kernel[short] = cs
                for k, v in cs.items():
                    counters[k] = counters.get(k, 0.0) + v
        if not counters:
            continue
        o = {"spp": c["spp"], "samples_per_launch": samples, "tool": "rocprofv3 --pmc (separate passes, one child process per pass renders every configuration once) --kernel-trace",
             "counters": counters, "kernels": sorted(per_kernel)}
        if errors:
            o["errors"] = errors
        if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
            # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE tallies a 128-byte request as 64 (MI355X_MICROARCH.md, HBM section)
            o["hbm_bytes_per_sample"] = (counters["FETCH_SIZE"] * 2048.0 + counters["WRITE_SIZE"] * 1024.0) / samples
            o["hbm_read_bytes_per_sample"] = counters["FETCH_SIZE"] * 2048.0 / samples
            o["hbm_write_bytes_per_sample"] = counters["WRITE_SIZE"] * 1024.0 / samples
        if "SQ_INSTS_VALU" in counters:
            o["valu_wave_instr_per_sample"] = counters["SQ_INSTS_VALU"] / samples
            classes = {k: counters[k] for k in VALU_CLASS_CYCLES if k in counters}
            if len(classes) == len(VALU_CLASS_CYCLES):
                other = max(counters["SQ_INSTS_VALU"] - sum(classes.values()), 0.0)
                o["cycles_per_wave_instr"] = (sum(VALU_CLASS_CYCLES[k] * v for k, v in classes.items()) + VALU_OTHER_CYCLES * other) / counters["SQ_INSTS_VALU"]
                o["valu_mix"] = {**{k.replace("SQ_INSTS_VALU_", "").lower(): v / counters["SQ_INSTS_VALU"] for k, v in classes.items()}, "other": other / counters["SQ_INSTS_VALU"]}
        out[c["key"]] = o
    return out


def roofline_block(workload, variant, kernel_ms, launch_samples, bytes_per_sample, pmc, pmc_source, stats, simds):
    """the `roofline` object of one configuration: measured memory traffic against the HBM peak (the contract's achieved / peak / frac / traffic),
    and beside it what actually bounds these kernels -- VALU issue priced per kernel, the waves' wait / issue-stall / active split, the L2"""
    seconds = kernel_ms * 1e-3
    algorithmic_gbps = bytes_per_sample * launch_samples / seconds / 1e9
    traffic = pmc["hbm_bytes_per_sample"] * launch_samples if pmc and pmc.get("hbm_bytes_per_sample") else None
    achieved = traffic / seconds / 1e9 if traffic else None
    r = {
        # What bounds these kernels is neither roof of the contract's vocabulary: a wave issues one instruction of any kind every ~4.5 cycles and
        # waits for two dependent gathers per traversal iteration; at four waves per SIMD (128 VGPRs, 39 KB of LDS per block) the SIMD is neither
        # full nor starved (the stall probe's tables: profiles/r06f_stalls*.txt).  The contract's HBM figures stay: `achieved` /
        # `frac` / `traffic` are the MEASURED memory-side traffic over the kernel's duration (FETCH_SIZE counts Infinity-Cache hits too: an
        # upper bound on HBM traffic); one number per roof follows in `valu`, `waves`, `lanes`, `l2`.
        "bound": "issue + latency at 4 waves per SIMD", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS if achieved else None, "traffic": traffic, "traffic_source": pmc_source,
        "kernel": kernel_name(variant) if not variant & 1024 else " + ".join(kernel_names_of(variant)), "kernel_ms": kernel_ms,
        "algorithmic_bytes_per_sample": bytes_per_sample, "algorithmic_gbps": algorithmic_gbps,
        "algorithmic_frac_of_hbm_peak": algorithmic_gbps / HBM_PEAK_GBPS,
        "note": "achieved / frac = MEASURED memory traffic (PMC FETCH_SIZE x 2 + WRITE_SIZE, Infinity-Cache hits included) per sample x the timed launch's samples over "
                "its HIP-event duration; algorithmic_* = the canonical-BVH2 byte count of SURVEY 8(d), most of which L1 / L2 / LDS serve on chip (it may exceed "
                "the HBM peak and is not a traffic figure); valu.issue_frac = priced VALU issue cycles over the launch's SIMD cycles, waves = where a wave's cycles go, "
                "lanes = how many of the issued lanes did useful work, l2 = the kernel's own requests against the L2's bandwidth",
    }
    counters = (pmc or {}).get("counters") or {}
    if counters:
        # the raw counter totals the figures of this block are made of, for the counter passes' own frame (pmc.samples paths): with a calculator,
        # traffic = (FETCH_SIZE x 2048 + WRITE_SIZE x 1024) / pmc.samples x the timed launch's samples; frac = traffic / kernel_ms / 8 TB/s
        r["pmc"] = {"spp": pmc.get("spp"), "samples": pmc.get("samples_per_launch"), "timed_launch_samples": launch_samples, "kernels": pmc.get("kernels"),
                    "counters": {k: v for k, v in sorted(counters.items())}}
    if pmc and pmc.get("valu_wave_instr_per_sample"):
        instr = pmc["valu_wave_instr_per_sample"] * launch_samples
        simd_cycles = simds * seconds * SHADER_CLOCK_HZ
        price = pmc.get("cycles_per_wave_instr") or VALU_CYCLES_PER_WAVE_INSTR_MIX
        r["valu"] = {
            "issue_frac": instr * price / simd_cycles, "wave_instr_per_sample": pmc["valu_wave_instr_per_sample"], "wave_instr_per_launch": instr,
            "cycles_per_wave_instr": price, "valu_mix": pmc.get("valu_mix"), "simds": simds, "shader_clock_hz": SHADER_CLOCK_HZ, "simd_cycles_per_launch": simd_cycles,
            "calibration": "issue_frac = wave_instr_per_launch x cycles_per_wave_instr / simd_cycles_per_launch.  cycles_per_wave_instr = the DYNAMIC class mix of the kernel(s) that ran "
                           "(PMC: SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 x 2.4, TRANS_F32 x 8.2, CVT x 4.2, everything else x 3.55 = the mean price of the unclassified opcodes in the pool "
                           "kernel's census) with the issue costs measured on the box (profiles/archive/r04h_cndmask_forms.json, tools/valu_peak2.hip)"
                           if pmc.get("cycles_per_wave_instr") else "static mix of the pool kernel's traversal loop (profiles/r05_isa_census_pool.txt): the class counters were not collected",
            # what a change of the instruction count buys, MEASURED (round 5, profiles/r05b_sensitivity_probes.txt, r05c_pmc_*.json): 9.3 % more VALU instructions
            # (32 dependent v_fma per node step) cost 4.3 % time, 6.3 % fewer gained 2.4 %, three waves per SIMD instead of four cost 16 %
            "time_elasticity_to_valu_instructions": 0.42,
            "elasticity_source": "profiles/r05b_sensitivity_probes.txt (pn32: +9.3 % instructions, +4.3 % time), profiles/r05c_pmc_{r04,base}.json (-6.3 %, -2.4 %); measured on C2's pool kernel",
        }
    if counters.get("SQ_WAVE_CYCLES"):
        wc = counters["SQ_WAVE_CYCLES"]
        # MI355X_MICROARCH.md: WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES, disjoint
        r["waves"] = {k: counters[c] / wc for k, c in (("waiting_at_waitcnt", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"), ("instruction_in_flight", "SQ_ACTIVE_INST_ANY")) if c in counters}
        if counters.get("SQ_BUSY_CYCLES") and counters.get("kernel_ns:fetch"):
            r["waves"]["effective_clock_ghz"] = counters["SQ_BUSY_CYCLES"] / 32.0 / counters["kernel_ns:fetch"]  # (busy cycles of the 32 shader engines over the counter pass's own kernel time)
        # (SQ_INST_LEVEL_VMEM / SQ_INST_LEVEL_LDS are collected and kept in `counters`, but their ratio to the instruction counts comes out at ~37 and ~3
        # "cycles" -- not a gather's latency in any unit the guide documents -- so no latency is derived from them; the stall probe times the waits)
    if counters.get("TCC_HIT_sum") is not None and counters.get("TCC_MISS_sum"):
        r["l2_hit_rate"] = counters["TCC_HIT_sum"] / (counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"])
    if stats:
        r["lanes"] = stats["lanes"]
        r["wave_life"] = stats.get("wave_life")
        req_gbps = stats["requested_bytes_per_sample"] * launch_samples / seconds / 1e9
        r["l2"] = {
            "frac": req_gbps / L2_PEAK_GBPS, "requested_gbps": req_gbps, "peak": L2_PEAK_GBPS, "unit": "GB/s",
            "requested_bytes_per_sample": stats["requested_bytes_per_sample"], "pool_state_bytes_per_sample": stats.get("pool_state_bytes_per_sample"),
            "note": "the kernel's own requests per sample from its counters (path_statistics: 64 B x nodes + 48 B x triangle tests + 128 B x surface hits "
                    "+ 136 B x light samples + the pool kernels' path state: 96 B per shaded vertex, 32 B per path) x samples / kernel time, against the aggregate L2 bandwidth "
                    "(MI355X_MICROARCH.md); the vector L1 serves part of them (the BVH's top levels), so this is an upper bound on L2 traffic; texel fetches are not in it",
        }
    return r


def run_workload(workload, args, rank, world, local_rank, tmp, steps, warmup, spp_override=None, sampler="Independent", warmup_spp=None):
    """times `steps` frames of one workload -> (value Msamples/s, ms_per_step, kernel_ms, variant, scene, res, spp, desc)"""
    import numpy as np
    import torch
    import torch.distributed as dist
    from luisarender_amd.parallel import FilmReducer
    from luisarender_amd.render import MegaPathRenderer
    scene, desc, res, spp = build_scene(workload, tmp, spp_override, sampler)
    renderer = MegaPathRenderer(local_rank)  # no CPU fallback: raises if the HIP library / GPU is missing
    if getattr(args, "scheduler", "auto") != "auto":  # A/B and profiling runs only; the line says so (config.scheduler_override)
        renderer.set_scheduler(args.scheduler == "pool")
    renderer.upload(scene)
    film = torch.zeros((res[1], res[0], 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    renderer.bind_film(film.data_ptr())
    # communicator through the C ABI: the timed collective is the product's lrhip_film_reduce.  LR_BENCH_FORCE_COLLECTIVE=1 runs
    # the same code with one rank (1-GPU boxes).  Should the C-ABI communicator fail on a node this was never run on, the same
    # RCCL reduce goes through torch.distributed instead and the line says so (config.collective) -- a bench that dies has no line.
    # ALL ranks take the same path: the outcome of the attempt is all-reduced (MIN) over the process group the launcher made, so a
    # rank whose ncclCommInitRank failed cannot sit in dist.reduce while the others sit in ncclReduce.
    force = os.environ.get("LR_BENCH_FORCE_COLLECTIVE") == "1" and dist.is_initialized()
    collective = "none (single GPU)"
    reducer = None
    if world > 1 or force:
        error = ""
        try:
            reducer = FilmReducer(renderer, rank, world, force=force)
        except Exception as e:  # noqa: BLE001
            error = str(e)
        ok = torch.tensor([0 if reducer is None or reducer.comm is None else 1], dtype=torch.int32, device=f"cuda:{local_rank}")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok[0]) == 1:
            collective = "lrhip_film_reduce (ncclReduce through the C ABI, on the render stream)"
        else:
            if reducer is not None:
                reducer.close()
            reducer = None
            collective = f"torch.distributed.reduce (the C-ABI communicator failed on at least one rank{': ' + error if error else ''})"
    run_workload.collective = collective
    torch.cuda.synchronize()
    reduce_ms = []

    def step(n=None):
        film.zero_()
        torch.cuda.synchronize()  # film clear is on torch's stream, the megakernel on the context's stream
        renderer.render(0, n or spp, rank=rank, world=world, balance_shards=world)
        if reducer is not None or world > 1:
            renderer.synchronize()  # (one host round trip per step: it separates the collective's time from the kernel's)
            t = time.perf_counter()
            if reducer is not None:
                reducer.reduce(0)  # ncclReduce on the renderer's stream, behind the megakernel
                renderer.synchronize()
            else:
                dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
                torch.cuda.synchronize()
            reduce_ms.append((time.perf_counter() - t) * 1e3)
        else:
            renderer.synchronize()
        return renderer.last_render_ms()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step(warmup_spp)  # (warmup_spp: the extra configurations warm up -- allocations, code load -- on a few samples of their frame)
    reduce_ms.clear()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = [step() for _ in range(steps)]
    barrier()
    elapsed = time.perf_counter() - t0
    mean_kernel_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
    run_workload.multi_gpu = None
    if world > 1 or force:
        mean_reduce_ms = sum(reduce_ms) / max(len(reduce_ms), 1)
        t = torch.tensor([elapsed, mean_kernel_ms, -mean_kernel_ms, mean_reduce_ms], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, mean_kernel_ms = float(t[0]), float(t[1])
        # what every rank actually saw -- its device, and the size / rank of ITS communicator as RCCL reports them (lrhip_comm_info) -- gathered to
        # rank 0: the line itself says whether the collective spanned the N ranks the launcher started
        mine = reducer.info() if reducer is not None else None
        seen = torch.tensor([rank, local_rank, torch.cuda.current_device(), mine["ranks"] if mine else -1, mine["rank"] if mine else -1,
                             mine["device"] if mine else -1, dist.get_world_size()], dtype=torch.int64, device=f"cuda:{local_rank}")
        gathered = [torch.zeros_like(seen) for _ in range(dist.get_world_size())]
        dist.all_gather(gathered, seen)
        ranks_seen = [dict(zip(("rank", "local_rank", "hip_device", "comm_ranks", "comm_rank", "comm_device", "process_group_size"), (int(x) for x in g.tolist()))) for g in gathered]
        info = {"kernel_ms_max_over_ranks": float(t[1]), "kernel_ms_min_over_ranks": -float(t[2]), "reduce_ms": float(t[3]),
                "ranks": ranks_seen, "rccl_saw_all_ranks": all(r["comm_ranks"] == world for r in ranks_seen) if reducer is not None else None,
                "distinct_devices": len({r["hip_device"] for r in ranks_seen}),
                "reduce_bytes": res[0] * res[1] * 16, "note": "reduce_ms = host time from the end of the slowest rank's megakernel to the end of the collective (waiting for stragglers included)"}
        # The reduced film must BE the frame: rank 0 renders the first 8 tile rows (64 pixel rows, every rank owns tiles there) on its
        # own with the same balance_shards and compares them bit for bit with what the collective delivered.  A wrong collective
        # fails the line instead of only slowing it.
        if rank == 0:
            reduced = film.cpu().numpy().copy()
            check = torch.zeros_like(film)
            renderer.bind_film(check.data_ptr())
            tiles_x = (res[0] + 7) // 8
            rows = min(8, (res[1] + 7) // 8)
            p_world, p_rank = 1, 0
            renderer.render(0, spp, rank=p_rank, world=p_world, balance_shards=world, sync=True, tile_end=rows * tiles_x)
            mine = check.cpu().numpy()
            band = slice(0, min(rows * 8, res[1]))
            info["reduced_film_equals_1gpu_render"] = bool(np.array_equal(reduced[band], mine[band]))
            info["checked_pixel_rows"] = int(band.stop)
            info["checked_sample_counts_ok"] = bool((reduced[..., 3] == float(spp)).all())
            renderer.bind_film(film.data_ptr())
        run_workload.multi_gpu = info
    variant = renderer.last_variant()
    if reducer is not None:
        reducer.close()
    renderer.close()
    value = res[0] * res[1] * spp * steps / elapsed / 1e6
    return value, elapsed / steps * 1e3, mean_kernel_ms, variant, scene, res, spp, desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=None, help="override the workload's spp (invalidates the headline number)")
    ap.add_argument("--sampler", default="Independent", choices=["Independent", "PaddedSobol", "Sobol"],
                    help="sampler of the timed frame (anything but Independent invalidates the headline number)")
    ap.add_argument("--scheduler", default="auto", choices=["auto", "legacy", "pool"],
                    help="force one kernel family (lrhip_set_scheduler) for A/B and profiling runs; anything but auto is recorded in config.scheduler_override")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (roofline.traffic from profiles/ or null)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other BASELINE configs")
    ap.add_argument("--extra-spp", type=int, default=None, help="N > 1: spp of the C4 / C5 entries instead of the configurations' own (tests)")
    ap.add_argument("--no-stats", action="store_true", help="skip the short counting render behind rays_per_s / mean_path_length (the PMC passes: ONE megakernel dispatch per process)")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)  # (the process live_pmc profiles: renders the given configurations once)
    args = ap.parse_args()
    if args.pmc_child:
        pmc_child(args.pmc_child, int(os.environ.get("LOCAL_RANK", "0")))
        return

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    if world > 1 or (os.environ.get("LR_BENCH_FORCE_COLLECTIVE") == "1" and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    with tempfile.TemporaryDirectory(prefix="lr_bench_") as tmp:
        value, ms_per_step, mean_kernel_ms, variant, scene, res, spp, desc = run_workload(
            args.workload, args, rank, world, local_rank, tmp, args.steps, args.warmup, args.spp, args.sampler)
        elapsed = ms_per_step * args.steps * 1e-3
        main_multi, main_collective = getattr(run_workload, "multi_gpu", None), getattr(run_workload, "collective", None)

        # The configurations BASELINE.json names FOR SEVERAL GPUs, in the same line of every N > 1 run (round 5: the driver's first scaling run
        # must produce them): C4 = "Camera scene, 3840x2160, 1024spp, screen-tile shard + RCCL film reduce" and C5 (kitchen class, wavefront
        # mode, 2048 of its 65 536 spp), each sharded over the N ranks like the headline frame, reduced by the same lrhip_film_reduce, with
        # its own multi_gpu record (reduce_ms, what RCCL saw, the reduced film against a 1-GPU render of the same tiles).  Every rank takes part.
        multi_extra = []
        forced = os.environ.get("LR_BENCH_FORCE_COLLECTIVE") == "1" and dist.is_initialized()
        if (world > 1 or forced) and not args.no_extra:
            for w, spp_cfg in (("c4", None), ("c5", BENCH_SPP_CAP["c5"])):
                v, ms, kms, var, sc, r, sp, d = run_workload(w, args, rank, world, local_rank, tmp, 1, 1, args.extra_spp or spp_cfg, "Independent",
                                                             warmup_spp=None if w == "c5" else min(8, args.extra_spp or 8))  # (C5: wavefront mode sizes its queues by the frame's spp)
                if rank == 0:
                    e = {"workload": d, "sampler": "Independent", "spp_timed": sp, "spp_config": WORKLOADS[w][2], "value": v, "unit": "Msamples/s", "steps": 1, "ms_per_step": ms,
                         "kernel_ms": kms, "kernel": kernel_name(var), "n_gpus": world, "collective": getattr(run_workload, "collective", None),
                         "multi_gpu": getattr(run_workload, "multi_gpu", None)}
                    if e["multi_gpu"] and e["multi_gpu"].get("reduced_film_equals_1gpu_render") is False:
                        e["value"], e["error"] = None, "the film the collective delivered differs from a 1-GPU render of the same tiles"
                    multi_extra.append(e)

        if rank == 0:
            samples_per_step = res[0] * res[1] * spp
            out = {
                "metric": METRIC, "value": value, "unit": "Msamples/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": desc, "integrator": "MegaPath", "sampler": "Independent (seed 19980810)" if args.sampler == "Independent" else args.sampler,
                           "resolution": list(res), "spp": spp, "parallelism": f"screen-tile shard x{world} + RCCL film reduce" if world > 1 else "single GPU",
                           "collective": main_collective,
                           "scheduler": "path pool, two contexts per lane (megapool_kernel.h)" if variant & 4096 else "one path per lane (megapath_kernel.h)"},
            }
            if args.scheduler != "auto":
                out["config"]["scheduler_override"] = args.scheduler
            if args.spp is not None or args.sampler != "Independent":
                out["config"]["note"] = "spp / sampler overridden: not the headline configuration"
            elif args.workload in BENCH_SPP_CAP:
                out["config"]["note"] = f"timed at {spp} spp of the same frame (Independent sampler: throughput is spp-invariant)"
            bytes_per_sample = ALGORITHMIC_BYTES_PER_SAMPLE[args.workload]
            if world == 1 and not args.no_cpu_baseline:  # the CPU leg runs on rank 0 at N = 1 only
                cpu, bytes_per_sample, _, (cpu_frame, cpu_spp) = cpu_baseline(scene, res, args.cpu_seconds)
                out["cpu_baseline"] = cpu
                out["parity"] = device_parity(scene, local_rank, cpu_frame, cpu_spp, args.scheduler)
            if world == 1 and not args.no_stats:
                stats = path_statistics(scene, local_rank, spp, args.scheduler)
                out["rays_per_s"] = value * 1e6 * stats["rays_per_sample"]
                out["mean_path_length"] = stats["mean_path_length"]
                out["path_statistics"] = stats
            if main_multi:
                out["multi_gpu"] = main_multi
                if main_multi.get("reduced_film_equals_1gpu_render") is False:
                    out["value"] = None  # a frame that is not the frame has no throughput
                    out["error"] = "the film the collective delivered differs from a 1-GPU render of the same tiles"
            # one launch renders this rank's shard: samples_per_step / world samples
            launch_samples = samples_per_step / world
            import torch
            simds = torch.cuda.get_device_properties(local_rank).multi_processor_count * 4
            headline_key = args.workload if args.sampler == "Independent" else f"{args.workload}:{args.sampler}"
            # every configuration whose kernels the counter passes see: the headline at the timed spp, the others at PMC_SPP
            pmc_configs = [{"key": headline_key, "workload": args.workload, "spp": spp, "sampler": args.sampler, "variant": variant, "scheduler": args.scheduler}]
            extra, extra_state = [], {}
            if world == 1 and not args.no_extra and args.workload == "c2" and args.spp is None and args.sampler == "Independent":
                # Every other BASELINE configuration AT ITS STATED SIZE on this GPU (round 4; round 3 timed them at reduced spp), each with
                # its own oracle leg on a bounded sample and the parity of the device against it: C1 = configs[0] whole (+ the reference's
                # own code beside it), C3 at 4096 spp, C4 = the 8-GPU configuration's 1-GPU number at 1024 spp, C5 at 2048 of its 65 536 spp
                # (>= 1024; the Independent sampler's throughput is spp-invariant beyond the item drain), and C2
                # once more with the sampler class + filter the reference's converter writes for the README scenes
                # (tools/tungsten2luisa.py:373-412 there: a low-discrepancy sampler, Gaussian r = 1 -- the stand-in's filter already).
                # (workload, timed steps, spp (None = the configuration's), seconds of oracle, whole configuration on the CPU, sampler)
                for w, steps, spp_o, cpu_s, whole, sampler in (("c1", 3, None, 30.0, True, "Independent"), ("c3", 1, None, 8.0, False, "Independent"),
                                                               ("c4", 1, None, 10.0, False, "Independent"), ("c5", 1, BENCH_SPP_CAP["c5"], 8.0, False, "Independent"),
                                                               ("c2", 1, None, 8.0, False, "PaddedSobol")):  # (round 5: the low-discrepancy line has its own oracle leg and parity)
                    v, ms, kms, var, sc, r, sp, d = run_workload(w, args, rank, world, local_rank, tmp, steps, 1, spp_o, sampler, warmup_spp=None if w in ("c1", "c5") else 8)  # (C5: wavefront mode sizes its queues by the frame's spp -- a warm-up on fewer samples leaves a 76-89 GB allocation inside the timed step)
                    e = {"workload": d, "sampler": sampler, "spp_timed": sp, "spp_config": WORKLOADS[w][2], "value": v, "unit": "Msamples/s", "steps": steps, "ms_per_step": ms,
                         "kernel_ms": kms, "kernel": kernel_name(var), "algorithmic_bytes_per_sample": ALGORITHMIC_BYTES_PER_SAMPLE[w]}
                    if cpu_s > 0.0 and not args.no_cpu_baseline:
                        e["cpu_baseline"], e["algorithmic_bytes_per_sample"], _, (cpu_frame, cpu_spp) = cpu_baseline(sc, r, cpu_s, full_spp=sp if whole else None)
                        e["parity"] = device_parity(sc, local_rank, cpu_frame, cpu_spp)
                    else:
                        e["parity"] = None
                    if whole and not args.no_cpu_baseline:  # BASELINE configs[0]: Cornell Box 512x512, 64 spp, depth 8 on the CPU -- the whole configuration --
                        # and the REFERENCE'S OWN code beside it (oracle/_ref, a bounded sample of the same frame); absent where oracle/_ref was not built
                        ref = reference_leg("c1", r, sp, sampler, 10.0)
                        if ref is not None:
                            e["cpu_reference"] = ref
                    key = w if sampler == "Independent" else f"{w}:{sampler}"
                    pmc_spp = min(sp, PMC_SPP.get(key, sp))
                    if not args.no_stats:
                        extra_state[key] = path_statistics(sc, local_rank, pmc_spp)
                    pmc_configs.append({"key": key, "workload": w, "spp": pmc_spp, "sampler": sampler, "variant": var, "scheduler": "auto"})
                    e["_key"] = key
                    e["_variant"] = var
                    e["_samples"] = r[0] * r[1] * sp
                    extra.append(e)
            # the reference's own code beside the HEADLINE number too (round 6): MegaPath::Li of oracle/_ref on the bench's own C2 description
            # (InlineMesh form: the same instances, transforms and materials), one process per host thread, a bounded time
            if world == 1 and not args.no_cpu_baseline and args.workload == "c2":
                ref = reference_leg("c2", res, spp, args.sampler, 12.0)
                if ref is not None:
                    out["cpu_reference"] = ref
            pmc_all = {}
            if world == 1 and not args.no_pmc:
                pmc_all = live_pmc(pmc_configs)

            def pmc_of(key, timed_spp):
                p = pmc_all.get(key)
                if p is None or "hbm_bytes_per_sample" not in p:
                    return None, None
                src = (f"live: rocprofv3 --pmc passes of this run ({', '.join(n for n, _ in PMC_PASSES)}), the same frame at {p['spp']} spp"
                       + ("" if p["spp"] == timed_spp else f" (per-sample counters x the timed launch's samples; the frame is timed at {timed_spp} spp)"))
                return p, src

            pmc, pmc_source = pmc_of(headline_key, spp)
            if pmc is None:
                prof = os.path.join(ROOT, "profiles", f"pmc_{args.workload}.json")
                if os.path.exists(prof):
                    try:
                        p = json.load(open(prof))
                        if p.get("source_hash") == source_hash() and p.get("hbm_bytes_per_sample"):  # a profile of another build is not evidence
                            p.setdefault("valu_wave_instr_per_sample", p.get("valu_wave_instructions_per_sample"))
                            pmc, pmc_source = p, f"profiles/pmc_{args.workload}.json (same source hash)"
                    except Exception:
                        pass
            # (a scheduler override reaches the counting and parity renders as well: ADVICE r05)
            out["roofline"] = roofline_block(args.workload, variant, mean_kernel_ms, launch_samples, bytes_per_sample, pmc, pmc_source, out.get("path_statistics"), simds)
            for e in extra:
                key, var, n = e.pop("_key"), e.pop("_variant"), e.pop("_samples")
                p, src = pmc_of(key, e["spp_timed"])
                e["roofline"] = roofline_block(key.split(":")[0], var, e["kernel_ms"], n, e["algorithmic_bytes_per_sample"], p, src, extra_state.get(key), simds)
            if extra:
                out["extra_configs"] = extra
            if multi_extra:
                out["extra_configs"] = out.get("extra_configs", []) + multi_extra
            out["source_hash"] = source_hash()  # the kernel + BVH-builder sources this line was measured on (profiles/*.json carry the same)
            print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
