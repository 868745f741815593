#!/usr/bin/env python3
"""Summarise a tools/profile_c2.sh output directory (rocprofv3 rocpd sqlite files) into a small JSON
that is committed under profiles/.

    python tools/summarize_prof.py gpurun_out/prof_r01c profiles/archive/r01_c2_1024spp.json [samples_per_launch]
"""
import json
import os
import sqlite3
import sys

KERNEL = "%megap%_kernel%"


def main():
    src, dst = sys.argv[1], sys.argv[2]
    samples = float(sys.argv[3]) if len(sys.argv) > 3 else None
    out = {"source": src, "tool": "rocprofv3 --kernel-trace --stats / --pmc (separate passes)", "kernel": None}
    try:  # the state of the kernel sources this profile belongs to (the same hash bench.py puts into its line)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        out["source_hash"] = bench.source_hash()
    except Exception as e:  # noqa: BLE001
        out["source_hash"] = f"unavailable: {e}"
    db = sqlite3.connect(os.path.join(src, "trace", "trace_results.db"))
    rows = list(db.execute("select name, (end-start)/1e6, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
                           "from kernels where name like ? order by start", (KERNEL,)))
    if rows:
        out["kernel"] = rows[0][0]
        out["launches"] = len(rows)
        out["kernel_ms_all"] = [r[1] for r in rows]
        out["kernel_ms_mean"] = sum(r[1] for r in rows) / len(rows)
        out["grid_threads"], out["workgroup"], out["lds_bytes"], out["scratch_bytes_per_lane"] = rows[0][2:6]
        out["arch_vgpr"], out["accum_vgpr"], out["sgpr"] = rows[0][6:9]
    out["top_kernels"] = [dict(zip(("name", "calls", "total_us", "avg_us", "percent"), r)) for r in db.execute("select * from top_kernels limit 6")]
    counters = {}
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
        path = os.path.join(src, sub, "pmc_results.db")
        if not os.path.exists(path):
            continue
        d = sqlite3.connect(path)
        for name, value, n in d.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? group by counter_name", (KERNEL,)):
            counters[name] = value
    out["counters_mean_per_launch"] = counters
    c = counters
    if "FETCH_SIZE" in c:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reads exactly
        # half of a wide coalesced stream (128-B requests tallied as 64 B) -> doubled before comparing with byte counts;
        # "other access widths and WRITE_SIZE are uncalibrated", so the raw figure is kept next to it.
        out["hbm_read_bytes_per_launch_raw"] = c["FETCH_SIZE"] * 1024
        out["hbm_read_bytes_per_launch_corrected_x2"] = c["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in c:
        out["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes_per_launch"] = c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024
        if rows:
            out["hbm_gbps"] = out["hbm_bytes_per_launch"] / (out["kernel_ms_mean"] * 1e-3) / 1e9
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        out["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
    if "SQ_WAVE_CYCLES" in c:
        out["wave_cycle_breakdown"] = {k: c[k] / c["SQ_WAVE_CYCLES"] for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY") if k in c}
    if samples:
        out["samples_per_launch"] = samples
        if "hbm_bytes_per_launch" in out:
            out["hbm_bytes_per_sample"] = out["hbm_bytes_per_launch"] / samples
        if "SQ_INSTS_VALU" in c:
            out["valu_wave_instructions_per_sample"] = c["SQ_INSTS_VALU"] / samples
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
