#!/usr/bin/env python3
"""Per-loop instruction census of a disassembled megakernel variant, priced with the on-box issue-cost table
(profiles/archive/r03_valu_peak.json, tools/valu_peak2.hip): for the traversal loop (the innermost loop that holds the four
global_load_lds), its node step and its leaf step, and for the whole kernel: instructions per class and the VALU issue cycles
one wave-level pass costs a SIMD.

    tools/kernel_resources.sh 0 && /opt/rocm/lib/llvm/bin/llvm-objdump -d /tmp/kres/v0.co > /tmp/kres/v0.s
    python tools/isa_census.py /tmp/kres/v0.s [profiles/archive/r03_valu_peak.json]

Cost classes (cycles per wave64 instruction per SIMD with >= 2 waves resident, measured):
  full   v_fma/mul/add/sub/fmac_f32, v_and/or/xor_b32, v_add/sub_u32, v_mov_b32                                ~2.4
  half   every other VALU op: min/max/med3, cvt, cmp, cndmask (e64), shifts, bfe/bfi/perm, lshl_add, mad, DPP, packed ~4.2
  trans  v_rcp/rsq/sqrt/exp/log/sin/cos                                                                          ~8.2
  cndmask_e32 right behind the v_cmp that wrote vcc: the pair costs 6.1, i.e. ~2.0 for the select
"""
import json
import re
import sys

FULL = re.compile(r"v_(fma_f32|mul_f32|add_f32|sub_f32|subrev_f32|fmac_f32|mac_f32|and_b32|or_b32|xor_b32|add_u32|sub_u32|subrev_u32|mov_b32|add_co_u32|addc_co_u32|not_b32)(_e32|_e64)?$")
TRANS = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32")


def classify(op):
    if not op.startswith("v_"):
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "scratch_", "buffer_", "flat_")):
            return "vmem"
        if op.startswith("s_waitcnt"):
            return "waitcnt"
        if op.startswith("s_nop"):
            return "nop"
        if op.startswith(("s_cbranch", "s_branch")):
            return "branch"
        return "salu"
    if "_dpp" in op or "_sdwa" in op:
        return "half"
    if op.startswith("v_cndmask_b32_e32"):
        return "cndmask_e32"
    if TRANS.match(op):
        return "trans"
    if FULL.match(op):
        return "full"
    return "half"


def main():
    path = sys.argv[1]
    cost = {"full": 2.4, "half": 4.2, "trans": 8.2, "cndmask_e32": 2.0}
    if len(sys.argv) > 2:
        t = {r["op"]: r["w4"] for r in json.load(open(sys.argv[2]))["results"]}
        cost["full"] = (t["v_fma_f32"] + t["v_mul_f32"] + t["v_add_f32"] + t["v_and_b32"] + t["v_add_u32"] + t["v_mov_b32"]) / 6
        cost["half"] = (t["v_max_f32"] + t["v_cvt_f32_ubyte0"] + t["v_cmp_lt_f32 vcc"] + t["v_cndmask_b32 sgpr mask (e64)"] + t["v_lshl_add_u32"] + t["v_min_u32"]) / 6
        cost["trans"] = t["v_rcp_f32"]
        cost["cndmask_e32"] = 2 * t["v_cmp + v_cndmask pair"] - t["v_cmp_lt_f32 vcc"]
    lines = open(path).read().split("\n")
    ops = []
    for l in lines:
        m = re.match(r"\s+(\S+)\s.*//\s+([0-9A-F]{12}):", l)
        ops.append((m.group(1), int(m.group(2), 16)) if m else None)
    base = None
    for l in lines:
        m = re.match(r"([0-9a-f]{16}) <(.*)>:", l)
        if m and any(k in m.group(2) for k in ("megapath_kernel", "megapool_kernel", "megavpt_kernel", "heavy_kernel")):
            base = int(m.group(1), 16)
    addr = {o[1]: i for i, o in enumerate(ops) if o}

    def census(lo, hi, title):
        c = {}
        for o in ops[lo:hi + 1]:
            if o:
                k = classify(o[0])
                c[k] = c.get(k, 0) + 1
        valu = sum(c.get(k, 0) for k in ("full", "half", "trans", "cndmask_e32"))
        cycles = sum(c.get(k, 0) * cost[k] for k in ("full", "half", "trans", "cndmask_e32"))
        detail = {}
        for o in ops[lo:hi + 1]:
            if o and o[0].startswith(("v_", "ds_", "global_", "scratch_")):
                name = re.sub(r"_e(32|64)$", "", o[0])
                detail[name] = detail.get(name, 0) + 1
        top = sorted(detail.items(), key=lambda kv: -kv[1])[:14]
        print(f"{title}: lines {lo}-{hi}, {hi - lo + 1} instructions, VALU {valu} = {cycles:.0f} issue cycles  " +
              "  ".join(f"{k} {c.get(k, 0)}" for k in ("full", "half", "cndmask_e32", "trans", "lds", "vmem", "salu", "branch", "waitcnt", "nop")))
        print("    " + ", ".join(f"{k} x{v}" for k, v in top))
        return cycles

    # loops = backward branches
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"(s_cbranch\w+|s_branch)\s+\S+\s+//\s+([0-9A-F]{12}):\s+\S+\s+<[^+]+\+0x([0-9A-Fa-f]+)>", l)
        if m and base is not None:
            cur, tgt = int(m.group(2), 16), base + int(m.group(3), 16)
            if tgt < cur and tgt in addr:
                loops.append((addr[tgt], i))
    fetch = [i for i, o in enumerate(ops) if o and o[0].startswith("global_load_lds")]
    print(f"cost table: " + ", ".join(f"{k} {v:.2f}" for k, v in cost.items()))
    whole = [i for i, o in enumerate(ops) if o]
    census(whole[0], whole[-1], "whole object")
    if fetch:
        enclosing = sorted([(hi - lo, lo, hi) for lo, hi in loops if lo <= fetch[0] and hi >= fetch[-1]])
        if enclosing:
            _, lo, hi = enclosing[0]
            census(lo, hi, "traversal loop (one wave-level iteration: node step + leaf step + bookkeeping)")
            # node step: from the loop head to the first triangle fetch (three global_load_dwordx4 in a row) after the packet fetch
            tri = [i for i in range(fetch[-1], hi) if ops[i] and ops[i][0].startswith("global_load_dwordx4")]
            if tri:
                census(lo, tri[0] - 1, "  node step (fetch + slab test + sort + pushes; up to the triangle fetch)")
                census(tri[0], hi, "  leaf step + ray switch + loop control")


if __name__ == "__main__":
    main()
