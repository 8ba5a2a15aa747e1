#!/usr/bin/env python3
"""Static per-region instruction counts of the production kernels, taken from the gfx950 assembly of the code object.

build() runs this after compiling libraytrace_hip.so: it asks hipcc for the device assembly of the same translation unit with the
same flags, cuts the production instantiation of wf_travq at the labels the kernel plants (WQ_MARK: labels, not instructions) and
writes raytracinggpu_amd/static_counts.json.  bench.py multiplies these counts with the step counters of the counting
instantiation (rt_count_work: iterations, refill passes, rounds, fetches, TRI steps, BOX steps) to get the vector wave-instructions
of a launch IN THE RUN, and prices them against the vector-issue peak (roofline.bound = "valu_issue").

Every instruction is also weighted by its measured issue cost on gfx950 (tools/ubench/issue_table.hip, profiles/round3/
issue_table.jsonl): fma / mul / add / sub / mov / and / or / add_u32 / lshrrev issue at the full rate (weight 1), min / max /
compares / cndmask / mbcnt / every three-operand integer form / lshlrev / cvt / binary64 arithmetic at half of it (weight 2),
rcp / sqrt / rsq at a quarter (weight 4); one weight unit = 2 cycles of one SIMD.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "raytracinggpu_amd", "csrc", "rt_capi.hip")
OUT = os.path.join(ROOT, "raytracinggpu_amd", "static_counts.json")

FULL_RATE = {"v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32",
             "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32"}
QUARTER_RATE = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
EIGHTH_RATE = {"v_rcp_f64", "v_sqrt_f64", "v_rsq_f64"}
NOT_SALU = ("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_load", "s_buffer_load", "s_sleep", "s_setprio", "s_code_end", "s_getpc", "s_setpc", "s_swappc", "s_memtime", "s_memrealtime", "s_sendmsg", "s_trap")


def classify(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_"):
        w = 1 if base in FULL_RATE else 4 if base in QUARTER_RATE else 8 if base in EIGHTH_RATE else 2
        return "valu", w
    if base.startswith("ds_"):
        return "lds", 0
    if base.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", 0
    if base.startswith(("s_load", "s_buffer_load")):
        return "smem", 0
    if base.startswith(("s_cbranch", "s_branch")):
        return "branch", 0
    if base.startswith("s_") and not base.startswith(NOT_SALU):
        return "salu", 0
    return "other", 0


def kernel_text(asm, mangled):
    out, on = [], False
    for line in asm.splitlines():
        if line.startswith(mangled + ":"):
            on = True
        if on:
            out.append(line)
            if line.startswith(".Lfunc_end"):
                break
    return out


def count(lines):
    c = {"valu": 0, "valu_weight": 0, "salu": 0, "branch": 0, "lds": 0, "vmem": 0, "smem": 0, "other": 0}
    for line in lines:
        t = line.split(";")[0].strip()
        if not t or t.endswith(":") or t.startswith("."):
            continue
        kind, w = classify(t.split()[0])
        c[kind] += 1
        c["valu_weight"] += w
    return c


def sub(a, b):
    return {k: a[k] - b[k] for k in a}


def regions(lines):
    """{name: [(first, last)]} line ranges between rt_mark_<name>_begin_N / rt_mark_<name>_end_N style labels (in layout order)."""
    marks = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"\s*rt_mark_(\w+?)_(\d+):", l)] if m]
    return marks


def travq_counts(lines):
    marks = regions(lines)
    pos = {}
    for i, name in marks:
        pos.setdefault(name, []).append(i)
    need = ("head", "refill_begin", "round_begin", "fetch_begin", "fetch_end", "round_end", "refill_end", "tri_begin", "tri_end", "box_begin", "box_end")
    missing = [n for n in need if n not in pos]
    if missing:
        raise SystemExit(f"static_counts: markers missing from the code object: {missing}")
    one = {n: pos[n][0] for n in need}
    reg = lambda a, b: count(lines[one[a]:one[b]])
    inside = lambda name, a, b_: [(x, y) for x, y in zip(pos.get(name + "_begin", []), pos.get(name + "_end", [])) if one[a] < x < one[b_]]
    span = lambda pairs: count([l for x, y in pairs for l in lines[x:y]])
    trilit = span(inside("trilit", "tri_begin", "tri_end"))
    tdivs = inside("tdiv", "tri_begin", "tri_end")                   # the two inlined triangle tests of a TRI step: t = dot(AO, N) / det, entered when some lane accepted
    tdiv = span(tdivs)
    n_tdiv = max(len(tdivs), 1)
    lflag = span(inside("lflag", "tri_begin", "tri_end"))            # fixed-point instantiations: the real-box check of accepted triangles in flagged leaves (behind a vote)
    lp = [x for x in pos.get("lpush_begin", []) if one["box_begin"] < x < one["box_end"]]
    lp2 = [x for x in pos.get("lpush2_begin", []) if one["box_begin"] < x < one["box_end"]]
    lpe = [x for x in pos.get("lpush_end", []) if one["box_begin"] < x < one["box_end"]]
    if not (lp and lp2 and lpe):
        raise SystemExit("static_counts: leaf-push markers missing")
    lpush1, lpush2 = count(lines[lp[0]:lp2[0]]), count(lines[lp2[0]:lpe[0]])
    loop_first = one["head"]
    loop_last = max(one["box_end"], one["tri_end"], one["refill_end"])
    out = {
        "loop_head": reg("head", "refill_begin"),                      # per loop iteration: the step dispatch up to the refill test
        "retire": sub(reg("refill_begin", "round_begin"), reg("fetch_begin", "fetch_end")),   # per refill pass: retire finished rays, look for free slots
        "round": reg("round_begin", "round_end"),                       # per hand-off round (staged rays -> free slots)
        "fetch": reg("fetch_begin", "fetch_end"),                       # per queue fetch (64 slots)
        "dispatch": reg("refill_end", "tri_begin"),                     # which step runs next: scalar, per loop iteration; its few vector instructions are the TRI
                                                                        # step's first ones, placed in front of the label (priced per TRI step)
        "tri": sub(sub(sub(reg("tri_begin", "tri_end"), trilit), tdiv), lflag),     # per TRI step (128 triangle tests) without the blocks below
        "tri_literal_blocks": trilit,
        "lflag": lflag,                                   # literal beta / gamma divisions: rare, not priced
        "tdiv": {k: v / n_tdiv for k, v in tdiv.items()},               # per t-division block entered (a TRI step has two)
        "box": sub(sub(reg("box_begin", "box_end"), lpush1), lpush2),   # per BOX step (64 sibling pairs) without the leaf-queue pushes
        "lpush": lpush1, "lpush2": lpush2,                              # per first / second leaf-queue push entered
        "whole_kernel": count(lines),
        "prologue_epilogue": sub(count(lines), count(lines[loop_first:loop_last])),
    }
    return out


def travq_qw_counts(lines):
    """The 4-wide instantiation (wf_travq<.., QN, QW>): the same regions, its BOX step = the core (pop, four loads, unpack, four box tests, masks, the per-slot counter
    update) + one block per child that some lane pushes as an internal node / as a leaf (counted per block entered: the step counters' leaf_push2 / leaf_push)."""
    marks = regions(lines)
    pos = {}
    for i, name in marks:
        pos.setdefault(name, []).append(i)
    need = ("head", "refill_begin", "round_begin", "fetch_begin", "fetch_end", "round_end", "refill_end", "tri_begin", "tri_end", "boxw_begin", "ipushw_begin", "lpushw_begin", "lpushw_end", "boxw_end")
    missing = [n for n in need if n not in pos]
    if missing:
        raise SystemExit(f"static_counts: markers missing from the 4-wide code object: {missing}")
    one = {n: pos[n][0] for n in need}
    reg = lambda a, b: count(lines[one[a]:one[b]])
    inside = lambda name, a, b_: [(x, y) for x, y in zip(pos.get(name + "_begin", []), pos.get(name + "_end", [])) if one[a] < x < one[b_]]
    span = lambda pairs: count([l for x, y in pairs for l in lines[x:y]])
    trilit = span(inside("trilit", "tri_begin", "tri_end"))
    tdivs = inside("tdiv", "tri_begin", "tri_end")
    tdiv = span(tdivs)
    n_tdiv = max(len(tdivs), 1)
    lflag = span(inside("lflag", "tri_begin", "tri_end"))            # the real-box check of accepted triangles in flagged leaves: behind a vote, counted per block entered (literal_box_fallbacks)
    quarter = lambda c: {k: v / 4.0 for k, v in c.items()}
    def add(a, b):
        return {k: a[k] + b[k] for k in a}
    return {
        "loop_head": reg("head", "refill_begin"),
        "retire": sub(reg("refill_begin", "round_begin"), reg("fetch_begin", "fetch_end")),
        "round": reg("round_begin", "round_end"),
        "fetch": reg("fetch_begin", "fetch_end"),
        "dispatch": reg("refill_end", "tri_begin"),
        "tri": sub(sub(sub(reg("tri_begin", "tri_end"), trilit), tdiv), lflag),
        "tri_literal_blocks": trilit,
        "lflag": lflag,
        "tdiv": {k: v / n_tdiv for k, v in tdiv.items()},
        "box": add(reg("boxw_begin", "ipushw_begin"), reg("lpushw_end", "boxw_end")),    # per BOX step (64 quads = 256 boxes) without the push blocks
        "lpush2": quarter(reg("ipushw_begin", "lpushw_begin")),                            # per internal-child push block entered
        "lpush": quarter(reg("lpushw_begin", "lpushw_end")),                               # per leaf-child push block entered
        "whole_kernel": count(lines),
    }


def advance_counts(lines):
    """wf_advance<false, false>: what a path pays for per launch, region by region (ADV_MARK labels of rt_wavefront.hip.h).  A region's instructions execute for the lanes
    (paths) that take its branch; `rest` = decode, record reads and writes, the emission's root-box tests and flag words."""
    marks = regions(lines)
    pos = {}
    for i, name in marks:
        pos.setdefault(name, []).append(i)
    out, covered = {}, count([])
    for name in ("closex", "closey", "diffuse", "bounce", "fold", "spheres"):
        b, e = pos.get("adv_" + name + "_begin"), pos.get("adv_" + name + "_end")
        if not b or not e or e[0] < b[0]:
            out[name] = None
            continue
        out[name] = count(lines[b[0]:e[0]])
        covered = {k: covered[k] + out[name][k] for k in covered}
    whole = count(lines)
    out["rest"] = sub(whole, covered)
    out["whole_kernel"] = whole
    return out


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    flags = [f for f in ge.HIP_FLAGS if f not in ("-fPIC", "-shared")]
    asm = subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", "-o", "-", SRC], check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    res = {"source": "hipcc -S --cuda-device-only of raytracinggpu_amd/csrc/rt_capi.hip with the flags of __graft_entry__.HIP_FLAGS",
           "weights": "valu_weight: 1 = full-rate instruction (2 SIMD cycles per wave64), 2 = half rate, 4 = quarter, 8 = binary64 transcendental"}
    tq = kernel_text(asm, "_ZN3rtk8wf_travqILb0ELi64ELb0ELb0ELb0ELb0EEEvNS_5SceneENS_5FrameENS_7WfStateEiiii")
    if not tq:
        raise SystemExit("static_counts: wf_travq<false, 64, false, false> not found in the assembly")
    res["wf_travq"] = travq_counts(tq)
    tw = kernel_text(asm, "_ZN3rtk8wf_travqILb0ELi64ELb0ELb0ELb1ELb1EEEvNS_5SceneENS_5FrameENS_7WfStateEiiii")
    if not tw:
        raise SystemExit("static_counts: wf_travq<false, 64, false, false, true, true> (the 4-wide BOX step) not found in the assembly")
    res["wf_travq_qw"] = travq_qw_counts(tw)
    for name, mangled in (("wf_advance", "_ZN3rtk10wf_advanceILb0ELb0EEEvNS_5SceneENS_5FrameENS_7WfStateE"),
                          ("wf_advance_first", "_ZN3rtk10wf_advanceILb0ELb1EEEvNS_5SceneENS_5FrameENS_7WfStateE")):
        t = kernel_text(asm, mangled)
        if t:
            res[name] = advance_counts(t) if name == "wf_advance" else {"whole_kernel": count(t)}
    m = re.search(r"\.name:\s+_ZN3rtk8wf_travqILb0ELi64ELb0ELb0EEE.*?\n(.*?)\.wavefront_size", asm, re.S)
    with open(OUT, "w") as f:
        json.dump(res, f, indent=1)
    w = res["wf_travq_qw"]
    print("static_counts: wf_travq 4-wide per step: BOX %d valu (weight %d) %d salu + %.1f per internal push block + %.1f per leaf push block | TRI %d (%d) %d | round %d | fetch %d" % (
        w["box"]["valu"], w["box"]["valu_weight"], w["box"]["salu"], w["lpush2"]["valu"], w["lpush"]["valu"], w["tri"]["valu"], w["tri"]["valu_weight"], w["tri"]["salu"], w["round"]["valu"], w["fetch"]["valu"]))
    a = res.get("wf_advance", {})
    if a:
        print("static_counts: wf_advance regions (valu / weight): " + ", ".join(f"{k} {v['valu']}/{v['valu_weight']}" for k, v in a.items() if v))
    t = res["wf_travq"]
    print("static_counts: wf_travq per step: BOX %d valu (weight %d) %d salu + leaf pushes %d / %d | TRI %d (%d) %d + %.0f per t-division block | round %d (%d) %d | fetch %d | retire %d | head+dispatch %d" % (
        t["box"]["valu"], t["box"]["valu_weight"], t["box"]["salu"], t["lpush"]["valu"], t["lpush2"]["valu"], t["tri"]["valu"], t["tri"]["valu_weight"], t["tri"]["salu"], t["tdiv"]["valu"],
        t["round"]["valu"], t["round"]["valu_weight"], t["round"]["salu"], t["fetch"]["valu"], t["retire"]["valu"], t["loop_head"]["valu"] + t["dispatch"]["valu"]))


if __name__ == "__main__":
    main()
