#!/usr/bin/env python
"""Instruction-class histogram of the gfx950 kernels, and the mix-weighted cost of a vector instruction.

    python scripts/isa_hist.py [--costs profiles/r03_valu_ceiling.csv] [--waves 5] [kernel-name-substring ...]

Compiles every device source with the build's own flags to assembly (`hipcc -S --cuda-device-only`,
no GPU needed), splits it per kernel, and counts instructions by class -- over the whole kernel
(static) and over its innermost loops only (a label followed by a backward branch to it: the walk,
the sweeps), which is where the time goes.  With --costs (the CSV scripts/valu_ceiling.hip prints)
every class is priced at the measured cycles a SIMD spends per wave64 instruction at --waves
waves per SIMD, and the mix-weighted mean is printed: the figure bench.py multiplies
SQ_INSTS_VALU with instead of a fixed 4.
"""
import argparse
import collections
import csv
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# mnemonic (regex, first match wins) -> class measured by valu_ceiling.hip.  Measured on MI355X
# (profiles/r03_valu_ceiling.csv): plain VOP2 add / sub / and / or / xor / mov / mul_f32 / fmac_f32 /
# lshrrev issue in ~2.3 cycles per wave64 instruction at saturation, everything else (compares, selects,
# min / max, fma, conversions, every VOP3-only op, f64 arithmetic, 64-bit shifts, readlane, DPP) in ~4.2.
VALU_CLASSES = [
    (r"v_(add|sub|subrev)_(u32|i32)(_e32|_e64)?$|v_(add|sub|subrev)_nc_u32", "add_u32"),
    (r"v_(and|or|xor|not)_b32", "o_and_b32"),
    (r"v_mov_b32(_e32|_e64)?$", "o_mov_b32"),
    (r"v_(add|sub|subrev)_f32", "add_f32"),
    (r"v_mul_f32|v_mul_legacy_f32", "o_mul_f32"),
    (r"v_(fmac|mac)_f32", "o_fmac_f32"),
    (r"v_lshrrev_b32|v_ashrrev_i32", "o_lshrrev_b32"),
    (r"v_lshlrev_b32", "lshl_b32"),
    (r"v_(add|mul|fma|fmac)_f64|v_(ldexp|fract|div_\w+|rcp|rsq|sqrt|frexp\w*)_f64", "add_f64"),
    (r"v_(min|max)_f64", "o_max_f64"),
    (r"v_(floor|ceil|trunc|rndne)_f64", "floor_f64"),
    (r"v_cmpx?_\w+_f64", "cmp_f64"),
    (r"v_cvt_f64_|v_cvt_\w+_f64", "cvt_f64_f32+cvt_f32_f64"),
    (r"v_(lshlrev|lshrrev|ashrrev)_b64|v_lshl_add_u64", "lshl_b64"),
    (r"v_mov_b64", "o_mov_b64"),
    (r"v_mad_[ui]64_[ui]32", "mad_u64_u32"),
    (r"v_mul_(lo|hi)_[ui]32", "mul_lo_u32"),
    (r"v_pk_", "pk_add_f32"),
    (r"v_readlane|v_readfirstlane|v_writelane", "readlane"),
    (r"v_\w+_dpp|v_permlane", "dpp_mov"),
    (r"v_(add|sub|subrev)_co_|v_(addc|subb|subbrev)_co_", "o_add_co_u32"),
    (r"v_cmpx?_\w+_(u32|i32|u16|i16|u64|i64)", "cmp_u32"),
    (r"v_cmpx?_", "cmp_f32"),
    (r"v_cndmask", "cnd_sgpr"),
    (r"v_(min|max)3?_[iu]32", "min_u32"),
    (r"v_(min|max)3?_f32", "o_max_f32"),
    (r"v_(fma|mad)_f32", "fma_f32"),
    (r"v_cvt_f32_(u32|i32)", "o_cvt_f32_u32"),
    (r"v_cvt_", "cvt_u32_f32"),
    (r"v_bfe_|v_bfi_|v_alignb", "bfe_u32"),
    (r"v_(lshl_add|add_lshl|lshl_or|and_or|or3|xad)_", "o_lshl_add_u32"),
    (r"v_add3_", "o_add3_u32"),
    (r"v_mbcnt", "mbcnt"),
    (r"v_bcnt", "bcnt"),
    (r"v_ff[bh]", "ffbl"),
    (r"v_mul_[ui]32_[ui]24|v_mad_[ui]32_[ui]24", "mad_u32_u24"),
    (r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", "trans_f32"),
    (r"v_", "and_or_b32"),          # anything else: priced as a VOP3 integer op
]
OTHER = [
    (r"ds_", "lds"),
    (r"(global|buffer|flat|scratch)_(load|store|atomic)", "vmem"),
    (r"s_(load|buffer_load|store|memtime|memrealtime|dcache)", "smem"),
    (r"s_waitcnt|s_nop|s_barrier|s_sleep|s_setprio", "wait"),
    (r"s_cbranch|s_branch|s_endpgm|s_setpc|s_getpc|s_swappc", "branch"),
    (r"s_", "salu"),
]


def classify(mn):
    if mn.startswith("v_"):
        for pat, cls in VALU_CLASSES:
            if re.match(pat, mn):
                return "valu:" + cls
    for pat, cls in OTHER:
        if re.match(pat, mn):
            return cls
    return "other"


def kernels_of(asm_text):
    """name -> list of (label or None, mnemonic) in program order"""
    out, cur = {}, None
    for line in asm_text.splitlines():
        m = re.match(r"^(_Z\w+|pya_\w+):\s", line + " ")
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        ml = re.match(r"^(\.LBB\w+):", line)
        if ml:
            out[cur].append((ml.group(1), None, None))
            continue
        mi = re.match(r"^\s+([a-z]\w+)\s*(.*?)(;.*)?$", line)
        if mi and not mi.group(1).startswith("."):
            out[cur].append((None, mi.group(1), mi.group(2)))
    return out


def inner_loops(stream):
    """indices of instructions inside innermost loops (label ... backward branch to that label,
    with no other backward-branch loop nested inside)"""
    pos = {}
    for i, (lab, mn, ops) in enumerate(stream):
        if lab:
            pos[lab] = i
    loops = []
    for i, (lab, mn, ops) in enumerate(stream):
        if mn and mn.startswith("s_cbranch") or mn == "s_branch":
            tgt = (ops or "").strip().split()[-1] if ops else ""
            if tgt in pos and pos[tgt] < i:
                loops.append((pos[tgt], i))
    inner = [l for l in loops if not any(o != l and l[0] <= o[0] and o[1] <= l[1] for o in loops)]
    idx = set()
    for a, b in inner:
        idx.update(range(a, b + 1))
    return idx


def demangle(names):
    try:
        p = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True)
        return dict(zip(names, p.stdout.splitlines()))
    except Exception:
        return {n: n for n in names}


def load_costs(path, waves):
    """class -> cycles a SIMD spends per wave64 instruction (rows without scalar filler, at `waves`
    waves per SIMD) from the CSV scripts/valu_ceiling.hip prints"""
    cost = {}
    with open(path, newline="") as f:
        rows = [r for r in csv.DictReader(l for l in f if not l.startswith("#"))]
    for r in rows:
        if r["salu_per_valu"] not in ("0", "0.19") or int(r["waves_per_simd"]) != waves:
            continue
        cost[r["class"]] = float(r["cycles_per_inst_simd"])
    cost.setdefault("trans_f32", 2.0 * cost.get("and_or_b32", 4.25))
    return cost


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--costs")
    ap.add_argument("--waves", type=int, default=6)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("filters", nargs="*")
    args = ap.parse_args()
    from pyascore_amd import build as b
    result = {}
    cost = load_costs(args.costs, args.waves) if args.costs else None
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(f for f in os.listdir(b.CSRC) if f.endswith(".hip")):
            s = os.path.join(tmp, src + ".s")
            subprocess.check_call([b.HIPCC] + b.DEVICE_FLAGS + ["-S", "--cuda-device-only", "-o", s, os.path.join(b.CSRC, src)],
                                  stderr=subprocess.DEVNULL)
            ks = kernels_of(open(s).read())
            names = demangle(list(ks))
            for raw, stream in ks.items():
                nice = re.sub(r"^void ", "", names[raw]).split("(")[0]
                if not nice.startswith("pya_"):
                    continue
                if args.filters and not any(f in nice for f in args.filters):
                    continue
                loops = inner_loops(stream)
                h_all, h_loop = collections.Counter(), collections.Counter()
                for i, (lab, mn, ops) in enumerate(stream):
                    if not mn:
                        continue
                    c = classify(mn)
                    h_all[c] += 1
                    if i in loops:
                        h_loop[c] += 1
                result[nice] = {"all": dict(h_all), "inner_loops": dict(h_loop)}
    for name, hh in sorted(result.items()):
        for scope in ("all", "inner_loops"):
            h = hh[scope]
            valu = {k[5:]: v for k, v in h.items() if k.startswith("valu:")}
            nv = sum(valu.values())
            hh[scope + "_valu"] = nv
            hh[scope + "_salu_per_valu"] = (h.get("salu", 0) + h.get("branch", 0)) / nv if nv else None
            if cost and nv:
                hh[scope + "_cycles_per_valu"] = sum(cost.get(k, cost["and_or_b32"]) * v for k, v in valu.items()) / nv
    if args.json:
        print(json.dumps(result, indent=1, sort_keys=True))
        return
    for name, hh in sorted(result.items()):
        print("==", name)
        for scope in ("all", "inner_loops"):
            h = hh[scope]
            tot = sum(h.values())
            parts = ", ".join("%s %d" % kv for kv in sorted(h.items(), key=lambda kv: -kv[1]))
            extra = ""
            if hh.get(scope + "_cycles_per_valu"):
                extra = "  -> %.2f cycles per vector instruction (mix-weighted)" % hh[scope + "_cycles_per_valu"]
            print("  %-11s %5d instr, %5d vector, %.2f scalar per vector%s\n      %s"
                  % (scope, tot, hh[scope + "_valu"], hh[scope + "_salu_per_valu"] or 0.0, extra, parts))


if __name__ == "__main__":
    main()
