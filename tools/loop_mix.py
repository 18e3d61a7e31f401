#!/usr/bin/env python3
"""Instruction mix of the loops of one function in an AMDGPU assembly listing (hipcc -S --cuda-device-only).

    python tools/loop_mix.py kernel.s [function-name-part] [min-instructions]

A loop = a backward branch (s_cbranch_* / s_branch to a label defined earlier in the same function); loops
are listed outermost-first by size with the wave-instruction counts the SQ counters see: VALU split into
fp64 arithmetic, DPP / lane moves, v_readlane / v_writelane (SGPR spill traffic), v_mov / v_cndmask and the
rest; LDS, scalar, scratch.  Static counts: an unrolled straight-line body executes each once per trip."""
import collections
import re
import sys


def classify(op, line):
    if op.startswith("scratch_") or op.startswith("buffer_") and "offen" in line:
        return "scratch"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    if not op.startswith("v_"):
        return "other"
    if op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
        return "valu:lane<->sgpr"
    if "dpp" in line or "permlane" in op:
        if re.match(r"v_(fma|mul|add|max|min)_f64", op):
            return "valu:f64"      # arithmetic that carries its own DPP operand does not exist for f64; kept for safety
        return "valu:dpp/permlane"
    if re.match(r"v_(fma|fmac|mul|add|max|min|rcp|rsq|sqrt|div_fmas|div_fixup|div_scale|ldexp|frexp_mant|log|trunc|floor|ceil|rndne|fract)_f64", op):
        return "valu:f64"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "valu:cmp"
    if op.startswith("v_mov") or op.startswith("v_cndmask") or op.startswith("v_accvgpr") or op.startswith("v_pk_mov"):
        return "valu:mov/select"
    return "valu:int/other"


def loops(path, part=""):
    """[(function name, first instruction, last instruction, Counter of classes, Counter of opcodes)] of every
    loop of the functions whose name contains `part`, largest first."""
    lines = open(path).read().split("\n")
    out = []
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if not m or part not in m.group(1):
            i += 1
            continue
        name, start = m.group(1), i
        while i < len(lines) and not lines[i].startswith(".Lfunc_end"):
            i += 1
        body = lines[start:i]
        labels, insts = {}, []
        for l in body:
            s = l.strip()
            if not s or s.startswith(";") or s.startswith("."):
                lm = re.match(r"^(\.LBB\w+):", s)
                if lm:
                    labels[lm.group(1)] = len(insts)
                continue
            if re.match(r"^_Z\w+:", s):
                continue
            insts.append(s)
        found = []
        for k, s in enumerate(insts):
            op = s.split()[0]
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = s.split()[-1]
                if tgt in labels and labels[tgt] <= k:
                    found.append((labels[tgt], k))
        for lo, hi in sorted(set(found), key=lambda r: r[0] - r[1]):
            c = collections.Counter()
            ops = collections.Counter()
            for s in insts[lo:hi + 1]:
                op = s.split()[0]
                c[classify(op, s)] += 1
                ops[op + ("_dpp" if "dpp" in s else "")] += 1
            out.append((name, lo, hi, c, ops, len(insts)))
    return out


def main():
    path = sys.argv[1]
    part = sys.argv[2] if len(sys.argv) > 2 else ""
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    last = None
    for name, lo, hi, c, ops, n_insts in loops(path, part):
        if name != last:
            print("== %s: %d instructions" % (name, n_insts))
            last = name
        if hi - lo < floor:
            continue
        valu = sum(n for k2, n in c.items() if k2.startswith("valu"))
        print("  loop @%d..%d: %d instructions, VALU %d" % (lo, hi, hi - lo + 1, valu))
        for k2, n in sorted(c.items(), key=lambda kv: -kv[1]):
            print("      %-20s %5d" % (k2, n))
        print("      top:", ", ".join("%s %d" % kv for kv in ops.most_common(14)))


if __name__ == "__main__":
    main()
