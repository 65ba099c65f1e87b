#!/usr/bin/env python3
"""Instruction-class picture of one kernel's loops in a device assembly file: one line per bf16 / f32 32x32 MFMA with what
follows it.  usage: isa_classes.py file.s kernel-name-substring [min-lines-of-a-loop]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 300
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^_Z\w+:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(body):
    m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] >= minlen:
        loops.append((labels[m.group(1)], i))


def cls(op):
    if op.startswith("v_mfma_f32_32x32x16"): return "\nM16 |"
    if op.startswith("v_mfma_f32_32x32x2"): return "\nM2  |"
    if op.startswith("v_mfma"): return " m4"
    if op.startswith("v_accvgpr_read"): return " ar"
    if op.startswith("v_accvgpr_write"): return " aw"
    if op.startswith("v_accvgpr"): return " am"
    if re.match(r"v_(exp|rcp|log|sqrt|rsq)", op): return " T"
    if op.startswith("v_cvt_pk"): return " C"
    if op.startswith("v_pk_"): return " PK"
    if op.startswith("v_permlane") or "dpp" in op: return " X"
    if op.startswith("v_"): return " v"
    if op.startswith("ds_read") or op.startswith("ds_load"): return " LR"
    if op.startswith("ds_write") or op.startswith("ds_store"): return " LW"
    if op.startswith("scratch_"): return " SCR"
    if op.startswith("global_") or op.startswith("buffer_"): return " G"
    if op.startswith("s_waitcnt"): return " W"
    if op.startswith("s_nop"): return " n"
    if op.startswith("s_"): return " s"
    return ""


for a, b in loops:
    ops = [l.split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith(";") and not l.strip().startswith(".")]
    waits = [l.strip() for l in body[a:b + 1] if "s_waitcnt" in l]
    print(f"==== loop lines {a}..{b} ({len(ops)} instructions, {sum(o.startswith('v_mfma_f32_32x32x16') for o in ops)} bf16 MFMAs, "
          f"{sum(o.startswith('v_accvgpr') for o in ops)} accvgpr, {sum(o.startswith('scratch') for o in ops)} scratch, {len(waits)} waits)")
    print("".join(cls(o) for o in ops))
