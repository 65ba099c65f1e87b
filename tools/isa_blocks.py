#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in an assembly listing (hipcc -S --cuda-device-only):
usage: isa_blocks.py file.s mangled_kernel_name [--dump N]   (blocks without an MFMA are skipped; --dump prints the first
N lines of the block with the most MFMAs)"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
name = sys.argv[2]
dump = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[3] == "--dump" else 0
i = s.index(name + ":")
j = s.index("s_endpgm", i)
blocks, cur = [], None
for ln in s[i:j].split("\n"):
    m = re.match(r"^(\.LBB\d+_\d+):", ln)
    if m:
        cur = [m.group(1), []]
        blocks.append(cur)
        continue
    if cur is None:
        cur = ["entry", []]
        blocks.append(cur)
    t = ln.strip()
    if t and not t.startswith(";") and not t.startswith("."):
        cur[1].append(t)
KINDS = (("v_mfma_f32_32x32x2", "mfma_f32"), ("v_mfma_f32_4x4", "mfma4"), ("v_mfma", "mfma_other"), ("v_cvt_pk", "cvt"),
         ("v_pk_", "pk"), ("v_exp", "trans"), ("v_rcp", "trans"), ("v_log", "trans"), ("v_", "valu"), ("ds_", "ds"),
         ("s_waitcnt", "wait"), ("s_nop", "nop"), ("s_barrier", "barrier"), ("global_load", "gload"),
         ("global_store", "gstore"), ("scratch_", "scratch"), ("buffer_", "buffer"), ("s_", "salu"))
best = None
for b, ins in blocks:
    c = Counter()
    for t in ins:
        op = t.split()[0]
        for pre, k in KINDS:
            if op.startswith(pre):
                c[k] += 1
                break
    nm = c["mfma_f32"] + c["mfma4"] + c["mfma_other"]
    if nm == 0:
        continue
    print(b, dict(c), [t for t in ins if t.startswith("s_cbranch") or t.startswith("s_branch")][-2:])
    if best is None or nm > best[0]:
        best = (nm, ins)
if dump and best:
    print("\n".join(best[1][:dump]))
