#!/usr/bin/env python3
"""Build step: close the MFMA-read / load-write hazard the gfx950 compiler does not know (ROCm 7.2, clang 22).

What goes wrong.  An MFMA reads its SrcC registers while it runs, not when it issues.  When the register allocator
gives those registers to the destination of a LOAD that follows within a few instructions --
      v_mfma_f64_16x16x4_f64 a[16:23], v[26:27], v[34:35], a[24:31]
      ds_read_b64  v[16:17], v160 offset:20480
      ds_read_b128 a[24:27], v183 offset:128          <- SrcC of the MFMA above
-- the LDS data can land before the MFMA has read its operand, and the product is computed on the loaded values.  The
hazard recogniser inserts wait states between an MFMA and a VALU instruction that overwrites its operands, but has no entry
for loads (on earlier chips every load's latency exceeded every MFMA's run time; gfx950's f64 16x16x4 runs 16 passes = 64
cycles, an idle LDS answers in about that).  Found with tools/f16_asm_bisect.py as the cause of the wrong -O1 builds of
ey_fused16.hip; measured with tools/mfma_war_probe.hip (profiles/r04_mfma_war_probe.txt): an LDS load
issued directly behind v_mfma_f64_16x16x4_f64 into its SrcC registers corrupts the product, on an idle CU and on a busy
one, every time; one s_nop between them was enough in the probe, another LDS instruction between them (as above) is not;
global loads, v_mfma_f64_4x4x4_4b_f64, v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16 could not be provoked at any
distance.

What this does.  On the device assembly of a translation unit, for every f64 MFMA it walks the following instructions
along every control-flow path for passes + 2 wait states -- counting only what holds the wave's issue back: s_nop, VALU,
SALU, other MFMAs; memory instructions count nothing -- and a load (LDS, global, scratch, buffer, flat) whose destination
overlaps the MFMA's SrcC or vDst registers inside that window gets `s_nop`s in front of it that make up the difference
(the probe's boundary is one wait state; the whole run time of the MFMA is the margin, and the pattern is rare).
`--check` only reports (exit code 1 if anything is found): tests/test_build_hazards.py runs it on the shipped assembly.

    mfma_load_hazard.py in.s out.s [--report]        mfma_load_hazard.py --check in.s
"""
import os
import re
import sys

# wait states that must lie between an MFMA and a load into its SrcC / vDst registers: the passes of the instruction (4
# cycles each) plus a margin of two.  f64 16x16x4 is the measured case; f64 4x4x4 (not provoked by the probe) is covered
# with its own run time because it costs nothing.  The f32 / bf16 MFMAs, measured safe at distance zero, are left alone.
PASSES = [
    (re.compile(r"v_mfma_f64_16x16x4"), 16),
    (re.compile(r"v_mfma_f64_4x4x4"), 4),
]
MARGIN = 2
MEMORY = re.compile(r"^(ds_|global_|scratch_|buffer_|flat_|s_load|s_buffer_load|s_waitcnt|s_store|s_dcache)")

LOAD = re.compile(r"^(ds_(read|load|bpermute|permute|swizzle|consume|append|ordered)|ds_\w+_rtn|global_load|scratch_load|"
                  r"buffer_load|flat_load|global_atomic\w*\s|buffer_atomic|flat_atomic)")
LDS_DMA = re.compile(r"_lds_|\blds\b")
REG = re.compile(r"\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b")
END = re.compile(r"^(s_endpgm|s_setpc_b64|s_swappc_b64|s_trap)")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def window(mnemonic):
    for rx, passes in PASSES:
        if rx.match(mnemonic):
            if passes == 16 and os.environ.get("EY_HAZARD_W16"):   # A/B builds only (tools/ab_hazard_window.sh)
                return int(os.environ["EY_HAZARD_W16"])
            return passes + MARGIN
    return 0


def parse(lines):
    """-> (instructions [(line index, mnemonic, operand text)], label name -> index of the next instruction)."""
    ins, labels, pending = [], {}, []
    for no, raw in enumerate(lines):
        s = raw.split(";")[0].strip()
        if not s:
            continue
        if s.endswith(":") and not s.startswith("."):
            pending.append(s[:-1])
            continue
        if re.match(r"^\.L\w+:$", s) or re.match(r"^\.LBB\w+:$", s):
            pending.append(s[:-1])
            continue
        if s.startswith(".") or not raw.startswith(("\t", " ")):
            continue
        parts = s.split(None, 1)
        for lab in pending:
            labels[lab] = len(ins)
        pending = []
        ins.append((no, parts[0], parts[1] if len(parts) > 1 else ""))
    return ins, labels


def find(lines):
    """-> {instruction index of a load: s_nop wait states to put in front of it}, list of findings for the report."""
    ins, labels = parse(lines)
    need, found = {}, []
    for i, (no, mn, ops) in enumerate(ins):
        if not (mn.startswith("v_mfma") or mn.startswith("v_smfmac")):
            continue
        win = window(mn)
        opl = [o.strip() for o in ops.split(",")]
        if len(opl) < 4:
            continue
        protect = regs(opl[0]) | regs(opl[3])   # vDst and SrcC
        # walk every path for `win` wait states; state = (next instruction index, wait states so far)
        best = {}
        stack = [(i + 1, 0)]
        while stack:
            j, ws = stack.pop()
            while j < len(ins) and ws < win:
                if best.get(j, 1 << 30) <= ws:
                    break
                best[j] = ws
                no2, mn2, ops2 = ins[j]
                if mn2 == "s_nop":
                    ws += int(ops2.strip() or "0", 0) + 1
                    j += 1
                    continue
                if END.match(mn2):
                    break
                if LOAD.match(mn2 + " ") and not LDS_DMA.search(mn2):
                    dst = regs(ops2.split(",")[0])
                    if dst & protect:
                        pad = win - ws
                        if pad > need.get(j, 0):
                            need[j] = pad
                        found.append((no + 1, f"{mn} {ops}", no2 + 1, f"{mn2} {ops2}", ws, win))
                        ws = win  # the pad puts everything behind this load outside the window
                        break
                if MEMORY.match(mn2):   # issues beside the vector pipe: no wait state
                    j += 1
                    continue
                if mn2 == "s_branch":
                    tgt = labels.get(ops2.strip())
                    if tgt is None:
                        break
                    j = tgt
                    ws += 1
                    continue
                if mn2.startswith("s_cbranch"):
                    tgt = labels.get(ops2.strip())
                    if tgt is not None:
                        stack.append((tgt, ws + 1))
                ws += 1
                j += 1
    return ins, need, found


def fix(lines):
    ins, need, found = find(lines)
    out = list(lines)
    for j in sorted(need, reverse=True):
        no = ins[j][0]
        pad, nops = need[j], []
        while pad > 0:
            k = min(pad, 16)
            nops.append(f"\ts_nop {k - 1}  ; mfma_load_hazard.py: a load into the operands of an MFMA still running")
            pad -= k
        out[no:no] = nops
    return out, found


def main(argv):
    if len(argv) >= 2 and argv[0] == "--check":
        _, _, found = find(open(argv[1]).read().split("\n"))
        for f in found:
            print(f"{argv[1]}:{f[0]}: {f[1]}\n    line {f[2]}: {f[3]}   ({f[4]} of {f[5]} wait states)")
        print(f"{argv[1]}: {len(found)} load(s) into the operands of a running MFMA")
        return 1 if found else 0
    if len(argv) < 2:
        print(__doc__)
        return 2
    lines = open(argv[0]).read().split("\n")
    out, found = fix(lines)
    open(argv[1], "w").write("\n".join(out))
    if "--report" in argv:
        for f in found:
            print(f"  line {f[0]}: {f[1]}  ->  line {f[2]}: {f[3]}  ({f[4]} of {f[5]} wait states)")
    print(f"mfma_load_hazard: {argv[0]}: {len(found)} hazard(s) padded")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
