"""Scan a gfx950 assembly listing for the distance, in wait states, between every f64 MFMA and the first later
instruction that reads or overwrites its result registers (other than the next MFMA of the same accumulation chain).
Prints, per (producer, consumer kind), the smallest distance found -- what the compiler's hazard tables granted.
  python tools/mfma_hazard_scan.py file.s [kernel-name-substring]"""
import re
import sys
from collections import defaultdict


def regs(tok):
    """register tokens -> set of (file, index)"""
    out = set()
    for m in re.finditer(r'\b([av])\[(\d+):(\d+)\]|\b([av])(\d+)\b', tok):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def main():
    lines = open(sys.argv[1]).read().split('\n')
    want = sys.argv[2] if len(sys.argv) > 2 else None
    ins = []  # (lineno, mnemonic, operands)
    active = want is None
    for no, ln in enumerate(lines, 1):
        s = ln.split(';')[0].strip()
        if want and s.endswith(':') and not s.startswith('.L'):
            active = want in s
        if not active or not s or s.startswith('.') or s.endswith(':'):
            continue
        parts = s.split(None, 1)
        ins.append((no, parts[0], parts[1] if len(parts) > 1 else ''))
    best = defaultdict(lambda: (10 ** 9, None))
    for i, (no, mn, ops) in enumerate(ins):
        if not mn.startswith('v_mfma_f64'):
            continue
        dst = regs(ops.split(',')[0])
        ws = 0
        for j in range(i + 1, min(i + 200, len(ins))):
            no2, mn2, ops2 = ins[j]
            if mn2 == 's_nop':
                ws += int(ops2.strip() or 0) + 1
                continue
            if mn2.startswith('s_cbranch') or mn2 == 's_branch' or mn2 == 's_endpgm':
                break  # control flow: not followed
            opl = ops2.split(',')
            touched = regs(ops2)
            if touched & dst:
                if mn2.startswith('v_mfma'):
                    srcc = regs(opl[3]) if len(opl) > 3 else set()
                    srcab = regs(opl[1]) | regs(opl[2]) if len(opl) > 2 else set()
                    kind = 'mfma-srcAB' if srcab & dst else ('mfma-srcC-same' if srcc == dst and regs(opl[0]) == dst else 'mfma-srcC-overlap')
                else:
                    wr = regs(opl[0]) & dst
                    rd = set().union(*[regs(o) for o in opl[1:]]) & dst if len(opl) > 1 else set()
                    kind = mn2.split('_e')[0] + ('(read)' if rd else '(overwrite)')
                    if mn2.startswith(('ds_', 'global_', 'scratch_', 'buffer_', 'flat_')):
                        kind = mn2 + '(mem)'
                key = (mn, kind)
                if ws < best[key][0]:
                    best[key] = (ws, (no, no2))
                break
            ws += 1
    for (mn, kind), (ws, where) in sorted(best.items()):
        print(f"{mn:28s} -> {kind:40s} min wait states {ws:3d}   lines {where}")


if __name__ == '__main__':
    main()
