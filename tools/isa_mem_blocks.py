#!/usr/bin/env python3
"""Basic blocks of one kernel that issue global loads / stores, as a compressed trace (GL load, GS store, W(...) s_waitcnt,
SL / SS scratch, v / s other vector / scalar instructions): shows whether an epilogue's loads are batched or every element
is a serial round trip.   usage: isa_mem_blocks.py file.s mangled_kernel_name [min_ops]"""
import re, sys
s = open(sys.argv[1]).read(); name = sys.argv[2]; lim = int(sys.argv[3]) if len(sys.argv) > 3 else 8
i = s.index(name + ':'); j = s.index('s_endpgm', i)
cur = None; blocks = {}
for ln in s[i:j].split('\n'):
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m: cur = m.group(1); blocks[cur] = []; continue
    t = ln.strip()
    if cur and t and not t.startswith(';') and not t.startswith('.'): blocks[cur].append(t)
def kind(t):
    op = t.split()[0]
    if op.startswith('global_load'): return 'GL'
    if op.startswith('global_store'): return 'GS'
    if op.startswith('s_waitcnt'): return 'W(' + t.split(None, 1)[1].split(';')[0].strip() + ')'
    if op.startswith('scratch_load'): return 'SL'
    if op.startswith('scratch_store'): return 'SS'
    if op.startswith('v_'): return 'v'
    if op.startswith('s_'): return 's'
    return op
for b, ins in blocks.items():
    ngl = sum(1 for t in ins if t.startswith('global_load')); ngs = sum(1 for t in ins if t.startswith('global_store'))
    if ngl >= lim or ngs >= lim:
        out = []; prev = None; n = 0
        for t in ins:
            k = kind(t)
            if k == prev: n += 1
            else:
                if prev: out.append(prev + (f'x{n}' if n > 1 else ''))
                prev = k; n = 1
        out.append(prev + (f'x{n}' if n > 1 else ''))
        print(b, 'loads', ngl, 'stores', ngs, 'len', len(ins)); print(' '.join(out)[:420]); print()
