#!/usr/bin/env python3
"""Instruction mix of the big loops of one kernel in an ISA file (hipcc -S --cuda-device-only).
usage: loop_mix2.py <file.s> <substring of the mangled kernel name> [min instructions]"""
import re, sys
from collections import Counter
txt = open(sys.argv[1]).read().splitlines()
sub = sys.argv[2]
minin = int(sys.argv[3]) if len(sys.argv) > 3 else 800
start = next(i for i, l in enumerate(txt) if l.startswith('_Z') and sub in l and ':' in l)
end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))
body = txt[start:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        labels[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
CHEAP = ('v_add_f32_e32', 'v_sub_f32_e32', 'v_mul_f32_e32', 'v_fma_f32', 'v_fmac_f32_e32', 'v_trunc_f32_e32', 'v_add_u32', 'v_sub_u32_e32',
         'v_lshrrev_b32', 'v_or_b32', 'v_xor_b32', 'v_mov_b32_e32', 'v_sub_f32_e64', 'v_add_f32_e64', 'v_mul_f32_e64', 'v_and_b32')
seen = set()
for (a, b) in sorted(loops, key=lambda t: t[0]):
    if a in seen:
        continue
    # outermost loop with this head
    b = max(bb for (aa, bb) in loops if aa == a)
    seen.add(a)
    c = Counter()
    for l in body[a:b + 1]:
        l = l.strip()
        if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
            continue
        c[l.split()[0]] += 1
    tot = sum(c.values())
    if tot < minin:
        continue
    cls = lambda pre: sum(n for o, n in c.items() if o.startswith(pre))
    valu = cls('v_')
    cheap = sum(n for o, n in c.items() if o in CHEAP)
    print(f"loop {a}-{b}: instrs {tot} valu {valu} (cheap {cheap}, other {valu - cheap}; ~{cheap * 2.5 + (valu - cheap) * 4.4:.0f} cyc) salu {cls('s_')} "
          f"vmem {cls(('global_', 'buffer_', 'flat_'))} scratch {cls('scratch_')} ds {cls('ds_')} barrier {c['s_barrier']}")
    print("    ", c.most_common(16))
