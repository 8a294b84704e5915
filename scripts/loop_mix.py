#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in /tmp/ako_plan.s (written by scripts/kernel_regs.py).
usage: loop_mix.py <substring of the mangled kernel name>"""
import re, sys
from collections import Counter
txt = open('/tmp/ako_plan.s').read().splitlines()
sub = sys.argv[1]
start = next(i for i, l in enumerate(txt) if l.startswith('_ZN3ako') and sub in l and l.split(':')[0].endswith('E'))
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
print(body[0].split(':')[0], "lines", len(body), "loops", loops)
for (a, b) in loops:
    c = Counter()
    for l in body[a:b + 1]:
        l = l.strip()
        if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
            continue
        c[l.split()[0]] += 1
    tot = sum(c.values())
    cls = lambda pre: sum(n for o, n in c.items() if o.startswith(pre))
    print(f"loop {a}-{b}: instrs {tot} valu {cls('v_')} salu {cls('s_')} vmem {cls(('global_', 'buffer_', 'flat_'))} ds {cls('ds_')}")
    print("    ", c.most_common(24))
