#!/usr/bin/env python3
"""Issue-cost census of the biggest loop of one kernel in /tmp/ako_plan.s (written by scripts/kernel_regs.py),
priced with the per-class issue costs measured by scripts/valu_rates.hip on MI355X.
usage: loop_cost.py <substring of the mangled kernel name>"""
import re, sys
from collections import Counter
FAST = {'v_fma_f32', 'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_ashrrev_i32', 'v_lshlrev_b32', 'v_lshrrev_b32', 'v_add_f32', 'v_sub_f32', 'v_subrev_f32',
        'v_mul_f32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_fmac_f32', 'v_mov_b32', 'v_max_f32', 'v_min_f32', 'v_cndmask_b32',
        'v_add_co_u32', 'v_addc_co_u32', 'v_mul_u32_u24', 'v_mul_i32_i24', 'v_not_b32', 'v_fmaak_f32', 'v_fmamk_f32'}
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
import os
sel = int(os.environ.get('LOOP', '-1'))
loops = sorted(set(loops))
print('big loops:', [(i, a, b, b - a) for i, (a, b) in enumerate(loops) if b - a > 1500][:int(os.environ.get('SHOW', '6'))])
a, b = max(loops, key=lambda t: t[1] - t[0]) if sel < 0 else loops[sel]
c = Counter()
for l in body[a:b + 1]:
    l = l.strip()
    if not l or l.startswith(('.', ';', '//')) or l.endswith(':'):
        continue
    op = l.split()[0]
    if op.startswith('v_') and ('dpp' in l or 'row_' in l or 'wave_' in l):
        op += ':dpp'
    c[op] += 1
fast = sum(n for o, n in c.items() if o.replace('_e32', '').replace('_e64', '') in FAST)
valu = sum(n for o, n in c.items() if o.startswith('v_'))
slow = valu - fast
print(f"{body[0].split(':')[0][:80]} loop {a}-{b}: valu {valu} fast {fast} slow {slow} -> est cycles {fast * 2.4 + slow * 4.15:.0f}; other {sum(c.values()) - valu}")
print("  slow:", [(o, n) for o, n in c.most_common() if o.startswith('v_') and o.replace('_e32', '').replace('_e64', '') not in FAST])
print("  fast:", [(o, n) for o, n in c.most_common() if o.replace('_e32', '').replace('_e64', '') in FAST])
print("  other:", [(o, n) for o, n in c.most_common() if not o.startswith('v_')][:14])
