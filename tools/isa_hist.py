#!/usr/bin/env python3
"""Instruction histogram of the kernels in a hipcc -S listing (static counts, per kernel).
usage: isa_hist.py file.s [substring-of-kernel-name]"""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
cur = None
hist = {}
for ln in src:
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        cur = m.group(1); hist[cur] = collections.Counter(); continue
    if ln.startswith('.Lfunc_end'):
        cur = None; continue
    if cur is None: continue
    t = ln.strip()
    if not t or t.startswith(('.', ';', '//')) or t.endswith(':'): continue
    op = t.split()[0]
    hist[cur][op] += 1
def cls(op):
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'): return 'valu'
    return 'other'
for k, h in hist.items():
    if want not in k: continue
    tot = sum(h.values())
    print('==', k[:150], 'total', tot)
    by = collections.Counter()
    for op, n in h.items(): by[cls(op)] += n
    print('  ', dict(by))
    for op, n in h.most_common(45): print(f'   {n:6d} {op}')
