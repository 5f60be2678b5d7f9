#!/usr/bin/env python3
"""Dev aid: print the instruction mix and issue-order sketch of the MFMA-bearing basic blocks of a
kernel in a hipcc -save-temps .s file.  usage: isa_loop.py file.s <substring of mangled name>"""
import re, sys
from collections import Counter

def cls(o):
    if o.startswith('v_mfma'): return 'M'
    if 'accvgpr' in o: return 'a'
    if 'add_f32' in o: return '+'
    if o.startswith('ds_'): return 'd'
    if o == 's_waitcnt': return 'w'
    if o == 's_nop': return 'n'
    if o.startswith('global_'): return 'g'
    if o.startswith('scratch'): return 'S'
    if o == 's_barrier': return 'B'
    return '.'

s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    i = s.index(re.search(r'^' + re.escape(name) + r'[A-Za-z0-9_]*:', s, re.M).group(0))
    j = s.index('.end_amdhsa_kernel', i)
    body = s[i:j]
    blocks = re.split(r'\n(\.LBB\d+_\d+):', body)
    for k in range(1, len(blocks), 2):
        lab, txt = blocks[k], blocks[k + 1]
        seq = re.findall(r'^\s+([a-z_0-9]+)', txt, re.M)
        ops = Counter(seq)
        if any(o.startswith('v_mfma') for o in ops):
            print(name, lab, {k: v for k, v in ops.items() if v >= 4})
            print(''.join(cls(o) for o in seq)[:900])
