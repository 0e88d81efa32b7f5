#!/usr/bin/env python3
"""Static instruction counts of the loops of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).
usage: tools/isa_loops.py file.s <substring of the mangled kernel name>"""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r'^(\w*' + re.escape(sys.argv[2]) + r'\w*):', s, re.M)
start = m.end(); end = s.index('.Lfunc_end', start)
body = s[start:end].splitlines()
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm: labels[mm.group(1)] = i
tot = [x.strip() for x in body if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
print(m.group(1)[:60], 'total insts', len(tot))
for i, l in enumerate(body):
    mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if mm:
        t = mm.group(1) or mm.group(2)
        if t in labels and labels[t] < i:
            seg = body[labels[t]:i + 1]
            ins = [x.strip() for x in seg if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
            v = sum(1 for x in ins if x.startswith('v_')); sa = sum(1 for x in ins if x.startswith('s_'))
            print(t, 'lines', labels[t], i, 'insts', len(ins), 'valu', v, 'salu', sa, 'mem', len(ins) - v - sa)
