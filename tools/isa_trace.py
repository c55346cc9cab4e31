#!/usr/bin/env python3
"""Prints the instruction-class sequence of the biggest loop of a kernel: M mfma, R ds_read, W ds_write, G global/buffer load,
T global store/atomic, v VALU, e v_exp, a accvgpr move, s SALU, | s_waitcnt (with counts), B barrier, n nop, X scratch."""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end', s, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    lines = body.split('\n')
    labels = {mm.group(1): i for i, l in enumerate(lines) for mm in [re.match(r'^(\.LBB\d+_\d+):', l)] if mm}
    best = None
    for i, l in enumerate(lines):
        mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            a, b = labels[mm.group(1)], i
            nm = sum(1 for x in lines[a:b] if 'v_mfma' in x)
            if best is None or nm > best[0] or (nm == best[0] and b - a < best[2] - best[1]):
                best = (nm, a, b)
    nm, a, b = best
    out = []
    for l in lines[a:b]:
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma'): out.append('M')
        elif op.startswith('ds_read') or op.startswith('ds_load'): out.append('R')
        elif op.startswith('ds_write') or op.startswith('ds_store'): out.append('W')
        elif op.startswith('ds_'): out.append('D')
        elif op.startswith('global_load') or op.startswith('buffer_load'): out.append('G')
        elif op.startswith('global_') or op.startswith('buffer_'): out.append('T')
        elif op.startswith('scratch_'): out.append('X')
        elif op.startswith('s_waitcnt'):
            mm = re.search(r'lgkmcnt\((\d+)\)', t); vv = re.search(r'vmcnt\((\d+)\)', t)
            out.append('|' + ('l%s' % mm.group(1) if mm else '') + ('v%s' % vv.group(1) if vv else '') + '|')
        elif op.startswith('s_barrier'): out.append(' B ')
        elif op.startswith('s_nop'): out.append('n')
        elif op.startswith('v_accvgpr'): out.append('a')
        elif op.startswith('v_exp'): out.append('e')
        elif op.startswith('v_'): out.append('v')
        elif op.startswith('s_'): out.append('s')
        else: out.append('?')
    txt = ''.join(out)
    print(name, f"loop [{a}-{b}] mfma={nm}")
    for i in range(0, len(txt), 160):
        print(txt[i:i + 160])
