#!/usr/bin/env python3
"""Counts instruction classes inside every loop of a kernel in a hipcc -S dump.  usage: isa_loops.py file.s symbol-substring"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2]
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end', s, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    lines = body.split('\n')
    labels = {}
    for i, l in enumerate(lines):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm:
            labels[mm.group(1)] = i
    print(name, "lines", len(lines))
    pats = dict(mfma=r'v_mfma', ds_read=r'ds_read', ds_write=r'ds_write', bperm=r'ds_bpermute', gload=r'global_load|buffer_load',
                gstore=r'global_store|global_atomic', valu=r'^\s+v_(?!mfma)', salu=r'^\s+s_(?!waitcnt|barrier|nop)', waitcnt=r's_waitcnt',
                barrier=r's_barrier', nop=r's_nop', scratch=r'scratch_', accvgpr=r'v_accvgpr', exp=r'v_exp', cvt=r'v_cvt')
    for i, l in enumerate(lines):
        mm = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            a, b = labels[mm.group(1)], i
            seg = lines[a:b]
            c = {k: sum(1 for x in seg if re.search(p, x)) for k, p in pats.items()}
            print(f"  loop {mm.group(1)} [{a}-{b}] n={b - a}: " + " ".join(f"{k}={v}" for k, v in c.items()))
