#!/usr/bin/env python3
"""Compressed instruction-class pattern of a range of lines of an AMDGPU .s file (MFMA / V (vector ALU) / DS / VMEM / WAIT / ...):
   python tools/isa_pattern.py file.s first_line last_line"""
import sys, re
lines = open(sys.argv[1]).read().splitlines()[int(sys.argv[2]) - 1:int(sys.argv[3])]
out = []
for l in lines:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') and not t.startswith('.LBB'):
        continue
    op = t.split()[0]
    if op.startswith('.LBB'):
        c = '\n' + op
    elif 'mfma' in op: c = 'M'
    elif op.startswith('ds_'): c = 'D'
    elif op.startswith('global_') or op.startswith('buffer_'): c = 'G'
    elif op.startswith('v_cvt_f64') or op.startswith('v_add_f64') or op.startswith('v_fma_f64'): c = 'F'
    elif op.startswith('v_'): c = 'v'
    elif op == 's_waitcnt': c = 'W(' + t.split(None, 1)[1].replace('lgkmcnt', 'l').replace('vmcnt', 'vm') + ')'
    elif op == 's_nop': c = 'n'
    elif op == 's_barrier': c = '\nBARRIER\n'
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): c = ' ' + op + ' '
    elif op == 's_setprio': c = 'P' + t.split()[1]
    else: c = 's'
    out.append(c)
# run-length compress
res, prev, cnt = [], None, 0
for c in out + [None]:
    if c == prev and c in ('v', 's', 'M', 'D', 'F', 'n', 'G'):
        cnt += 1
    else:
        if prev is not None:
            res.append((prev if cnt == 1 else '%d%s' % (cnt, prev)))
        prev, cnt = c, 1
print(' '.join(res))
