#!/usr/bin/env python3
"""Memory-operation sequence of a kernel in an assembly dump (hipcc -S --cuda-device-only): runs of global / LDS / scratch
loads and stores between barriers, with the instruction count of every run.   asm_ops.py file.s kernel-name-substring"""
import sys
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and sys.argv[2] in l and l.rstrip().split(':')[0] == l.split(':')[0] and ':' in l)
ev = []
n = 0
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith('s_endpgm'):
        break
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    n += 1
    op = t.split()[0]
    kind = None
    if op == 's_barrier': kind = 'BAR'
    elif op.startswith('scratch_store'): kind = 'ss'
    elif op.startswith('scratch_load'): kind = 'sl'
    elif op.startswith('global_load'): kind = 'gl'
    elif op.startswith('global_store'): kind = 'gs'
    elif op.startswith('ds_read') or op.startswith('ds_load'): kind = 'dr'
    elif op.startswith('ds_write') or op.startswith('ds_store'): kind = 'dw'
    if kind:
        ev.append((n, kind))
out = []
last, cnt, at = None, 0, 0
for i, e in ev:
    if e == last:
        cnt += 1
    else:
        if last:
            out.append('%s%d@%d' % (last, cnt, at))
        last, cnt, at = e, 1, i
out.append('%s%d@%d' % (last, cnt, at))
print(n, 'instructions')
print(' '.join(out))
