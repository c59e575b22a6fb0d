#!/usr/bin/env python3
"""Skeleton of a kernel's ISA after its last s_barrier (the tile loop of the streaming kernels):
waits on vmcnt, branches, runs of loads / stores / MFMAs.  usage: isa_loop.py file.s mangled_name"""
import re, sys
s = open(sys.argv[1]).read()
name = sys.argv[2]
a = s.index(name + ':'); b = s.index('.Lfunc_end', a)
body = s[a:b].split('\n')
out = []; prev = None; cnt = 0
for i, l in enumerate(body):
    t = l.strip(); m = None
    if re.search(r's_waitcnt.*vmcnt', t): m = 'W ' + t
    elif t.startswith('.LBB'): m = t.split(';')[0].strip() + (' LOOPHDR' if 'Loop Header' in t else '')
    elif 's_cbranch' in t: m = t
    elif 'global_load' in t or 'buffer_load' in t: m = 'LOAD'
    elif 'global_store' in t or 'buffer_store' in t: m = 'STORE'
    elif 'v_mfma' in t: m = 'MFMA'
    elif 's_barrier' in t: m = 'BARRIER'
    if m is None: continue
    if m == prev and m in ('LOAD', 'STORE', 'MFMA'): cnt += 1; continue
    if prev in ('LOAD', 'STORE', 'MFMA'): out[-1] += ' x%d' % cnt
    out.append('%d %s' % (i, m)); prev = m; cnt = 1
bars = [k for k, o in enumerate(out) if 'BARRIER' in o]
idx = bars[int(sys.argv[3])] if len(sys.argv) > 3 else (bars[-1] if bars else 0)
print('\n'.join(out[idx:idx + int(sys.argv[4]) if len(sys.argv) > 4 else idx + 100]))
