#!/usr/bin/env python3
"""Load / MFMA / wait / barrier sequence of the kernels in a hipcc -S listing: shows at a glance whether a prefetch
ring survived the scheduler (many loads, then `w<N>` with large N) or was sunk to its uses (load, w1, load, w1 ...).
usage: python tools/isa_sequence.py file.s <kernel-name-substring>"""
import collections, re, subprocess, sys
txt = open(sys.argv[1]).read().split('\n')
cur, data = None, collections.OrderedDict()
for l in txt:
    m = re.match(r'^(_Z\S+):\s', l)
    if m:
        cur = m.group(1); data[cur] = []; continue
    if l.startswith('.Lfunc_end'):
        cur = None; continue
    if cur:
        data[cur].append(l)
names = subprocess.run(['c++filt'], input='\n'.join(data), capture_output=True, text=True).stdout.splitlines()
for (k, body), nm in zip(data.items(), names):
    nm = re.sub(r'\(anonymous namespace\)::|^void ', '', nm).split('(')[0]
    if sys.argv[2] not in nm:
        continue
    seq = []
    for l in body:
        t = l.strip()
        if t.startswith('global_load_dwordx4'): seq.append('L4')
        elif t.startswith('global_load_dwordx2'): seq.append('L2')
        elif t.startswith('global_load'): seq.append('L1')
        elif t.startswith('global_store') : seq.append('S')
        elif t.startswith('global_atomic'): seq.append('A')
        elif t.startswith('v_mfma'): seq.append('M')
        elif t.startswith('s_barrier'): seq.append('|')
        elif t.startswith('s_cbranch'): seq.append('br')
        elif t.startswith('s_waitcnt') and 'vmcnt' in t:
            seq.append('w' + re.search(r'vmcnt\((\d+)\)', t).group(1))
    out, prev, cnt = [], None, 0
    for x in seq + [None]:
        if x == prev:
            cnt += 1
        else:
            if prev:
                out.append(prev + ('x%d' % cnt if cnt > 1 else ''))
            prev, cnt = x, 1
    print(nm); print('   ', ' '.join(out)[:2500])
