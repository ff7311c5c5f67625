#!/usr/bin/env python3
"""Basic-block instruction census of one kernel in a hipcc -S listing.
   python tools/isa_blocks.py /tmp/pc_kernels.s _ZN3pck14trace16_kernelILi24ELb0E [min_instrs]"""
import re, collections, sys
lines = open(sys.argv[1]).read().split('\n')
pref = sys.argv[2]
mini = int(sys.argv[3]) if len(sys.argv) > 3 else 20
start = [i for i, l in enumerate(lines) if l.startswith(pref) and ':' in l.split(';')[0]][0]
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
blocks = []; cur = ['<entry>', []]
for l in lines[start + 1:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append(cur); cur = [l.split(':')[0], []]
    else:
        s = l.split(';')[0].strip()
        if s and not s.startswith('.'):
            cur[1].append(s)
blocks.append(cur)
for name, ins in blocks:
    c = collections.Counter(i.split()[0] for i in ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    if len(ins) >= mini:
        print(name, 'instrs', len(ins), 'VALU', valu, 'max3', c.get('v_pk_maximum3_f16', 0))
        print('   ', sorted(c.items(), key=lambda kv: -kv[1])[:40])
