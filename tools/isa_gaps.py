#!/usr/bin/env python3
"""Static look at a kernel's ISA (hipcc -S --cuda-device-only): instruction histogram and the number of instructions
between consecutive MFMAs.  usage: isa_gaps.py file.s kernel-name-substring [mfma-mnemonic-prefix [index count]]"""
import re
import sys
from collections import Counter
s = open(sys.argv[1]).read()
key = sys.argv[2]
pref = sys.argv[3] if len(sys.argv) > 3 else "v_mfma"
for m in re.finditer(r"^(_Z\w+):", s, flags=re.M):
    name = m.group(1)
    if key not in name:
        continue
    i = m.start()
    body = s[i:s.index('s_endpgm', i)]
    lines = [l.strip() for l in body.split('\n')[1:] if l.strip() and not l.strip().startswith((';', '.')) and not re.match(r"^[\w.$]+:", l.strip())]
    print(name, len(lines), 'instructions')
    c = Counter(l.split()[0] for l in lines)
    print('  ' + ', '.join(f'{v} {k}' for k, v in c.most_common(36)))
    idx = [n for n, l in enumerate(lines) if l.startswith(pref)]
    gaps = [b - a - 1 for a, b in zip(idx, idx[1:])]
    print('  mfma', len(idx), 'gap histogram', sorted(Counter(gaps).items()))
    if len(sys.argv) > 4:
        k = idx[int(sys.argv[4])]
        print('\n'.join('    ' + l for l in lines[k - 2:k + int(sys.argv[5])]))
