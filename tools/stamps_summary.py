#!/usr/bin/env python3
"""Phase durations (cycles, mean over the waves) from the in-kernel stamps `tools/kbench.py --stamps` prints for the
four-plane recurrence kernel (CG_X & 64 builds): gather of every step, rotate, copy-out, barriers, turn-over."""
import sys
import numpy as np

def blocks(path):
    lines = open(path).read().splitlines()
    i = 0
    while i < len(lines):
        if lines[i].strip().startswith('id '):
            ids = [int(x) for x in lines[i].split()[1:]]
            rows = []
            i += 1
            while i < len(lines) and lines[i].strip().startswith('w'):
                rows.append([int(x) for x in lines[i].split()[1:]])
                i += 1
            name = lines[i].split()[0] if i < len(lines) else '?'
            yield name, ids, np.array(rows, float)
        else:
            i += 1

for name, ids, t in blocks(sys.argv[1]):
    col = {k: t[:, j] for j, k in enumerate(ids)}
    K = max(k for k in ids if 24 < k < 40) - 24 + 1          # steps 1..K-1 have gather-end stamps 25..
    out = []
    start = col[2]
    for s in range(1, K):
        end = col[24 + s]
        out.append('gather%d %.1fk' % (s, (end - start).mean() / 1e3))
        fs = 4 * s
        if fs + 3 in col:
            out.append('out+wait%d %.1fk' % (s, (col[fs + 1] - end).mean() / 1e3))
            out.append('rotate%d %.1fk' % (s, (col[fs + 2] - col[fs + 1]).mean() / 1e3))
            start = col[fs + 3]
    out.append('turnover %.1fk' % ((col[41] - col[40]).mean() / 1e3))
    out.append('group %.1fk' % ((col[41] - col[2]).mean() / 1e3))
    print('%-24s' % name, '  '.join(out))
