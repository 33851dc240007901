#!/usr/bin/env python3
"""Phase durations (cycles, mean over the waves) from the in-kernel stamps `tools/kbench.py --stamps` prints for the
four-plane recurrence kernels (CG_X & 64 builds; recurrence4.hip and recurrence_ord.hip): gather of every step, rotate,
copy-out, barriers, turn-over / group boundary."""
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

def ordered_kernel(name, col):
    """cheb_ord_kernel (recurrence_ord.hip): 0 group start, 1 staged, 2 image complete, 4s+0 gather of step s done, 4s+1
    barrier behind it (adjoint: G_j requested in between), 4s+2 rotated, 4s+3 step closed, 40 last gather done, 41 next
    group's input requested and the last slab stored."""
    steps = sorted({k // 4 for k in col if 4 <= k < 40})
    out = ['stage %.1fk' % ((col[2] - col[0]).mean() / 1e3)]
    start = col[2]
    for s in steps:
        f = 4 * s
        if f in col:
            out.append('gather%d %.1fk' % (s, (col[f] - start).mean() / 1e3))
        if f + 1 in col and f in col:
            out.append('request+barrier%d %.1fk' % (s, (col[f + 1] - col[f]).mean() / 1e3))
        if f + 2 in col and f + 1 in col:
            out.append('rotate%d %.1fk' % (s, (col[f + 2] - col[f + 1]).mean() / 1e3))
        if f + 3 in col:
            start = col[f + 3]
    if 40 in col and 41 in col:
        out.append('boundary %.1fk' % ((col[41] - col[40]).mean() / 1e3))
        out.append('group %.1fk' % ((col[41] - col[0]).mean() / 1e3))
    print('%-24s' % name, '  '.join(out))


for name, ids, t in blocks(sys.argv[1]):
    t = np.where(t < 0, np.nan, t)
    col = {k: t[:, j] for j, k in enumerate(ids)}
    if not any(24 < k < 40 for k in ids):                       # the ordered kernel has no gather-end ids 25..
        ordered_kernel(name, col)
        continue
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
