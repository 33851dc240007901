#!/usr/bin/env python3
"""Per-layer durations of the recurrence / contraction launches of the bench step from a rocprofv3 kernel trace CSV."""
import csv, sys, glob, collections, re
path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
seq = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    key = None
    m = re.search(r'cheb\w*_kernel<([^>]*)>', n)
    if m: key = 'rec_bwd' if m.group(1).split(',')[4].strip() == 'true' else 'rec_fwd'
    if 'contract_fwd' in n: key = 'c_fwd'
    if 'contract_bwd_x' in n: key = 'c_bwx'
    if 'contract_bwd_w' in n: key = 'c_bww'
    if key: seq[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, per in (('rec_fwd', 6), ('c_fwd', 6), ('rec_bwd', 5), ('c_bwx', 5), ('c_bww', 6)):
    v = seq[k]
    v = v[len(v) // 2:]                       # second half: steady state
    n = len(v) // per * per
    v = v[len(v) - n:]
    print(k, 'per position in the step (us):', ['%.1f' % (sum(v[i::per]) / len(v[i::per])) for i in range(per)], 'launches', n)
