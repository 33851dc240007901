#!/usr/bin/env python3
"""Prints one step's kernel sequence (name, duration, grid) from a rocprofv3 kernel trace CSV: the last N launches."""
import csv, sys, glob
path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows[-n:]:
    print('%8.1f us  grid %-8s wg %-5s %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Grid_Size_X', '?'), r.get('Workgroup_Size_X', '?'), r['Kernel_Name'][:110]))
