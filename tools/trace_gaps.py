"""Reads a rocprofv3 --kernel-trace CSV of bench.py and reports, for the timed steps, how busy the GPU was: union of the
kernel intervals, idle gaps between them, the largest gaps and what ran on either side."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70], r.get('Stream_Id', r.get('Queue_Id', ''))) for r in rows))
# the steps: from one perm_data_kernel to the next
marks = [i for i, e in enumerate(ev) if 'perm_data_kernel' in e[2]]
print('kernels', len(ev), 'steps', len(marks))
lo, hi = marks[len(marks) // 2], marks[len(marks) // 2 + 10]
seg = ev[lo:hi]
t0, t1 = seg[0][0], ev[hi][0]
busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]; gaps = []
for s, e, n, q in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, prev_n, n)); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev_n = n if e >= cur_e else prev_n if 'prev_n' in dir() else n
busy += cur_e - cur_s
print('10 steps: %.3f ms per step, busy %.3f ms per step, idle %.3f ms per step in %d gaps' % ((t1 - t0) / 1e7, busy / 1e7, (t1 - t0 - busy) / 1e7, len(gaps)))
gaps.sort(reverse=True)
for g, a, b in gaps[:25]:
    print('  gap %7.1f us   after %-50s before %s' % (g / 1e3, a[:50], b[:50]))
import collections
h = collections.Counter(min(int(g / 1000), 20) for g, _, _ in gaps)
print('gap histogram (us: count):', sorted(h.items()))
