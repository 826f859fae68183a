#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv): the kernels between two
consecutive adam_kernel launches near the end of the run, with start offset, gap to the previous kernel and duration,
then totals per kernel family.    python tools/step_timeline.py <kernel_trace.csv> [steps-from-the-end=12] [--brief]"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 12
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
a, b = idx[-back - 1] + 1, idx[-back] + 1
step = rows[a:b]
t0, prev = step[0][0], None
fam = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in step:
    short = re.sub(r'\(anonymous namespace\)::|void |at::native::', '', n)
    short = re.sub(r'\(.*', '', short)[:78]
    gap = (s - prev) / 1e3 if prev else 0.0
    if '--brief' not in sys.argv:
        print('%8.1f gap %6.1f dur %7.1f %s' % ((s - t0) / 1e3, gap, (e - s) / 1e3, short))
    key = re.sub(r'<.*', '', short)
    if key == 'gemm_kernel':
        key = 'gemm_kernel<' + ('double' if 'gemm_kernel<double' in short else 'float') + (', MIX' if short.rstrip('>').endswith(', 1') and 'double, 128' in short else '') + '>'
    fam[key][0] += 1
    fam[key][1] += (e - s) / 1e3
    prev = e
print('kernels %d  span %.1f us  busy %.1f us' % (len(step), (step[-1][1] - t0) / 1e3, sum(e - s for s, e, _ in step) / 1e3))
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print('%4d %9.1f us  %s' % (c, t, k))
