"""Inter-kernel gap analysis of a rocprofv3 kernel trace (CSV)."""
import collections
import csv
import sys

import numpy as np

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
ks = ks[len(ks) // 3:]
busy = sum(e - s for s, e, _ in ks)
span = ks[-1][1] - ks[0][0]
print('kernels', len(ks), 'busy ms', busy / 1e6, 'span ms', span / 1e6, 'idle frac', 1 - busy / span)
short = lambda n: n.split('(')[0].replace('void fedm::', '').replace('fedm::', '')[-44:]
gaps = collections.defaultdict(list)
dur = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    gaps[(short(n0), short(n1))].append(s1 - e0)
for s, e, n in ks:
    dur[short(n)].append(e - s)
print('--- kernel time')
for t, n, k in sorted(((sum(v), len(v), k) for k, v in dur.items()), reverse=True)[:22]:
    print(f'{t / 1e6:8.3f} ms {100 * t / span:5.1f}%  n={n:5d} avg={t / n / 1e3:7.2f} us  {k}')
print('--- gaps')
for t, n, k in sorted(((sum(v), len(v), k) for k, v in gaps.items()), reverse=True)[:18]:
    print(f'{t / 1e6:8.3f} ms {100 * t / span:5.1f}%  n={n:5d} avg={t / n / 1e3:7.2f} us  {k[0]} -> {k[1]}')
