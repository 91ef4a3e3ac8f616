"""Kernel-by-kernel listing of ONE time step from a rocprofv3 kernel trace (steps end at
field_error_kernel): start offset, duration, gap to the previous kernel, name."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
ends = [i for i, k in enumerate(ks) if 'field_error_kernel' in k[2]]
a, b = ends[which - 1] if which > 0 else ends[which - 1], ends[which]
seg = ks[a + 1:b + 1]
short = lambda n: n.split('(')[0].replace('void fedm::', '').replace('fedm::', '')[-48:]
t0, prev = seg[0][0], ks[a][1]
for s, e, n in seg:
    print(f'{(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.2f}  gap {(s - prev) / 1e3:7.2f}  {short(n)}')
    prev = e
print(f'{len(seg)} kernels, span {(seg[-1][1] - t0) / 1e3:.1f} us, busy {sum(e - s for s, e, _ in seg) / 1e3:.1f} us')
