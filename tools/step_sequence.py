"""Kernel-by-kernel listing of ONE time step from a rocprofv3 kernel trace (steps end at
field_error_kernel): start offset, duration, gap to the previous kernel, name.

usage: python tools/step_sequence.py TRACE.csv [STEP]     STEP >= 1: the STEP-th step of the process (1-based, warm-up
                                                          steps included); negative: counted from the end
       python tools/step_sequence.py TRACE.csv all        one line per step: kernels, span, busy time
(bench.py's profiling pass, at the end of its run, launches plainly with an event pair around the timed kernels --
 its steps have more kernels and gaps; the timed region replays graphs)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
ends = [i for i, k in enumerate(ks) if 'field_error_kernel' in k[2]]
short = lambda n: n.split('(')[0].replace('void fedm::', '').replace('fedm::', '')[-48:]
if len(sys.argv) > 2 and sys.argv[2] == "all":
    for n in range(1, len(ends)):
        seg = ks[ends[n - 1] + 1:ends[n] + 1]
        print(f'step {n + 1:3d}: {len(seg):4d} kernels, span {(seg[-1][1] - seg[0][0]) / 1e3:8.1f} us, busy '
              f'{sum(e - s for s, e, _ in seg) / 1e3:8.1f} us, scatter kernels '
              f'{sum("fs_scatter" in k[2] for k in seg)}')
    sys.exit(0)
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
a, b = (ends[which - 2], ends[which - 1]) if which > 0 else (ends[which - 1], ends[which])
seg = ks[a + 1:b + 1]
t0, prev = seg[0][0], ks[a][1]
for s, e, n in seg:
    print(f'{(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.2f}  gap {(s - prev) / 1e3:7.2f}  {short(n)}')
    prev = e
print(f'{len(seg)} kernels, span {(seg[-1][1] - t0) / 1e3:.1f} us, busy {sum(e - s for s, e, _ in seg) / 1e3:.1f} us')
