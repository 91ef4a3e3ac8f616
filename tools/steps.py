"""Per-time-step GPU busy/idle from a rocprofv3 kernel trace: steps end at field_error_kernel."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
ends = [i for i, k in enumerate(ks) if 'field_error_kernel' in k[2]]
short = lambda n: n.split('(')[0].replace('void fedm::', '').replace('fedm::', '')[-40:]
for a, b in zip(ends, ends[1:]):
    seg = ks[a + 1:b + 1]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    names = collections.Counter(short(n) for _, _, n in seg)
    graph = 'graphs' if names.get('dots_scatter_kernel<2>', 0) else 'plain'
    print(f'{len(seg):5d} kernels span {span/1e6:6.3f} ms busy {busy/1e6:6.3f} ms idle {100*(1-busy/span):4.1f}% {graph} '
          f'gap to prev step {(seg[0][0]-ks[a][1])/1e3:6.1f} us')

if len(sys.argv) > 2:
    a, b = ends[int(sys.argv[2])], ends[int(sys.argv[2]) + 1]
    seg = ks[a + 1:b + 1]
    d = collections.defaultdict(lambda: [0, 0])
    for s, e, n in seg:
        d[short(n)][0] += e - s
        d[short(n)][1] += 1
    for n, (t, c) in sorted(d.items(), key=lambda kv: -kv[1][0]):
        print(f'{t/1e3:8.1f} us  n={c:4d}  avg {t/c/1e3:6.2f}  {n}')
