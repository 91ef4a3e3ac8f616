# Kernel-by-kernel listing of glow-discharge time steps (BASELINE configs[2], 141 x 141 mesh) -> gpurun_out/gd_step_kernels.txt
# usage (GPU box): bash tools/trace_gd_step.sh [STEPS=12]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/gd_prof
rm -rf "$OUT"; mkdir -p "$OUT"
python3 -c 'import __graft_entry__ as g; g.build()' || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o q -- python3 tools/gd_steps.py ${1:-12} > "$OUT/out.txt" 2> "$OUT/err.txt" || exit 2
python3 - "$OUT" <<'PY' > gpurun_out/gd_step_kernels.txt
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/**/q_kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
t0 = rows[0][0]
# the second half of the trace: steady steps
half = rows[len(rows) // 2:]
span = (half[-1][1] - half[0][0]) / 1e3
busy = sum(e - s for s, e, _ in half) / 1e3
tot, cnt = collections.Counter(), collections.Counter()
for s, e, n in half:
    n = re.sub(r"\(.*", "", n)[:90]
    tot[n] += (e - s) / 1e3
    cnt[n] += 1
print(f"second half of the trace: {len(half)} kernels, span {span:.0f} us, busy {busy:.0f} us ({100 * busy / span:.0f} %)")
for n, v in tot.most_common(30):
    print(f"{v:10.1f} us {cnt[n]:6d} x {v / cnt[n]:7.2f}  {n}")
# the gaps: which kernel starts after how long an idle time
gaps = collections.Counter(); gcnt = collections.Counter()
for (s0, e0, n0), (s1, e1, n1) in zip(half, half[1:]):
    g = (s1 - e0) / 1e3
    if g > 3.0:
        key = re.sub(r"\(.*", "", n0)[:40] + "  ->  " + re.sub(r"\(.*", "", n1)[:40]
        gaps[key] += g; gcnt[key] += 1
print("idle times over 3 us, by the kernels on either side:")
for k, v in gaps.most_common(14):
    print(f"{v:10.1f} us {gcnt[k]:5d} x {v / gcnt[k]:7.1f}  {k}")
PY
rm -f $(find "$OUT" -name "q_kernel_trace.csv")
cat "$OUT/out.txt"; head -60 gpurun_out/gd_step_kernels.txt
