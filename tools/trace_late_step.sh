# Kernel-by-kernel listing of a time step of the DEVELOPED streamer (step 206 of the bench case, 1 ns: hard-regime
# preconditioner set, ~9 Krylov steps per Newton system) -> gpurun_out/late_step_sequence.txt.  usage (GPU box):
#   bash tools/trace_late_step.sh [STEP=206]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
STEP=${1:-206}
OUT=gpurun_out/late_prof
rm -rf "$OUT"; mkdir -p "$OUT"
python3 -c 'import __graft_entry__ as g; g.build()' || exit 1
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o q -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --late-start 200 --second-mesh off --big-mesh 0 --no-glow-discharge > "$OUT/bench.json" 2> "$OUT/err.txt" || exit 2
TRACE=$(find "$OUT" -name "q_kernel_trace.csv" | head -1)
python3 tools/step_sequence.py "$TRACE" "$STEP" > gpurun_out/late_step_sequence.txt
rm -f "$TRACE"
tail -1 gpurun_out/late_step_sequence.txt
