# One traced bench run -> gpurun_out/step_sequence.txt (kernel by kernel, a step of the timed region) and the
# per-step summary.  usage (GPU box): bash tools/trace_step.sh [STEP=6]
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
STEP=${1:-6}
OUT=gpurun_out/seq_prof
rm -rf "$OUT"; mkdir -p "$OUT"
python3 -c 'import __graft_entry__ as g; g.build()' || exit 1
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o q -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --late-start 0 --second-mesh off --big-mesh 0 --no-glow-discharge > "$OUT/bench.json" 2> "$OUT/err.txt" || exit 2
TRACE=$(find "$OUT" -name "q_kernel_trace.csv" | head -1)
python3 tools/step_sequence.py "$TRACE" "$STEP" > gpurun_out/step_sequence.txt
python3 tools/step_sequence.py "$TRACE" all > gpurun_out/step_summary.txt
rm -f "$TRACE"
tail -1 gpurun_out/step_sequence.txt; cat gpurun_out/step_summary.txt
