#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   profiles/<tag>_kernel_stats.csv           rocprofv3 --kernel-trace --stats of the default bench command
#   profiles/<tag>_bench_under_rocprof.json   the bench line of that same run
#   profiles/<tag>_pmc_traffic.json           FETCH_SIZE / WRITE_SIZE passes (separate runs) per workload -- the headline's
#                                             refined unstructured mesh, the 576x576 tensor-product mesh, the glow discharge --
#                                             with the gfx950 corrections
#   profiles/<tag>_pmc_assembly_sq.txt        SQ counters of the F + J and residual-only assembly kernels on both meshes
#   profiles/<tag>_step_sequence.txt          one time step of the headline, kernel by kernel
# Everything is written under gpurun_out/<tag>_prof/ (merged back by gpurun); copy_profiles.py then copies the summaries
# into profiles/ -- on the box, and once more at home: python tools/copy_profiles.py <tag> gpurun_out/<tag>_prof
# The program follows `--` directly (no env/bash hop).
set -u
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_prof
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# The library is built HERE, unprofiled: under rocprofv3 every child process inherits the profiler's
# preloaded tool, which initialises the GPU, and the compiler driver's exec hops would then happen in
# GPU-initialised processes (bench.py refuses to compile under a profiler for the same reason).
python3 -c 'import __graft_entry__ as g; g.build()' || exit 9
ONLY="--no-cpu-baseline --late-start 0 --second-mesh off --big-mesh 0 --no-glow-discharge"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err" || exit 1
rm -f "$OUT"/trace/t_kernel_trace.csv "$OUT"/trace/*/t_kernel_trace.csv      # tens of MB; the stats are what is kept
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
SQ2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
for W in unstructured tensor; do
  SHORT="bench.py --steps 2 --warmup 0 --repeats 1 --preroll 0 --family $W $ONLY"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$W/fetch" -o f -- python3 $SHORT > /dev/null 2> "$OUT/$W.fetch.err" || exit 2
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$W/write" -o w -- python3 $SHORT > /dev/null 2> "$OUT/$W.write.err" || exit 3
  rocprofv3 --pmc $SQ1 --output-format csv -d "$OUT/$W/sq1" -o s -- python3 $SHORT > /dev/null 2> "$OUT/$W.sq1.err" || exit 4
  rocprofv3 --pmc $SQ2 --output-format csv -d "$OUT/$W/sq2" -o s -- python3 $SHORT > /dev/null 2> "$OUT/$W.sq2.err" || exit 5
done
# glow discharge (configs[2]): traffic of its assembly kernels
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/glow_discharge/fetch" -o f -- python3 tools/gd_steps.py 4 > /dev/null 2> "$OUT/gd.fetch.err" || exit 6
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/glow_discharge/write" -o w -- python3 tools/gd_steps.py 4 > /dev/null 2> "$OUT/gd.write.err" || exit 7
find "$OUT" -name "*counter_collection.csv" | head -20
python3 tools/copy_profiles.py "$TAG" "$OUT"
find "$OUT" -name "*counter_collection.csv" -size +4M -delete         # (gpurun_out/ travels back: 64 MiB)
# kernel-by-kernel listing of one time step (step 6 of the run): a short traced run of its own
SEQ="bench.py --steps 6 --warmup 3 --repeats 1 --preroll 0 $ONLY"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/seq" -o q -- python3 $SEQ > /dev/null 2> "$OUT/seq.err" || exit 8
TRACE=$(find "$OUT/seq" -name "q_kernel_trace.csv" | head -1)
{ echo "One accepted time step of the headline case (refined unstructured mesh, step 6 of the run, Krylov steps replayed as graphs), kernel by kernel:";
  echo "start offset [us], duration [us], gap to the previous kernel [us], kernel (rocprofv3 --kernel-trace of python3 $SEQ; tools/step_sequence.py)";
  echo; python3 tools/step_sequence.py "$TRACE" 6; echo; echo "every step of that process (the last ones are bench.py's profiling pass: plain launches, event pairs):"; python3 tools/step_sequence.py "$TRACE" all; } > "profiles/${TAG}_step_sequence.txt"
cp "profiles/${TAG}_step_sequence.txt" "$OUT/step_sequence.txt"     # (only gpurun_out/ travels back from the GPU box)
rm -f "$TRACE"
