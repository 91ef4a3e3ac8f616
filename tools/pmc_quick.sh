#!/bin/bash
# SQ counters of chosen kernels in a short bench run (GPU box): bash tools/pmc_quick.sh [bench flags --] PATTERN...
#   e.g. bash tools/pmc_quick.sh --family tensor -- assemble_lean3 residual_lean3     -> gpurun_out/pmcq/summary.txt
set -u
EXTRA=""
case " $* " in *" -- "*) while [ "$1" != "--" ]; do EXTRA="$EXTRA $1"; shift; done; shift;; esac
OUT=gpurun_out/pmcq
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -c 'import __graft_entry__ as g; g.build()' || exit 9
ARGS="bench.py --steps 2 --warmup 0 --repeats 1 --preroll 0 --no-cpu-baseline --late-start 0 --second-mesh off --big-mesh 0 --no-glow-discharge $EXTRA"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
    --output-format csv -d "$OUT/sq1" -o s -- python3 $ARGS > /dev/null 2> "$OUT/sq1.err" || exit 4
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --output-format csv -d "$OUT/sq2" -o s -- python3 $ARGS > /dev/null 2> "$OUT/sq2.err" || exit 5
: > "$OUT/summary.txt"
for pat in "$@"; do
  { echo "== $pat ($ARGS)"; python3 tools/pmc_kernel.py "$pat" $(find "$OUT" -name "*counter_collection.csv"); } >> "$OUT/summary.txt"
done
find "$OUT" -name "*counter_collection.csv" -delete
cat "$OUT/summary.txt"
