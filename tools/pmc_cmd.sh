#!/bin/bash
# SQ counters of chosen kernels in any python command (GPU box): bash tools/pmc_cmd.sh "script.py args" PATTERN...
#   e.g. FEDM_GD_HAND=5 bash tools/pmc_cmd.sh "tools/gd_steps.py 3" gd_jacobian_rows   -> gpurun_out/pmcc/summary.txt
set -u
CMD=$1; shift
OUT=gpurun_out/pmcc
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 -c 'import __graft_entry__ as g; g.build()' || exit 9
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
    --output-format csv -d "$OUT/sq1" -o s -- python3 $CMD > /dev/null 2> "$OUT/sq1.err" || exit 4
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES \
    --output-format csv -d "$OUT/sq2" -o s -- python3 $CMD > /dev/null 2> "$OUT/sq2.err" || exit 5
: > "$OUT/summary.txt"
for pat in "$@"; do
  { echo "== $pat ($CMD, FEDM_GD_HAND=${FEDM_GD_HAND:-default})"; python3 tools/pmc_kernel.py "$pat" $(find "$OUT" -name "*counter_collection.csv"); } >> "$OUT/summary.txt"
done
find "$OUT" -name "*counter_collection.csv" -delete
cat "$OUT/summary.txt"
