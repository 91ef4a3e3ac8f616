# bench.py with N ranks sharing ONE GPU (host-staged transport; at most 6 processes may use the card): the several-GPU
# code path end to end -- partition, deep halos, distributed multigrid, collectives -- with the counts a real run would
# show (GMRES iterations, halo exchanges, all-reduces, bytes).  usage (GPU box): bash tools/rehearse_multi_rank.sh [family] [ranks...]
set -u
FAMILY=${1:-unstructured}; shift || true
RANKS=${*:-"2 4"}
python3 -c 'import __graft_entry__ as g; g.build()' || exit 9
for N in $RANKS; do
    OUT=gpurun_out/rehearsal_${FAMILY}_${N}ranks
    timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N*10)) \
        bench.py --gpus $N --steps 5 --warmup 2 --repeats 1 --preroll 0 --family $FAMILY --rehearse-on-one-gpu --late-start 0 --no-cpu-baseline --configs4 ${CONFIGS4:-auto} > $OUT.json 2> $OUT.err
    rc=$?
    echo "family=$FAMILY N=$N rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
python3 - $FAMILY $RANKS <<'PY'
import json, sys
fam = sys.argv[1]
for N in sys.argv[2:]:
    try:
        d = json.loads(open(f"gpurun_out/rehearsal_{fam}_{N}ranks.json").read().strip().splitlines()[-1])
        m = d["multi_gpu"]
        print(N, "ranks:", "ms/step", round(d["ms_per_step"], 2), "gmres", d["gmres_iterations_per_step"], "newton", d["newton_iterations_per_step"],
              "halo/step", round(m["halo_exchanges_per_step"], 1), "allreduce/step", round(m["allreduces_per_step"], 1),
              "halo KB/step", round((m["halo_bytes_per_step"] or 0) / 1e3, 1), "allreduce KB/step", round((m["allreduce_bytes_per_step"] or 0) / 1e3, 1),
              d["config"]["partition"][:110], "| assembly", d["roofline"]["kernel"][:40], "%.1f us" % (1e3 * d["roofline"]["ms_per_launch"]),
              "| tiles", (d["config"].get("fieldsplit_tiles") or {}).get("max_vertices"))
        if "configs4" in d:
            c = d["configs4"]
            print("   configs4:", "ms/step", round(c["ms_per_step"], 2), "gmres", c["gmres_iterations_per_step"], "allreduce KB/step", round(c["allreduce_bytes_per_step"] / 1e3, 1))
    except Exception as e:
        print(N, "failed", e)
PY
