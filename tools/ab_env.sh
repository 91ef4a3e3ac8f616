# A/B of an environment switch on the bench's headline window (no profiler): usage (GPU box)
#   bash tools/ab_env.sh NAME "VAR=a" "VAR=b" ...      -> gpurun_out/ab_NAME.txt
NAME=$1; shift
ONLY="--no-cpu-baseline --late-start 0 --second-mesh off --big-mesh 0 --no-glow-discharge"
python3 -c 'import __graft_entry__ as g; g.build()' || exit 1
: > gpurun_out/ab_$NAME.txt
for round in $(seq 1 ${ROUNDS:-2}); do
  for setting in "$@"; do
    env $setting timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 $ONLY > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err
    rc=$?
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$setting: killed" >> gpurun_out/ab_$NAME.txt; exit 1; fi
    python3 - "$setting" <<'PY' >> gpurun_out/ab_$NAME.txt
import json, sys
try:
    d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
    print(f"{sys.argv[1]:40s} {d['timesteps_per_sec']:8.1f} steps/s  {d['ms_per_step']:.3f} ms/step  gmres/step {d['gmres_iterations_per_step']}  newton/step {d['newton_iterations_per_step']}")
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/ab_tmp.err").read()[-600:])
PY
  done
done
cat gpurun_out/ab_$NAME.txt
