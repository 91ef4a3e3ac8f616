set -u
for N in 2 4; do
  for D in 8 1; do
    FEDM_HALO_DEPTH=$D timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500+N*10+D)) bench.py --gpus $N --steps 5 --warmup 2 --rehearse-on-one-gpu --late-start 0 --no-cpu-baseline > gpurun_out/rehearsal_${N}ranks_depth${D}.json 2> gpurun_out/rehearsal_${N}ranks_depth${D}.err
    rc=$?
    echo "N=$N depth=$D rc=$rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
  done
done
python - <<PY
import json
for N in (2,4):
    for D in (8,1):
        try:
            d=json.loads(open(f"gpurun_out/rehearsal_{N}ranks_depth{D}.json").read().strip().splitlines()[-1])
            m=d["multi_gpu"]
            print(N, D, "ms/step", round(d["ms_per_step"],2), "gmres", d["gmres_iterations_per_step"], "newton", d["newton_iterations_per_step"], "halo/step", round(m["halo_exchanges_per_step"],1), "allreduce/step", round(m["allreduces_per_step"],1), d["config"]["partition"][:120])
        except Exception as e:
            print(N, D, "failed", e)
PY
