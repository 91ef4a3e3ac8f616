"""Step time of the several-GPU solver on ONE rank over RCCL (distributed finest multigrid level, replicated
coarse levels, every all-reduce and the split Krylov graphs; no neighbour) against the plain single-GPU
solver on the same 576x576 mesh: what the several-GPU code path costs before any message travels."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"))
import torch
import torch.distributed as dist
from fedm_amd.cases import streamer, streamer_distributed

n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=0, world_size=1)


def timed(run, steps=20, warm=5):
    run.initialise()
    for _ in range(warm):
        run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k0 = run.prob.last_report.linear_iterations
    its = 0
    for _ in range(steps):
        run.step()
        its += run.linear_iterations_last_step if hasattr(run, "linear_iterations_last_step") else 0
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


rc = streamer_distributed.Runner(None, 0, 1, 0, grading=4.0, transport="rccl", n_per_gpu=n, distributed_multigrid=True)
ms_rccl = timed(rc)
stats = rc.prob.comm_stats()
msh = streamer.mesh(n, 4.0)
plain = streamer.Stepper(streamer.device_problem(msh.coords, msh.cells))
ms_plain = timed(plain)
print(f"{n}x{n}: several-GPU solver on one rank over RCCL {ms_rccl:.3f} ms/step ({stats['allreduces'] / 25:.1f} all-reduces, "
      f"{stats['halo_exchanges'] / 25:.1f} exchange calls per step), plain single-GPU solver {ms_plain:.3f} ms/step")
dist.destroy_process_group()
