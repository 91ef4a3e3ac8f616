"""GMRES iterations per step against the number of ranks (all ranks on ONE GPU, host-staged
transport): how much the rank-local potential multigrid (block Jacobi over ranks) costs."""
import os, socket, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
N_GLOBAL = int(sys.argv[1]) if len(sys.argv) > 1 else 192
LOCAL_COARSENINGS = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import torch.distributed as dist
    from fedm_amd.cases import streamer, streamer_distributed
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run = streamer_distributed.Runner(None, rank, world, 0, grading=4.0, transport="torch",
                                          n_per_gpu=int(round(N_GLOBAL / np.sqrt(world))))
        run.initialise()
        run.step()
        l0, n0 = run.linear_iterations, run.newton_iterations
        for _ in range(5):
            run.step()
        if rank == 0:
            q.put((world, run.global_n, run.global_n, (run.linear_iterations - l0) / 5, (run.newton_iterations - n0) / 5))
    except Exception as exc:   # the parent waits on the queue: tell it instead of leaving it to its timeout
        if rank == 0:
            q.put(exc)
        raise
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    for world in [int(w) for w in (sys.argv[3].split(",") if len(sys.argv) > 3 else "1,2,4".split(","))]:
        q = ctx.Queue()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
        procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs: p.start()
        res = q.get(timeout=300)
        if isinstance(res, Exception):
            raise res
        print("ranks %d: global mesh %dx%d, GMRES/step %.1f, Newton/step %.1f" % res, flush=True)
        for p in procs: p.join(timeout=60)
