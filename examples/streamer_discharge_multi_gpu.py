#!/usr/bin/env python3
"""The streamer case of examples/streamer_discharge.py over several MI355X of one node: one process per GPU,
the mesh partitioned among them, ghost values over RCCL (xGMI).  The counterpart of the reference's
`mpirun -np 8 python3 fedm-streamer.py` (README.md:63-67), where DOLFIN partitions the mesh and PETSc scatters
the ghosts.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/streamer_discharge_multi_gpu.py --mesh-spacing 4e-6 --end 1.4e-8 --out streamer_output

Every rank generates the same locally refined unstructured mesh (deterministic; or reads `--mesh file.xml`),
computes the same partition (bisection that counts the mesh edges a cut severs), keeps its part with eight
ghost layers, and the library does the rest: a Krylov step exchanges one vector and reduces twice
(DESIGN.md section 7).  Rank 0 writes `relative error.log` and, at the end, the fields as PVD/VTU.
`--share-one-gpu` runs all ranks on GPU 0 over a host-staged transport (a rehearsal: RCCL refuses two ranks
on one device).
"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--mesh-spacing", type=float, default=2.5e-5, help="finest spacing of the generated mesh [m]")
    ap.add_argument("--mesh", help="DOLFIN XML mesh to load instead of generating one")
    ap.add_argument("--end", type=float, default=1e-10, help="end time [s] (the reference script: 1.4e-8)")
    ap.add_argument("--out", default="streamer_output")
    ap.add_argument("--share-one-gpu", action="store_true")
    a = ap.parse_args()

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if a.share_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import __graft_entry__ as entry
    entry.build()                                  # before anything touches the GPU: the compiler driver exec()s its tools
    import torch
    import torch.distributed as dist
    from fedm_amd import mesh_io
    from fedm_amd.cases import streamer, streamer_distributed

    torch.cuda.set_device(local_rank)
    if a.share_one_gpu:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    try:
        mesh = mesh_io.read_dolfin_xml(a.mesh) if a.mesh else streamer.refined_mesh(a.mesh_spacing)
        out = Path(a.out)
        if rank == 0:
            out.mkdir(parents=True, exist_ok=True)
        dist.barrier()
        run = streamer_distributed.Runner(None, rank, world, local_rank, mesh=mesh,
                                          transport="torch" if a.share_one_gpu else "rccl",
                                          error_file=(out / "relative error.log") if rank == 0 else None)
        if rank == 0:
            print(run.partition_name, flush=True)
        run.initialise()
        while run.t < a.end * (1.0 - 1e-6):
            run.step()
            if rank == 0 and run.steps % 100 == 0:
                print(f"step {run.steps}: t = {run.t:.4e} s, dt = {run.dt.time_step:.3e} s", flush=True)
        # the owned rows of every rank, put together on rank 0 in the mesh's own vertex numbering
        owned = run.lm.vertex_global[:run.lm.n_owned]
        pieces = [None] * world
        dist.all_gather_object(pieces, (owned, run.prob.get_state()[:run.lm.n_owned]))
        if rank == 0:
            U = np.zeros((mesh.num_vertices(), 3))
            for ids, values in pieces:
                U[ids] = values
            for k, name in enumerate(("Ions", "electrons", "Phi")):
                mesh_io.PVDFile(out / name / f"{name}.pvd", mesh).write(U[:, k], name, run.t)
            np.save(out / "state.npy", U)           # ln n_i, ln n_e, Phi per mesh vertex
            stats = run.prob.comm_stats()
            print(f"{run.steps} steps to t = {run.t:.4e} s; {run.newton_iterations} Newton and {run.linear_iterations} "
                  f"GMRES iterations; {stats['halo_exchanges']} halo exchanges, {stats['allreduces']} all-reduces "
                  f"({stats['transport']}); ln n_e in [{U[:, 1].min():.2f}, {U[:, 1].max():.2f}]", flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
