"""Streamer case across several GPUs: one process per GPU, vertex partition by RCB,
ghost exchange + all-reduces inside libfedm_hip.so (RCCL over xGMI).

Weak scaling: the mesh handed in is the per-GPU mesh size; the global mesh has
``world`` times as many cells (the box is fixed, the resolution grows).
"""
import os
import sys

import numpy as np

from .. import partition
from ..device import DeviceProblem, rccl_unique_id
from ..mesh import Marking_boundaries, Mesh, RectangleMesh, geometric_lines
from . import streamer


def global_mesh(n_per_gpu, world, grading):
    """About world * n^2 * 2 cells: refine both directions by sqrt(world)."""
    n = int(round(n_per_gpu * np.sqrt(world)))
    return streamer.mesh(n, grading), n


class Runner(streamer.Stepper):
    fallback_reason = None

    def __init__(self, per_gpu_mesh, rank, world, local_rank, grading=4.0, transport="rccl",
                 group=None, n_per_gpu=None, global_n=None, distributed_multigrid=None, mesh=None,
                 halo_depth=None, **kw):
        """Mesh: ``mesh`` (any triangle mesh of the box, e.g. `streamer.refined_mesh`) is the GLOBAL
        mesh, partitioned as it is; else ``global_n`` cells per side of the whole tensor-product mesh
        (strong scaling, e.g. BASELINE configs[4]), else ``n_per_gpu`` (or the size of
        ``per_gpu_mesh``) cells per side PER GPU."""
        import torch.distributed as dist
        if mesh is not None:
            gmesh, n = mesh, 0
        elif global_n:
            gmesh, n = streamer.mesh(int(global_n), grading), int(global_n)
        else:
            n_per_gpu = n_per_gpu or int(round(np.sqrt(per_gpu_mesh.num_cells() / 2)))
            gmesh, n = global_mesh(n_per_gpu, world, grading)
        # Ghost layers: 8 = the first stage, the five further sweeps of the degree-6 species polynomial, the
        # coupling product, two multigrid smoothings and the Krylov product -- one exchange per Krylov step
        # instead of ten (fedm_amd/partition.py); 1: the one-layer halo with an exchange per operator.
        # (also with ONE rank when it runs the several-GPU solver: how the tests and
        # tools/one_rank_rccl_overhead.py reach that code path)
        if halo_depth is None:
            several = world > 1 or bool(distributed_multigrid)
            halo_depth = int(os.environ.get("FEDM_HALO_DEPTH", "8")) if several else 1
        self.halo_depth = halo_depth
        # FEDM_PARTITIONER=graph: multilevel recursive bisection of the vertex graph (fedm_amd/graph_partition.py), never
        # worse in edge cut than the default, the coordinate bisection along the axis that severs fewer mesh edges --
        # on the two bench meshes (a box, graded or locally refined) the two give the same cuts
        self.partitioner = os.environ.get("FEDM_PARTITIONER", "rcb")
        if self.partitioner == "graph":
            from .. import graph_partition
            part = graph_partition.partition_graph(gmesh.coords, gmesh.cells, world)
        elif self.partitioner == "rcb":
            part = partition.partition_rcb(gmesh.coords, world, gmesh.cells)
        else:
            raise ValueError(f"FEDM_PARTITIONER={self.partitioner}: 'rcb' or 'graph'")
        lm = partition.local_mesh(gmesh.coords, gmesh.cells, part, rank, depth=halo_depth)
        self.lm, self.world, self.rank = lm, world, rank
        # (True with one rank: the several-GPU solver -- distributed finest level, replicated coarse
        # levels, their all-reduces -- on a single rank: how the tests reach that code with RCCL)
        self.distributed_multigrid = world > 1 if distributed_multigrid is None else bool(distributed_multigrid)
        self.global_n = n
        # boundary tags and Dirichlet rows from the global mesh, restricted to the local cells
        gtags = Marking_boundaries(gmesh, streamer.BOUNDARIES)
        tags = gtags[lm.cell_global]
        ddofs, dvals = streamer.dirichlet(lm.coords)
        prob = DeviceProblem(lm.coords, lm.cells, streamer.model(), facet_tags=tags,
                             dirichlet_dofs=ddofs, dirichlet_vals=dvals, device=local_rank,
                             n_owned=lm.n_owned, identity_vertices=lm.identity_vertices if halo_depth > 1 else None,
                             halo_depth=halo_depth)
        self.transport_requested = transport
        if transport == "rccl":
            # Every rank must take the same branch, and no rank may enter ncclCommInitRank alone
            # (the others would block in it for ever): agree BEFORE the call on what can be known
            # beforehand -- librccl loads, and no two ranks sit on one device (RCCL refuses that) --
            # and AFTER it on its outcome.
            reason = None
            try:
                import torch
                props = torch.cuda.get_device_properties(local_rank)
                # physical identity of the GPU: everything the runtime reports (a field that is missing or
                # the same on every GPU must not make distinct GPUs look like one); nothing at all
                # reported: the device index
                fields = tuple(str(getattr(props, a, "")) for a in
                               ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"))
                ident = (os.uname().nodename, fields if any(fields) else ("index", local_rank))
                uid = rccl_unique_id() if rank == 0 else None
                mine = (ident, None)
            except Exception as exc:                     # noqa: BLE001 - reported below, by every rank
                uid, mine = None, (None, f"{type(exc).__name__}: {exc}")
            seen = [None] * world
            dist.all_gather_object(seen, mine, group=group)
            errors = [f"rank {r}: {e}" for r, (_, e) in enumerate(seen) if e]
            idents = [i for i, _ in seen]
            if errors:
                reason = "; ".join(errors)
            elif len(set(idents)) < world:
                reason = "several ranks share one GPU"
            ok = 0
            if reason is None:
                box = [uid]
                dist.broadcast_object_list(box, src=0, group=group)
                try:
                    prob.init_comm_rccl(lm, box[0], rank, world)
                    ok = 1
                except Exception as exc:                 # noqa: BLE001 - reported, then agreed on
                    reason = f"rank {rank}: {exc}"
                flags = [None] * world
                dist.all_gather_object(flags, (ok, reason), group=group)
                bad = [r for f, r in flags if not f]
                if bad:
                    ok, reason = 0, "; ".join(str(b) for b in bad)
            if not ok:
                print(f"[fedm_amd] rank {rank}: RCCL transport unavailable ({reason}); "
                      f"falling back to host-staged exchanges -- NOT the xGMI data path", file=sys.stderr, flush=True)
                transport = "torch-gloo (RCCL set-up failed)"
                self.fallback_reason = reason
                prob.init_comm_torch(lm, dist.new_group(backend="gloo"))
        else:
            prob.init_comm_torch(lm, group)
        self._group = group if transport != "torch-gloo (RCCL set-up failed)" else None
        super().__init__(prob, **kw)
        self.world_size = world
        self.transport = transport
        self.total_dofs = gmesh.num_vertices() * 3
        shape = f"{n}x{n}" if n else f"unstructured, {gmesh.num_vertices()} vertices"
        kind = ("multilevel graph partition (heavy-edge matching, FM refinement)" if self.partitioner == "graph"
                else "RCB vertex partition (cuts severing the fewest edges)")
        self.partition_name = (f"{kind}, {world} parts, global mesh {shape}, "
                               f"{lm.n_owned} owned + {lm.n_ghost} ghost vertices in {halo_depth} layer(s) on rank {rank}, "
                               f"{len(lm.neighbours)} neighbours, transport {transport}")

    def initialise(self):
        """As Stepper.initialise, with the multigrid whose coarsest level spans all ranks."""
        from ..device import chebyshev_weights
        prob = self.prob
        U = np.zeros((prob.nv, 3))
        U[:, 0], U[:, 1] = streamer.initial_log_densities(prob.coords)
        prob.set_state(U, U, U)
        if self.distributed_multigrid:
            # (the distributed finest level keeps the V(1,1) cycle: no alternative for hard systems)
            prob.setup_multigrid_distributed(self.lm, self._group, nu=streamer.MULTIGRID["nu"],
                                             omega=streamer.MULTIGRID["omega"])
            # one set of sweeps: across ranks the switch to a second set can only follow the all-reduced Krylov
            # counts, and with V(1,1) the cheaper degree-4 set buys nothing there (tools/late_sweep.py: 6.8
            # against 6.9 ms per step on the tensor-product mesh, 7.8 against 6.7 on the refined one)
            prob.set_fieldsplit(chebyshev_weights(6))
        else:
            prob.setup_multigrid(**streamer.MULTIGRID)
            prob.set_fieldsplit(chebyshev_weights(6), hard_weights=chebyshev_weights(4))
        its = prob.poisson_solve(rtol=1e-12)
        U = prob.get_state()
        prob.set_state(U, U, U)
        return U, its
