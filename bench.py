#!/usr/bin/env python3
"""Headline benchmark: BDF2 time steps of the streamer_discharge case on MI355X.

``python bench.py --gpus N --steps K --warmup W``.  A *step* is one accepted
adaptive BDF2 time step of examples/streamer_discharge/fedm-streamer.py:304-340
(shift states, Newton solve with F/J assembly and GMRES, error norm, step
controller) on the state resident in HBM.  Workload (BASELINE.json configs[3]):
2-D axisymmetric streamer, 576x576 "right" mesh graded towards the axis
(332 929 vertices x 3 equations = 998 787 DOFs).  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md, chip-level parameters (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mesh", type=int, default=576, help="cells per side (per GPU)")
    ap.add_argument("--grading", type=float, default=4.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-mesh", type=int, default=144)
    ap.add_argument("--cpu-steps", type=int, default=2)
    # rehearsal of the N > 1 path on a one-GPU box: all ranks on cuda:0, gloo process group (RCCL
    # refuses two ranks on one device, so the library falls back to its host-staged transport)
    ap.add_argument("--rehearse-on-one-gpu", action="store_true")
    return ap.parse_args()


def spmv_bytes(sz):
    """Algorithmic bytes of one SpMV in the sliced block-ELL layout:
    every structural block once (n_eq^2 values + one column index), x read once,
    y written once, one slice offset per 64 vertices.  (The Krylov product is the plain one:
    the field split sits on the right of the operator.)"""
    neq, nnzb, nv = sz["n_eq"], sz["nnz_blocks"], sz["n_vertices"]
    return nnzb * (neq * neq * 8 + 4) + nv * neq * 16 + (nv // 64 + 1) * 4


def assembly_bytes(sz):
    """coords + 3 state vectors + connectivity + cell slots + matrix values
    written once + residual."""
    neq, nnzb, nv, nc = sz["n_eq"], sz["nnz_blocks"], sz["n_vertices"], sz["n_cells"]
    return nv * (16 + 24 * neq) + nc * (12 + 36) + nnzb * neq * neq * 8 + nv * neq * 8


def pmc_traffic():
    """HBM bytes per launch from the committed PMC passes of this same workload
    (profiles/r01_pmc_traffic.json: 2*FETCH_SIZE + WRITE_SIZE, gfx950 correction applied);
    counters cannot be read from inside the run, so None when the file is absent."""
    f = ROOT / "profiles" / "r01_pmc_traffic.json"
    if not f.exists():
        return {}
    k = json.loads(f.read_text())["kernels"]
    out = {}
    for name, v in k.items():
        if "traffic_bytes_corrected" in v:
            out[name] = v["traffic_bytes_corrected"]
    # the Krylov SpMV is the plain instantiation (<n_eq, false>: the field split sits on the right)
    spmv = [n for n in out if "spmv_kernel" in n and "ell_" not in n]
    plain = [n for n in spmv if "false>" in n]
    if spmv:
        out["spmv"] = out[(plain or spmv)[0]]
    return out


def measured_copy_ceiling(device):
    """Device copy bandwidth on this box (read + write bytes / time, 1 GiB buffers, HIP events):
    the practical ceiling next to the 8 TB/s specification."""
    import torch
    n = 1 << 27                                           # 1 GiB of fp64
    a = torch.empty(n, dtype=torch.float64, device=device)
    b = torch.empty_like(a)
    a.fill_(1.0)
    b.copy_(a)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / reps
    del a, b
    torch.cuda.empty_cache()
    return 2 * n * 8 / (ms * 1e-3) / 1e9


def cpu_baseline(n, steps):
    """The oracle (numpy assembly + SuperLU, 'CPU restatement, not FEniCS') on a bounded
    sample of the same workload, timed on this box's host cores."""
    from oracle import streamer as ost
    from oracle.mesh import graded_axis, rectangle_right
    import warnings
    mesh = rectangle_right(0.0, 0.0, ost.BOX, ost.BOX, n, n, xs=graded_axis(ost.BOX, n, 4.0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = ost.build(mesh)
        ost.initial_state(model)          # untimed warm-up of the code path
        t0 = time.perf_counter()
        _, st, _, _ = ost.run(mesh=mesh, max_steps=steps)
        el = time.perf_counter() - t0
    ndof = mesh.nv * 3
    return {"value": ndof * steps / el, "unit": "DOF-updates/s", "cores": 1, "kind": "port",
            "timesteps_per_sec": steps / el,
            "sample": f"{steps} accepted BDF2 steps (incl. initial Poisson solve) of the same "
                      f"streamer case on a {n}x{n} graded mesh ({ndof} DOFs); oracle = numpy "
                      f"assembly + SuperLU direct solve, single thread; host has "
                      f"{os.cpu_count()} cores"}


def pin_to_gpu_numa_node(torch, local_rank):
    """Run this process on the CPUs next to its GPU (what `numactl --cpunodebind` would do): the
    solver's host side polls a mailbox in pinned host memory and launches a graph per Krylov step,
    so a remote NUMA node costs microseconds at every one of a dozen waits per time step.  Best
    effort: any failure leaves the placement as it was."""
    if os.environ.get("FEDM_BENCH_PIN", "1") == "0":
        return "unchanged (FEDM_BENCH_PIN=0)"
    try:
        bus = torch.cuda.get_device_properties(local_rank).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(local_rank), "pci_domain_id", 0)
        dev = getattr(torch.cuda.get_device_properties(local_rank), "pci_device_id", 0)
        path = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{dev:02x}.0/local_cpulist"
        cpus = set()
        for part in open(path).read().strip().split(","):
            if part:
                lo, _, hi = part.partition("-")
                cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = os.sched_getaffinity(0)
        target = cpus & allowed
        if target and target != allowed:
            os.sched_setaffinity(0, target)
            return f"cpus of the GPU's NUMA node ({len(target)} of {len(allowed)} allowed)"
        return "unchanged (all allowed CPUs are local, or none is)"
    except Exception as exc:   # noqa: BLE001 - placement is an optimisation only
        return f"unchanged ({type(exc).__name__})"


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    if not entry.LIB.exists():
        entry.build()
    from fedm_amd.cases import streamer
    from fedm_amd import functions as ff

    placement = pin_to_gpu_numa_node(torch, local_rank)

    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    n = args.mesh
    msh = streamer.mesh(n, args.grading)
    if distributed:
        from fedm_amd.cases import streamer_distributed
        runner = streamer_distributed.Runner(msh, rank, world, local_rank, args.grading)
    else:
        prob = streamer.device_problem(msh.coords, msh.cells, device=local_rank)
        runner = streamer.Stepper(prob)
    runner.initialise()
    for _ in range(args.warmup):
        runner.step()

    runner.profile(1)             # HIP events around the assembly kernel, on the library's stream
    barrier()
    t0 = time.perf_counter()
    n0 = (runner.newton_iterations, runner.linear_iterations)
    for _ in range(args.steps):
        runner.step()
    barrier()
    elapsed = time.perf_counter() - t0
    n1 = (runner.newton_iterations, runner.linear_iterations)
    prof = runner.profile_read()
    # Second, untimed pass for the kernels inside the Krylov iterations: timing them needs plain
    # launches (HIP events cannot sit inside the replayed per-iteration graphs).
    pass_steps = max(1, min(args.steps, 5))
    runner.profile(2)
    barrier()
    t1 = time.perf_counter()
    for _ in range(pass_steps):
        runner.step()
    barrier()
    elapsed2 = time.perf_counter() - t1
    prof2 = runner.profile_read()
    runner.profile(False)
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    total_dofs = runner.total_dofs
    sz = runner.sizes()
    # average launch duration of the hot kernels inside the timed region
    ms_asm = prof["assembly_FJ"][0] / max(prof["assembly_FJ"][1], 1)
    ms_spmv = prof2["spmv"][0] / max(prof2["spmv"][1], 1)
    ms_res = prof["assembly_F"][0] / max(prof["assembly_F"][1], 1)
    b_spmv, b_asm = spmv_bytes(sz), assembly_bytes(sz)
    gbs_spmv = b_spmv / (ms_spmv * 1e-3) / 1e9
    gbs_asm = b_asm / (ms_asm * 1e-3) / 1e9
    share = {k: v[0] / (elapsed * 1e3) for k, v in prof.items()}
    share2 = {k: v[0] / (elapsed2 * 1e3) for k, v in prof2.items()}
    second_pass = (f"separate profiling pass of {pass_steps} steps right after the timed region, "
                   f"kernels launched one by one ({1e3 * elapsed2 / pass_steps:.2f} ms/step)")
    rl_spmv = {"bound": "hbm", "kernel": "spmv_kernel<3,false> (Jacobian SpMV, sliced block-ELL)",
               "achieved": gbs_spmv, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": gbs_spmv / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": b_spmv,
               "ms_per_launch": ms_spmv, "launches": prof2["spmv"][1],
               "share_of_profiling_pass": share2["spmv"], "measured": second_pass}
    rl_asm = {"bound": "hbm", "kernel": runner.assembly_kernel_name,
              "achieved": gbs_asm, "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": gbs_asm / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes": b_asm,
              "ms_per_launch": ms_asm, "launches": prof["assembly_FJ"][1],
              "ms_residual_only": ms_res, "share_of_timed_region": share["assembly_FJ"]}
    tr = pmc_traffic() if (world == 1 and n == 576) else {}
    rl_spmv["traffic"] = tr.get("spmv")
    rl_asm["traffic"] = next((v for k, v in tr.items() if "assemble_lean" in k),
                             next((v for k, v in tr.items() if "assemble_patch" in k), None))
    dominant, other = rl_asm, rl_spmv   # the assembly kernel is timed inside the timed region
    # BASELINE.json's target is quoted on the assembly + SpMV path together: algorithmic bytes of
    # all assemblies and Krylov SpMVs of a step over the time their kernels take
    n_asm = prof["assembly_FJ"][1] / args.steps
    n_res = prof["assembly_F"][1] / args.steps          # residual-only assemblies (final Newton checks)
    n_spmv = (n1[1] - n0[1]) / args.steps               # one Jacobian SpMV per GMRES iteration
    neq_, nv_, nc_ = sz["n_eq"], sz["n_vertices"], sz["n_cells"]
    b_res = nv_ * (16 + 24 * neq_) + nc_ * 12 + nv_ * neq_ * 8      # no matrix values / slots
    path_bytes = n_asm * b_asm + n_res * b_res + n_spmv * b_spmv
    path_ms = n_asm * ms_asm + n_res * ms_res + n_spmv * ms_spmv
    path_gbs = path_bytes / (path_ms * 1e-3) / 1e9
    copy_gbs = measured_copy_ceiling(torch.device("cuda", local_rank)) if rank == 0 else None

    out = {
        "metric": "BDF2 DOF-updates/sec (streamer_discharge 2D axisym)",
        "value": total_dofs * args.steps / elapsed,
        "unit": "DOF-updates/s",
        "timesteps_per_sec": args.steps / elapsed,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "streamer_discharge 2D axisymmetric, LFA, 3 equations "
                               "(ions, electrons, Poisson), analytic Bagheri-2018 seed",
                   "mesh": f"{n}x{n} right-diagonal, geometric grading {args.grading} towards "
                           f"the axis, per GPU",
                   "dofs_total": total_dofs, "vertices_per_gpu": sz["n_vertices"],
                   "dt_max": 5e-12, "newton_rtol": 1e-4, "gmres": "flexible, restart 30, rtol 1e-5 on the true residual, right-preconditioned: "
                   "field split, Chebyshev(6) block Jacobi on species (degree 4 once a Newton system needs >= 5 Krylov steps) + multigrid V(1,1) on the potential", "partition": runner.partition_name,
                   "host_placement": placement},
        "newton_iterations_per_step": (n1[0] - n0[0]) / args.steps,
        "gmres_iterations_per_step": (n1[1] - n0[1]) / args.steps,
        "roofline": dominant,
        "roofline_other": other,
        "assembly_plus_spmv": {"bound": "hbm", "achieved": path_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": path_gbs / HBM_PEAK_GBS, "assemblies_per_step": n_asm,
                               "residual_only_assemblies_per_step": n_res,
                               "spmv_per_step": n_spmv, "algorithmic_bytes_per_step": path_bytes,
                               "kernel_ms_per_step": path_ms,
                               "measured_copy_ceiling_GBs": copy_gbs,
                               "frac_of_measured_copy": (path_gbs / copy_gbs) if copy_gbs else None},
        "vcycle": {"ms_per_cycle": prof2["vcycle"][0] / max(prof2["vcycle"][1], 1),
                   "cycles": prof2["vcycle"][1], "share_of_profiling_pass": share2["vcycle"],
                   "levels": runner.multigrid_levels, "measured": second_pass},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_mesh, args.cpu_steps)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
