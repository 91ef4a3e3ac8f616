#!/usr/bin/env python3
"""Headline benchmark: BDF2 time steps of the streamer_discharge case on MI355X.

``python bench.py --gpus N --steps K --warmup W``.  A *step* is one accepted
adaptive BDF2 time step of examples/streamer_discharge/fedm-streamer.py:304-340
(shift states, Newton solve with F/J assembly and GMRES, error norm, step
controller) on the state resident in HBM.  Workload at N = 1 (BASELINE.json configs[3]):
2-D axisymmetric streamer on the mesh KIND the reference's case runs on (fedm-streamer.py:116,
Mesh('mesh.xml')): a locally refined unstructured mesh, 4 um in the streamer channel, written to
and read back from DOLFIN XML (341 280 vertices x 3 equations = 1 023 840 DOFs); N > 1: the same
mesh family with N times the vertices (spacing / sqrt(N): weak scaling), plus -- at N = 8 -- a
second record on BASELINE configs[4] (1152x1152 tensor-product mesh, ~4 M DOFs, strong scaling).
`--family tensor` puts the headline on round 1-3's 576x576 graded tensor-product mesh instead.
Prints ONE JSON line (rank 0).

The timed region: W warm-up steps, a checkpoint of the time loop on the device
(fedm_state_snapshot), >= 0.3 s of real stepping from it (clocks up, graphs captured), then the
window of EXACTLY K steps between two barriers R times, each from the checkpoint (the restore is
untimed).  ``ms_per_step`` is the MEDIAN window; the first window, the spread and the per-step
spans of the median window are in ``windows``.

Records in the line, besides the driver's contract:
* ``roofline`` / ``roofline_other`` / ``assembly_plus_spmv``: HIP-event timings of the hot kernels
  against the 8 TB/s HBM3E peak (and against the copy rate measured on this box);
* ``late_window``: the same K steps timed again from the developed streamer (step 200, t ~ 1 ns),
  where a step needs several times the Krylov iterations of the first steps;
* ``multi_gpu`` (N > 1): transport, ranks, halo exchanges / all-reduces per step and their latency;
* ``tensor_mesh`` (N = 1): the same K steps on the 576x576 graded tensor-product mesh (998 787 DOFs: the
  headline of rounds 1-3) with its own roofline blocks;
* ``roofline_beyond_infinity_cache`` (N = 1): assembly and SpMV fractions on the 1152x1152 mesh (4 M
  DOFs, Jacobian 670 MB > the 256 MiB Infinity Cache): rates that cannot come from the last-level cache;
* ``glow_discharge`` (N = 1): BASELINE configs[2] (LMEA, 141x141 crossed, 200 225 DOFs) with the roofline
  of its assembly kernels;
* ``cpu_baseline`` (N = 1): the C/OpenMP restatement under oracle/ on the same mesh.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

# multi-process GPU work on this pool needs dmabuf IPC (already exported by the launcher; kept here so a
# bare `python -m torch.distributed.run ... bench.py` from a clean shell behaves the same)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md, chip-level parameters (spec)
PMC_PROFILE = ROOT / "profiles" / "r04_pmc_traffic.json"


def library_ready():
    """The HIP library is built (or rebuilt when a source is newer) BEFORE torch is imported or the GPU
    is touched: the compiler driver exec()s its tools, which a process that has initialised the GPU --
    or any child of one running under rocprofv3, whose preloaded tool initialises it -- must not do on
    this pool.  Under a profiler a missing or stale library is an error, not a reason to compile."""
    import __graft_entry__ as entry
    if not entry._stale(entry.LIB):
        return entry
    profiled = any(k in os.environ for k in ("ROCPROFILER_REGISTER_FORCE_LOAD", "ROCP_TOOL_LIBRARIES",
                                             "ROCPROF_OUTPUT_PATH")) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if profiled:
        raise SystemExit(f"{entry.LIB} is missing or older than its sources: build it first, unprofiled "
                         "(python3 -c 'import __graft_entry__ as g; g.build()')")
    entry.build()
    return entry
KERNEL_SOURCES = ["kernels.hip", "assemble3.hip", "element.hpp", "element_lean.hpp", "prep.cpp", "fedm_internal.hpp",
                  "species_planes.hpp"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--family", choices=["unstructured", "tensor"], default="unstructured",
                    help="mesh family of the headline: the locally refined unstructured mesh (the kind the "
                         "reference's case loads) or the graded tensor-product mesh of rounds 1-3")
    ap.add_argument("--repeats", type=int, default=7, help="timed windows of K steps, each from the same checkpoint")
    ap.add_argument("--preroll", type=float, default=0.3, help="seconds of untimed stepping before the first window")
    ap.add_argument("--mesh", type=int, default=576, help="tensor-product meshes: cells per side (per GPU)")
    ap.add_argument("--grading", type=float, default=4.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-mesh", type=int, default=None, help="CPU baseline mesh (default: --mesh)")
    ap.add_argument("--cpu-steps", type=int, default=0, help="0: as many as --steps")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0: every core this process may use")
    ap.add_argument("--late-start", type=int, default=200,
                    help="accepted steps before the late window (0 disables it)")
    ap.add_argument("--configs4", choices=["auto", "on", "off"], default="auto",
                    help="second record on the ~4 M-DOF mesh of BASELINE configs[4] (auto: at 8 GPUs)")
    ap.add_argument("--configs4-mesh", type=int, default=1152, help="global cells per side of that record")
    ap.add_argument("--second-mesh", choices=["auto", "on", "off"], default="auto",
                    help="second record on the other mesh family (auto: on one GPU)")
    ap.add_argument("--mesh-spacing", type=float, default=4e-6,
                    help="finest spacing of that mesh [m] (4e-6: 341 280 vertices, 1 023 840 DOFs)")
    ap.add_argument("--big-mesh", type=int, default=1152,
                    help="cells per side of the single-GPU mesh beyond the Infinity Cache (0 disables the record)")
    ap.add_argument("--no-glow-discharge", action="store_true", help="skip the configs[2] record")
    # The data path across GPUs is RCCL.  When its set-up fails, every rank agrees on a host-staged
    # transport (gloo), which is correct but measures the host, not xGMI: the run then stops with a
    # non-zero exit code unless this flag says that such a number is wanted.
    ap.add_argument("--allow-fallback", action="store_true")
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


def residual_bytes(sz):
    """SURVEY 8(d), residual only: no matrix values, no cell slots."""
    neq, nv, nc = sz["n_eq"], sz["n_vertices"], sz["n_cells"]
    return nv * (16 + 24 * neq) + nc * 12 + nv * neq * 8


def kernel_source_sha():
    """Identity of the kernel sources a PMC profile belongs to (profiles/*_pmc_traffic.json carry
    it): counters cannot be read from inside the run, so the HBM traffic in the bench line comes
    from the committed rocprofv3 passes -- and is dropped when they were taken on other kernels."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update((ROOT / "fedm_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()[:16]


def pmc_traffic(workload):
    """HBM bytes per launch from the committed PMC passes of this same workload ("unstructured" /
    "tensor": the profile holds one block per mesh family; 2*FETCH_SIZE + WRITE_SIZE, gfx950 correction
    applied).  {} when the profile is absent or was measured on other kernel sources."""
    if not PMC_PROFILE.exists():
        return {}, "no committed PMC profile"
    prof = json.loads(PMC_PROFILE.read_text())
    sha = kernel_source_sha()
    if prof.get("kernel_source_sha") != sha:
        return {}, (f"{PMC_PROFILE.name} was measured on kernel sources {prof.get('kernel_source_sha')}, "
                    f"this run uses {sha}: traffic dropped")
    block = prof.get("workloads", {}).get(workload)
    if block is None:
        return {}, f"{PMC_PROFILE.name} has no pass on the {workload} mesh"
    # (most launches first: of two instantiations of a kernel -- the first full assembly of a context writes every
    # plane, the later ones keep the constant ones -- `pick` then finds the steady-state one)
    ranked = sorted(block["kernels"].items(), key=lambda kv: -kv[1].get("launches_sampled", 0))
    out = {n: v["traffic_bytes_corrected"] for n, v in ranked if "traffic_bytes_corrected" in v}
    return out, f"{PMC_PROFILE.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this mesh, kernel sources {sha})"


def pick(tr, *needles):
    for n in needles:
        for k, v in tr.items():
            if n in k:
                return v
    return None


def measured_copy_ceiling(device_index):
    """Device copy rate on this box (read + write bytes / time): the library's own 16-byte-per-lane
    grid-stride copy between two 1 GiB buffers, timed with HIP events (`fedm_copy_bandwidth`) -- the
    practical ceiling next to the 8 TB/s specification (the guide measures 6.29 TB/s this way)."""
    import ctypes as C
    from fedm_amd import _lib
    gbs = C.c_double()
    rc = _lib.load().fedm_copy_bandwidth(int(device_index), 1 << 30, 10, C.byref(gbs))
    return gbs.value if rc == 0 else None


def cpu_baseline(mesh, steps, threads, label):
    """oracle/cpu_backend (C + OpenMP: coloured element loop -> block CSR -> Newton -> flexible
    GMRES with the same field split, Chebyshev sweeps and multigrid V-cycle as the device path),
    'CPU restatement, not FEniCS', on the SAME mesh as the headline, timed on this box's host cores."""
    from oracle import cpu_backend
    from fedm_amd.device import locality_order
    # the vertex order the device path gives itself (compact slices of 64): the CPU's caches profit alike
    order = locality_order(mesh.coords, mesh.cells)
    inv = np.empty(order.size, dtype=np.int64)
    inv[order] = np.arange(order.size)
    return cpu_backend.bench_mesh(mesh.coords[order], inv[mesh.cells].astype(np.int32), steps, threads,
                                  label + "; vertices in the device path's locality order")


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


def timed_steps(runner, steps, barrier, torch, dist, distributed):
    """K steps between two barriers; wall time is the MAX over ranks.  Returns (seconds, Newton
    iterations, GMRES iterations, assembly profile, per-step spans in ms on this rank)."""
    runner.profile(1)             # HIP events around the assembly kernels, on the library's stream
    barrier()
    t0 = time.perf_counter()
    n0 = (runner.newton_iterations, runner.linear_iterations)
    marks = [t0]
    for _ in range(steps):
        runner.step()             # (returns once the step's last norm has reached the host)
        marks.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    n1 = (runner.newton_iterations, runner.linear_iterations)
    prof = runner.profile_read()
    runner.profile(False)
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    spans = [1e3 * (b - a) for a, b in zip(marks[:-1], marks[1:])]
    return elapsed, n1[0] - n0[0], n1[1] - n0[1], prof, spans


def timed_windows(runner, steps, repeats, preroll_s, barrier, torch, dist, distributed):
    """The contract's window -- EXACTLY `steps` steps between two barriers -- `repeats` times from ONE
    checkpoint of the time loop (device-side copy of the three states + the script's scalars; the restore
    is outside the timed region), after `preroll_s` seconds of untimed stepping from the same checkpoint
    (clock ramp, graph capture, first-touch effects: the first 31-ms window of round 3's driver run was 20 %
    slower than the same window on a warm chip).  Returns the MEDIAN window's record plus the spread; the
    runner is left at the end of the last window."""
    snap = runner.snapshot()
    # pre-roll: whole windows, their number agreed between the ranks; with the in-run kernel timing on, as in the
    # timed windows (its first use is not free: event pool, the queue's profiling mode)
    runner.profile(1)
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step()
    first = time.perf_counter() - t0
    extra = int(min(20, max(0, np.ceil(preroll_s / max(first, 1e-4)) - 1)))
    if distributed:
        e = torch.tensor([float(extra)], dtype=torch.float64, device="cuda")
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        extra = int(e.item())
    for _ in range(extra):
        runner.restore(snap)
        for _ in range(steps):
            runner.step()
    runner.profile_read()
    runner.profile(False)
    windows = []
    prof_sum = None
    for r in range(max(1, repeats)):
        runner.restore(snap)
        el, nw, gm, prof, spans = timed_steps(runner, steps, barrier, torch, dist, distributed)
        windows.append(dict(elapsed=el, newton=nw, gmres=gm, spans=spans))
        prof_sum = prof if prof_sum is None else {k: (prof_sum[k][0] + v[0], prof_sum[k][1] + v[1]) for k, v in prof.items()}
    order = sorted(range(len(windows)), key=lambda i: windows[i]["elapsed"])
    med = windows[order[len(order) // 2]]
    ms = [1e3 * w["elapsed"] / steps for w in windows]
    record = {"repeats": len(windows), "window_ms_per_step": ms, "window_ms_min": min(ms), "window_ms_max": max(ms),
              "window_spread": (max(ms) - min(ms)) / min(ms), "first_window_ms_per_step": ms[0],
              "median_window_step_spans_ms": med["spans"], "preroll_windows": 1 + extra,
              "preroll": f"{1 + extra} untimed windows of {steps} steps from the checkpoint before the first timed one",
              "how": "every window restarts from the same device-side checkpoint taken after the warm-up steps "
                     "(fedm_state_snapshot / fedm_state_restore, untimed); ms_per_step and value are the median window's"}
    return med["elapsed"], med["newton"], med["gmres"], prof_sum, len(windows), record


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    entry = library_ready()          # before torch: nothing may exec once the GPU is initialised
    import torch
    import torch.distributed as dist
    # one rank per GPU: LOCAL_RANK is the device index when a rank sees every GPU of the node (the usual
    # torch.distributed.run set-up); a launcher that shows each rank only its own GPU leaves index 0
    n_visible = torch.cuda.device_count()
    if n_visible and local_rank >= n_visible:
        local_rank %= n_visible
    from fedm_amd.cases import streamer

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

    def refined(spacing):
        """The locally refined unstructured mesh through DOLFIN XML (every rank generates the same one).  The
        refined channel keeps its radius (100 x --mesh-spacing) when the spacing shrinks for N > 1: N times the
        vertices."""
        import tempfile
        channel = (0.0, 100.0 * args.mesh_spacing) + streamer.CHANNEL[2:]
        if distributed:
            # millions of vertices on every rank: one Delaunay pass instead of three, no XML round trip (minutes at N = 8)
            return streamer.refined_mesh(spacing, growth=0.1, channel=channel, retriangulate=False)
        with tempfile.TemporaryDirectory(prefix="fedm_mesh_") as tmp:
            return streamer.refined_mesh(spacing, growth=0.1, xml_path=Path(tmp) / "mesh.xml", channel=channel)

    def check_transport(r):
        if r.transport != "rccl" and not (args.allow_fallback or args.rehearse_on_one_gpu):
            if rank == 0:
                print(json.dumps({"error": "RCCL transport unavailable", "reason": r.fallback_reason,
                                  "hint": "--allow-fallback times the host-staged (gloo) transport instead"}))
            dist.destroy_process_group()
            raise SystemExit(3)
        return r

    def make_runner(family, n_per_gpu=None, global_n=None, spacing=None):
        """(runner, mesh or None, seconds spent generating the mesh)"""
        t_m = time.perf_counter()
        if family == "unstructured":
            msh = refined(spacing)
            t_mesh = time.perf_counter() - t_m
            if distributed:
                from fedm_amd.cases import streamer_distributed
                r = streamer_distributed.Runner(None, rank, world, local_rank, mesh=msh,
                                                transport="torch" if args.rehearse_on_one_gpu else "rccl")
                return check_transport(r), msh, t_mesh
            return streamer.Stepper(streamer.device_problem(msh.coords, msh.cells, device=local_rank)), msh, t_mesh
        if distributed:
            from fedm_amd.cases import streamer_distributed
            r = streamer_distributed.Runner(None, rank, world, local_rank, args.grading,
                                            n_per_gpu=n_per_gpu, global_n=global_n,
                                            transport="torch" if args.rehearse_on_one_gpu else "rccl")
            return check_transport(r), None, 0.0
        msh = streamer.mesh(n_per_gpu, args.grading)
        return streamer.Stepper(streamer.device_problem(msh.coords, msh.cells, device=local_rank)), msh, 0.0

    copy_gbs = measured_copy_ceiling(local_rank) if rank == 0 else None

    def hot_path(runner, steps, warmup, tr, tr_source, repeats):
        """W warm-up steps, the window of K timed steps `repeats` times from one checkpoint (HIP events around
        the assembly kernels, on the library's stream), then a second, untimed pass with plain launches for the
        kernels inside the Krylov iterations (events cannot sit inside the replayed per-iteration graphs).
        Returns the record pieces."""
        for _ in range(warmup):
            runner.step()
        before = runner.prob.comm_stats() if distributed else None
        elapsed, newton, gmres, prof, n_win, windows = timed_windows(runner, steps, repeats, args.preroll, barrier,
                                                                    torch, dist, distributed)
        prof_steps = steps * n_win           # the assembly events cover every timed window
        comm0 = runner.prob.comm_stats() if distributed else None
        if distributed:      # what travelled inside ONE timed window (set-up, warm-up and pre-roll apart)
            for key in ("halo_exchanges", "allreduces", "allreduce_bytes", "halo_bytes"):
                if key in comm0:
                    comm0[key + "_timed"] = (comm0[key] - before[key]) / (n_win + windows["preroll_windows"])
        pass_steps = max(1, min(steps, 5))
        runner.profile(2)
        barrier()
        t1 = time.perf_counter()
        for _ in range(pass_steps):
            runner.step()
        barrier()
        elapsed2 = time.perf_counter() - t1
        prof2 = runner.profile_read()
        runner.profile(False)
        sz = runner.sizes()
        # average launch duration of the hot kernels inside the timed windows
        ms_asm = prof["assembly_FJ"][0] / max(prof["assembly_FJ"][1], 1)
        ms_spmv = prof2["spmv"][0] / max(prof2["spmv"][1], 1)
        ms_res = prof["assembly_F"][0] / max(prof["assembly_F"][1], 1)
        b_spmv, b_asm, b_res = spmv_bytes(sz), assembly_bytes(sz), residual_bytes(sz)
        gbs_spmv = b_spmv / (ms_spmv * 1e-3) / 1e9
        gbs_asm = b_asm / (ms_asm * 1e-3) / 1e9
        share = {k: v[0] / (elapsed * n_win * 1e3) for k, v in prof.items()}
        share2 = {k: v[0] / (elapsed2 * 1e3) for k, v in prof2.items()}
        second_pass = (f"separate profiling pass of {pass_steps} steps right after the timed region, "
                       f"kernels launched one by one ({1e3 * elapsed2 / pass_steps:.2f} ms/step)")

        def moved(traffic, ms):          # HBM bytes the counters saw, over the kernel's time
            return traffic / (ms * 1e-3) / 1e9 if traffic and ms else None

        def below_ceiling(gbs):          # fraction of the box's own copy rate; a "ceiling" a kernel exceeds is noise
            return gbs / copy_gbs if copy_gbs and gbs <= copy_gbs else None
        t_spmv = pick(tr, "fedm::spmv_kernel<3, false", "fedm::spmv_kernel<3,false")
        t_asm = pick(tr, "assemble_lean3", "assemble_lean2", "assemble_patch")
        t_res = pick(tr, "residual_lean3", "residual_lean2", "residual_patch")
        rl_spmv = {"bound": "hbm", "kernel": "spmv_kernel<3,false,ZMASK> (Jacobian SpMV, sliced block-ELL; structurally zero value planes not loaded)",
                   "achieved": gbs_spmv, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": gbs_spmv / HBM_PEAK_GBS,
                   "frac_of_measured_copy": below_ceiling(gbs_spmv),
                   "traffic": t_spmv, "achieved_traffic_GBs": moved(t_spmv, ms_spmv),
                   "algorithmic_bytes": b_spmv,
                   # SURVEY 8(d) counts every structural block whole; the kernel does not load value
                   # planes that are structurally zero for the model (d(electron row)/d(ion density))
                   "bytes_not_loaded_structural_zero_planes": sz["nnz_blocks"] * 8 * sz.get("zero_planes", 0),
                   "ms_per_launch": ms_spmv, "launches": prof2["spmv"][1],
                   "share_of_profiling_pass": share2["spmv"], "measured": second_pass,
                   "timed_region_note": ("inside the replayed Krylov steps of the timed region the same product runs as "
                                         "spmv_dots_kernel<3,ZMASK,K>: the wave that has formed a slice's rows also "
                                         "multiplies them with the K-1 basis vectors (the step's dot products; w is not "
                                         "read back) -- 1-3 us longer than the plain product timed here, see "
                                         "profiles/r04_kernel_stats.csv"),
                   "infinity_cache_note": ("the Jacobian of this mesh (%.0f MB) fits the 256 MiB Infinity Cache and the "
                                           "counters include its hits: the cache-free rates are in "
                                           "'roofline_beyond_infinity_cache'" % (sz["stored_blocks"] * 72 / 1e6))
                                          if sz["stored_blocks"] * 72 < 256 * 2 ** 20 else
                                          "the Jacobian of this mesh does not fit the 256 MiB Infinity Cache"}
        rl_asm = {"bound": "hbm", "kernel": runner.assembly_kernel_name,
                  "achieved": gbs_asm, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": gbs_asm / HBM_PEAK_GBS,
                  "frac_of_measured_copy": below_ceiling(gbs_asm),
                  "traffic": t_asm, "achieved_traffic_GBs": moved(t_asm, ms_asm),
                  "traffic_source": tr_source, "algorithmic_bytes": b_asm,
                  # ... and the assembly neither recomputes nor rewrites planes that cannot change
                  # (potential-potential: geometry only; structurally zero species planes)
                  "bytes_not_rewritten_kept_planes": sz["stored_blocks"] * 8 * sz.get("kept_planes", 0),
                  "ms_per_launch": ms_asm, "launches": prof["assembly_FJ"][1],
                  "ms_residual_only": ms_res, "residual_only_algorithmic_bytes": b_res,
                  "residual_only_frac": b_res / (ms_res * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_res else None,
                  "residual_only_traffic": t_res,
                  "share_of_timed_region": share["assembly_FJ"]}

        # BASELINE.json's target is quoted on the assembly + SpMV path together: algorithmic bytes of
        # all assemblies and Krylov SpMVs of a step over the time their kernels take
        def path_record(prof_w, gmres_w, steps_w):
            n_asm = prof_w["assembly_FJ"][1] / steps_w
            n_res = prof_w["assembly_F"][1] / steps_w        # residual-only assemblies (final Newton checks)
            n_spmv = gmres_w / steps_w                       # one Jacobian SpMV per GMRES iteration
            m_asm = prof_w["assembly_FJ"][0] / max(prof_w["assembly_FJ"][1], 1)
            m_res = prof_w["assembly_F"][0] / max(prof_w["assembly_F"][1], 1)
            path_bytes = n_asm * b_asm + n_res * b_res + n_spmv * b_spmv
            path_ms = n_asm * m_asm + n_res * m_res + n_spmv * ms_spmv
            path_gbs = path_bytes / (path_ms * 1e-3) / 1e9
            return {"bound": "hbm", "achieved": path_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": path_gbs / HBM_PEAK_GBS, "assemblies_per_step": n_asm,
                    "residual_only_assemblies_per_step": n_res, "spmv_per_step": n_spmv,
                    "algorithmic_bytes_per_step": path_bytes, "kernel_ms_per_step": path_ms,
                    "measured_copy_ceiling_GBs": copy_gbs,
                    "frac_of_measured_copy": below_ceiling(path_gbs)}
        vcycle = {"ms_per_cycle": prof2["vcycle"][0] / max(prof2["vcycle"][1], 1),
                  "cycles": prof2["vcycle"][1], "share_of_profiling_pass": share2["vcycle"],
                  "levels": runner.multigrid_levels, "measured": second_pass}
        return dict(elapsed=elapsed, newton=newton, gmres=gmres, prof=prof, sz=sz, comm0=comm0, rl_asm=rl_asm,
                    rl_spmv=rl_spmv, path=path_record(prof, gmres * n_win, prof_steps), path_record=path_record,
                    vcycle=vcycle, windows=windows)

    def late_window(runner, hp, steps):
        """The developed streamer: the same K steps from step `late_start` on (one window: every step there
        is a different system)."""
        while runner.steps < args.late_start:
            runner.step()
        t_late = runner.t
        pol0 = runner.prob.fieldsplit_policy()
        l_elapsed, l_newton, l_gmres, l_prof, _ = timed_steps(runner, steps, barrier, torch, dist, distributed)
        pol1 = runner.prob.fieldsplit_policy()
        return {"preconditioner_sets": {"policy": pol1["policy"],
                                        "solves_main_set": pol1["solves_main_set"] - pol0["solves_main_set"],
                                        "solves_alternative_set": pol1["solves_alternative_set"] - pol0["solves_alternative_set"]},
                "what": f"{steps} accepted steps timed the same way from step {args.late_start + 1} on "
                        f"(t = {t_late:.3e} s: the streamer has formed and propagates)",
                "value": runner.total_dofs * steps / l_elapsed, "unit": "DOF-updates/s",
                "timesteps_per_sec": steps / l_elapsed, "ms_per_step": 1e3 * l_elapsed / steps,
                "newton_iterations_per_step": l_newton / steps,
                "gmres_iterations_per_step": l_gmres / steps,
                "assembly_plus_spmv": hp["path_record"](l_prof, l_gmres, steps)}

    def pattern_of(sz):
        return {"stored_blocks_over_nnz_blocks": sz["stored_blocks"] / sz["nnz_blocks"],
                "max_block_columns_per_slice": sz["max_patch_width"],
                "max_cells_per_patch": sz["max_patch_cells"], "max_staged_vertices_per_patch": sz["max_patch_verts"],
                "cell_visits_over_cells": sz["cell_visits"] / sz["n_cells"],
                "patch_workgroup_threads": sz["patch_threads"], "assembly_variant": sz["assembly_variant"]}

    def mesh_text(family, n, spacing):
        if family == "unstructured":
            return (f"locally refined unstructured mesh: Delaunay triangulation of nested hexagonal lattices, spacing "
                    f"{spacing:g} m in the streamer channel (r < {100.0 * args.mesh_spacing:g} m), growing 0.1 per unit distance "
                    "outside; " + ("generated on every rank (one Delaunay pass, no XML round trip); " if distributed else
                                   "written to DOLFIN XML and read back through the mesh reader (the way of the reference's "
                                   "Mesh('mesh.xml'), fedm-streamer.py:116); ") +
                    "vertices ordered by recursive bisection in the metric of the local spacing (device.locality_order)")
        return f"{n}x{n} right-diagonal tensor-product mesh, geometric grading {args.grading} towards the axis"

    # ---- the headline: K steps right after the warm-up (SURVEY 8(d)'s window) -----------------------
    family = args.family
    n = args.mesh
    spacing = args.mesh_spacing / np.sqrt(world)          # weak scaling: N times the vertices
    t_setup = time.perf_counter()
    runner, hmesh, t_mesh = make_runner(family, n_per_gpu=n, spacing=spacing)
    runner.initialise()
    setup_s = time.perf_counter() - t_setup - t_mesh
    profiled = world == 1 and ((family == "unstructured" and args.mesh_spacing == 4e-6) or (family == "tensor" and n == 576))
    tr, tr_source = pmc_traffic(family) if profiled else ({}, "not a profiled workload")
    hp = hot_path(runner, args.steps, args.warmup, tr, tr_source, args.repeats)
    elapsed, newton, gmres, sz, comm0 = hp["elapsed"], hp["newton"], hp["gmres"], hp["sz"], hp["comm0"]
    rl_asm, rl_spmv = hp["rl_asm"], hp["rl_spmv"]
    total_dofs = runner.total_dofs

    gmres_text = ("flexible, restart 30, rtol 1e-5 on the true residual, right-preconditioned: field split, "
                  "Chebyshev(6) block Jacobi on species (degree 4 once a Newton system needs >= 5 Krylov "
                  "steps; the sweeps run 3 + 2 per launch on tiles of 8 matrix slices whose vertex layers sit in "
                  "LDS) + multigrid V(1,1) on the potential; Jacobian product and the step's dot products in one "
                  "kernel; two Krylov steps per graph launch when the previous solve needed both")
    try:
        tiles = runner.prob.fieldsplit_tiles()
    except Exception:
        tiles = None
    weak = "" if world == 1 else (f"; weak scaling: the mesh family at {world} x the vertices (spacing / sqrt({world}))"
                                  if family == "unstructured" else f"; weak scaling: {n}x{n} cells per GPU")
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
                               "(ions, electrons, Poisson), analytic Bagheri-2018 seed, on a "
                               + ("locally refined unstructured mesh (the reference's mesh kind)" if family == "unstructured"
                                  else "graded tensor-product mesh") + weak,
                   "baseline_config": "configs[3] (~1M DOFs, 1 GPU)" if world == 1 else
                                      f"configs[3]'s mesh size per GPU x {world} GPUs (configs[4] itself: see 'configs4')",
                   "mesh": mesh_text(family, n, spacing),
                   "dofs_total": total_dofs, "vertices_per_gpu": sz["n_vertices"], "cells_per_gpu": sz["n_cells"],
                   "dt_max": 5e-12, "newton_rtol": 1e-4, "gmres": gmres_text, "fieldsplit_tiles": tiles,
                   "partition": runner.partition_name, "host_placement": placement,
                   "pattern": pattern_of(sz),
                   "mesh_seconds": t_mesh, "setup_seconds": setup_s},
        "window": f"steps {args.warmup + 1}..{args.warmup + args.steps} from the initial condition "
                  "(SURVEY 8(d)), repeated from a checkpoint: 'windows'; the developed streamer is 'late_window'",
        "windows": hp["windows"],
        "newton_iterations_per_step": newton / args.steps,
        "gmres_iterations_per_step": gmres / args.steps,
        "roofline": rl_asm,
        "roofline_other": rl_spmv,
        "assembly_plus_spmv": hp["path"],
        "vcycle": hp["vcycle"],
        "measured_copy_ceiling_GBs": copy_gbs,
        "copy_ceiling_how": "fedm_copy_bandwidth: the fastest of six 16-byte-per-lane copy kernels (1-8 loads in flight per "
                            "lane, plain / non-temporal), 2 x 1 GiB, HIP events",
        "run_to_reference_end_time": ("profiles/r04_refined_run_4um_1M_dofs.json: this mesh carries the streamer to the reference's "
                                      "T_final = 1.4e-8 s (2801 accepted steps, none rejected)") if family == "unstructured" else None,
    }
    if args.late_start > 0:
        out["late_window"] = late_window(runner, hp, args.steps)
        out["sustained_timesteps_per_sec"] = out["late_window"]["timesteps_per_sec"]

    # ---- multi-GPU plumbing: what travelled, and what one exchange / reduction costs ------------
    if distributed:
        cs = comm0
        lat = {}
        for name, kind in (("halo_state_us", 0), ("halo_scalar_us", 1), ("allreduce_32_doubles_us", 2)):
            barrier()
            lat[name] = 1e3 * runner.prob.time_comm(kind, 50)
        per_step = lambda key: cs[key + "_timed"] / args.steps if key + "_timed" in cs else None    # noqa: E731
        out["multi_gpu"] = {
            "transport": runner.transport, "transport_requested": runner.transport_requested,
            "ranks_in_communicator": cs["ranks"], "neighbours_rank0": cs["neighbours"],
            "halo_exchanges_per_step": per_step("halo_exchanges"),
            "allreduces_per_step": per_step("allreduces"),
            "halo_bytes_per_step": per_step("halo_bytes"),
            "allreduce_bytes_per_step": per_step("allreduce_bytes"),
            "halo_exchanges_per_krylov_step": cs["halo_exchanges_timed"] / max(gmres, 1),
            "allreduces_per_krylov_step": cs["allreduces_timed"] / max(gmres, 1),
            "halo_depth": getattr(runner, "halo_depth", 1),
            "counting": "collectives of rank 0 per timed window (the initial Poisson solve and the warm-up apart); "
                        "per Krylov step: all of a time step's collectives -- Newton norms, state halos -- over its Krylov steps",
            "assembly_patches_rank0": {"interior (assembled while the state halo travels)": cs["interior_patches"],
                                       "boundary": cs["boundary_patches"]},
            **lat,
            "halo_ms_per_step_if_serial": per_step("halo_exchanges") * lat["halo_state_us"] * 1e-3,
            "allreduce_ms_per_step_if_serial": per_step("allreduces") * lat["allreduce_32_doubles_us"] * 1e-3,
            "note": "latencies are back-to-back micro-benchmarks on the compute stream after the run; in the "
                    "run the exchanges overlap interior SpMV rows / sweeps / assembly patches"}

    # ---- the same K steps on the other mesh family (one GPU) ---------------------------------------
    if world == 1 and args.second_mesh != "off":
        other = "tensor" if family == "unstructured" else "unstructured"
        del runner
        t_o = time.perf_counter()
        orun, omesh, o_tmesh = make_runner(other, n_per_gpu=n, spacing=args.mesh_spacing)
        orun.initialise()
        o_setup = time.perf_counter() - t_o - o_tmesh
        o_prof = (other == "unstructured" and args.mesh_spacing == 4e-6) or (other == "tensor" and n == 576)
        otr, otr_source = pmc_traffic(other) if o_prof else ({}, "not a profiled workload")
        oh = hot_path(orun, args.steps, args.warmup, otr, otr_source, min(args.repeats, 3))
        osz = oh["sz"]
        rec = {"workload": "the same streamer case on the " + mesh_text(other, n, args.mesh_spacing),
               "vertices": osz["n_vertices"], "cells": osz["n_cells"], "dofs_total": orun.total_dofs,
               "value": orun.total_dofs * args.steps / oh["elapsed"], "unit": "DOF-updates/s",
               "timesteps_per_sec": args.steps / oh["elapsed"], "ms_per_step": 1e3 * oh["elapsed"] / args.steps,
               "windows": oh["windows"],
               "newton_iterations_per_step": oh["newton"] / args.steps,
               "gmres_iterations_per_step": oh["gmres"] / args.steps,
               "roofline": oh["rl_asm"], "roofline_other": oh["rl_spmv"], "assembly_plus_spmv": oh["path"],
               "vcycle": oh["vcycle"], "pattern": pattern_of(osz), "mesh_seconds": o_tmesh, "setup_seconds": o_setup}
        if args.late_start > 0:
            rec["late_window"] = late_window(orun, oh, args.steps)
        out["tensor_mesh" if other == "tensor" else "unstructured"] = rec
        runner = orun

    # ---- rates that cannot come from the Infinity Cache: the 4 M-DOF mesh on ONE GPU ------------------
    if world == 1 and args.big_mesh > 0:
        del runner
        t_b = time.perf_counter()
        bmesh = streamer.mesh(args.big_mesh, args.grading)
        brun = streamer.Stepper(streamer.device_problem(bmesh.coords, bmesh.cells, device=local_rank))
        brun.initialise()
        b_setup = time.perf_counter() - t_b
        bsteps = max(2, min(args.steps, 5))
        bh = hot_path(brun, bsteps, 1, {}, "no PMC pass on this mesh", 1)
        keep = ("achieved", "frac", "frac_of_measured_copy", "algorithmic_bytes", "ms_per_launch", "launches")
        out["roofline_beyond_infinity_cache"] = {
            "workload": f"{args.big_mesh}x{args.big_mesh} graded mesh on one GPU ({brun.total_dofs} DOFs; Jacobian values "
                        f"{bh['sz']['stored_blocks'] * 72 / 1e6:.0f} MB, more than twice the 256 MiB Infinity Cache)",
            "assembly_FJ": {k: bh["rl_asm"][k] for k in keep},
            "assembly_F_ms": bh["rl_asm"]["ms_residual_only"],
            "spmv": {k: bh["rl_spmv"][k] for k in keep},
            "assembly_plus_spmv_frac": bh["path"]["frac"],
            "timesteps_per_sec": bsteps / bh["elapsed"], "ms_per_step": 1e3 * bh["elapsed"] / bsteps,
            "gmres_iterations_per_step": bh["gmres"] / bsteps, "setup_seconds": b_setup,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "bound": "hbm"}
        runner = brun

    # ---- BASELINE configs[2]: glow discharge (LMEA), 141 x 141 crossed = 200 225 DOFs ------------------
    if world == 1 and not args.no_glow_discharge:
        import contextlib
        import io
        from fedm_amd.cases import glow_discharge as gdc
        del runner
        runner = None
        with contextlib.redirect_stdout(io.StringIO()):
            case = gdc.Case(nx=141, ny=141, T_final=1.0, device=local_rank)
        case.step()
        case.prob.profile(1)
        n0, l0 = case.newton_iterations, case.linear_iterations
        torch.cuda.synchronize()
        t_g = time.perf_counter()
        gsteps = 10
        for _ in range(gsteps):
            case.step()
        torch.cuda.synchronize()
        g_el = time.perf_counter() - t_g
        gp = case.prob.profile_read()
        case.prob.profile(False)
        gsz = case.prob.sizes()
        nv, nc, neq, nnzb = gsz["n_vertices"], gsz["n_cells"], gsz["n_eq"], gsz["nnz_blocks"]
        n_fields = case.prob.model.n_fields
        # SURVEY 8(d) with the LMEA's nodal coefficient fields: + n_fields * Nv * 8
        g_bytes = nv * (16 + 24 * neq) + nc * (12 + 36) + nnzb * neq * neq * 8 + nv * neq * 8 + n_fields * nv * 8
        g_ms = gp["assembly_FJ"][0] / max(gp["assembly_FJ"][1], 1)
        g_gbs = g_bytes / (g_ms * 1e-3) / 1e9
        gtr, gtr_source = pmc_traffic("glow_discharge")
        import re as _re
        # the F + J pair only: element blocks (STORE = 1..3) + the gather of the matrix entries
        gtr = {k: v for k, v in gtr.items()
               if _re.search(r"gd_jacobian_rows_kernel<\d+, ?[123],", k) or _re.search(r"gd_gather(_dest(_rows)?)?_kernel<", k)}
        out["glow_discharge"] = {
            "workload": "BASELINE configs[2]: argon glow discharge, LMEA (energy + 3 particle balances + Poisson), "
                        "141x141 crossed mesh, device-resident per-step pipeline (fedm_amd.cases.glow_discharge)",
            "dofs_total": case.prob.n, "timesteps_per_sec": gsteps / g_el, "ms_per_step": 1e3 * g_el / gsteps,
            "value": case.prob.n * gsteps / g_el, "unit": "DOF-updates/s",
            "newton_iterations_per_step": (case.newton_iterations - n0) / gsteps,
            "gmres_iterations_per_step": (case.linear_iterations - l0) / gsteps,
            "roofline": {"bound": "hbm", "kernel": case.prob.gd_assembly_kernel_name(),
                         "achieved": g_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": g_gbs / HBM_PEAK_GBS,
                         "algorithmic_bytes": g_bytes, "ms_per_launch": g_ms, "launches": gp["assembly_FJ"][1],
                         "traffic": sum(gtr.values()) if gtr else None, "traffic_source": gtr_source}}
        case.prob.close()

    # ---- BASELINE configs[4]: ~4 M DOFs over 8 GPUs (strong-scaled counterpart of the line above) ----
    want4 = args.configs4 == "on" or (args.configs4 == "auto" and world == 8)
    if want4 and distributed:
        del runner
        r4, _, _ = make_runner("tensor", global_n=args.configs4_mesh)
        r4.initialise()
        for _ in range(args.warmup):
            r4.step()
        b4 = r4.prob.comm_stats()
        e4, nw4, gm4, _, _ = timed_steps(r4, args.steps, barrier, torch, dist, distributed)
        a4 = r4.prob.comm_stats()
        out["configs4"] = {
            "workload": f"BASELINE configs[4]: streamer_discharge, {args.configs4_mesh}x{args.configs4_mesh} "
                        f"global mesh over {world} GPUs",
            "dofs_total": r4.total_dofs, "value": r4.total_dofs * args.steps / e4, "unit": "DOF-updates/s",
            "timesteps_per_sec": args.steps / e4, "ms_per_step": 1e3 * e4 / args.steps,
            "newton_iterations_per_step": nw4 / args.steps, "gmres_iterations_per_step": gm4 / args.steps,
            "halo_exchanges_per_step": (a4["halo_exchanges"] - b4["halo_exchanges"]) / args.steps,
            "allreduces_per_step": (a4["allreduces"] - b4["allreduces"]) / args.steps,
            "halo_bytes_per_step": (a4.get("halo_bytes", 0) - b4.get("halo_bytes", 0)) / args.steps,
            "allreduce_bytes_per_step": (a4.get("allreduce_bytes", 0) - b4.get("allreduce_bytes", 0)) / args.steps,
            "partition": r4.partition_name}
        runner = r4

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:    # (a reported baseline: its failure must not take the measured record with it)
            label = mesh_text(family, n, spacing)
            cmesh = hmesh if hmesh is not None else streamer.mesh(args.cpu_mesh or n, args.grading)
            out["cpu_baseline"] = cpu_baseline(cmesh, args.cpu_steps or args.steps, args.cpu_threads, label)
        except Exception as exc:                          # noqa: BLE001 - reported in the record
            out["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
