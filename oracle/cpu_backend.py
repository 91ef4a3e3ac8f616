"""C/OpenMP restatement of the hot path (oracle/cpu/fedm_cpu.c) -- TEST INFRASTRUCTURE AND THE
REPORTED CPU BASELINE ONLY: imported by tests/, __graft_entry__ (build) and bench.py's
``cpu_baseline`` leg; nothing under fedm_amd/ imports it.

"CPU restatement, not FEniCS" (BASELINE.md 3): the same algorithm as the device
library -- coloured element loop -> block CSR -> Newton (PETSc newtonls rules,
fedm/functions.py:1047) -> flexible GMRES with the field split (Chebyshev sweeps on the species
block, smoothed-aggregation V(1,1) on the constant potential block) -- in C with OpenMP, driven by
the script-level time loop of examples/streamer_discharge/fedm-streamer.py:304-340
(:mod:`oracle.controller`).  Checked against :mod:`oracle.forms` in tests/test_cpu_backend.py.
"""
import ctypes as C
import os
import subprocess
import time
from pathlib import Path

import numpy as np
import scipy.sparse as sp

from . import controller
from . import streamer as ost
from .forms import elementary_charge, epsilon_0
from .mesh import graded_axis, mark_boundaries, rectangle_right

HERE = Path(__file__).resolve().parent
SRC = HERE / "cpu" / "fedm_cpu.c"
LIB = HERE / "_build" / "libfedm_cpu.so"
MAXS, MAXR, MAXT, MAXTAG = 4, 8, 6, 8


class TermSumC(C.Structure):
    _fields_ = [("n_terms", C.c_int32), ("pad_", C.c_int32), ("c", C.c_double * MAXT),
                ("p", C.c_double * MAXT), ("q", C.c_double * MAXT), ("r", C.c_double * MAXT)]


class ModelC(C.Structure):
    _fields_ = [("ns", C.c_int32), ("n_reactions", C.c_int32), ("axisymmetric", C.c_int32), ("n_tags", C.c_int32),
                ("eq_type", C.c_int32 * MAXS), ("Z", C.c_double * MAXS),
                ("mu", TermSumC * MAXS), ("D", TermSumC * MAXS), ("k", TermSumC * MAXR),
                ("power", (C.c_int32 * MAXS) * MAXR), ("net", (C.c_int32 * MAXS) * MAXR),
                ("charge_over_eps", C.c_double), ("bc_neumann", (C.c_int32 * MAXS) * MAXTAG)]


class CsrC(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("n_cols", C.c_int32), ("indptr", C.POINTER(C.c_int64)),
                ("indices", C.POINTER(C.c_int32)), ("values", C.POINTER(C.c_double))]


class ReportC(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("linear_iterations", C.c_int32),
                ("reason", C.c_int32), ("fnorm0", C.c_double), ("fnorm", C.c_double)]


def build(force=False):
    """gcc -O3 -fopenmp, x86-64-v3 (AVX2/FMA: what every host of the GPU pool has; the file is
    built here and travels to the GPU box like the HIP library)."""
    if LIB.exists() and not force and LIB.stat().st_mtime >= SRC.stat().st_mtime:
        return LIB
    LIB.parent.mkdir(exist_ok=True)
    cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC", "-shared", "-std=gnu11",
           "-o", str(LIB), str(SRC), "-lm"]
    subprocess.run(cmd, check=True)
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        lib = C.CDLL(os.fspath(LIB))
        P, D, I32, I64 = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        lib.cpu_create.restype = P
        lib.cpu_create.argtypes = [C.c_int, C.c_int, D, I32, C.POINTER(C.c_int8), C.POINTER(ModelC), C.c_int, I32, D]
        lib.cpu_destroy.argtypes = [P]
        lib.cpu_nnz_blocks.restype = C.c_int64
        lib.cpu_nnz_blocks.argtypes = [P]
        lib.cpu_n_colours.argtypes = [P]
        lib.cpu_set_state.argtypes = [P, D, D, D]
        lib.cpu_get_state.argtypes = [P, D]
        lib.cpu_shift_state.argtypes = [P]
        lib.cpu_reset_state.argtypes = [P]
        lib.cpu_assemble.argtypes = [P, C.c_double, C.c_double, C.c_int, C.c_int]
        lib.cpu_get_residual.argtypes = [P, D]
        lib.cpu_jacobian_csr.argtypes = [P, I64, I32, D]
        lib.cpu_block_csr.argtypes = [P, C.c_int, C.c_int, I64, I32, D]
        lib.cpu_set_amg.argtypes = [P, C.c_int, C.POINTER(CsrC), C.POINTER(CsrC), C.POINTER(CsrC), D, C.c_double]
        lib.cpu_set_chebyshev.argtypes = [P, C.c_int, D]
        lib.cpu_newton.argtypes = [P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                                   C.c_double, C.c_int, C.c_int, C.POINTER(ReportC)]
        lib.cpu_poisson_solve.argtypes = [P, C.c_double, C.c_int, C.POINTER(C.c_int)]
        lib.cpu_field_error.restype = C.c_double
        lib.cpu_field_error.argtypes = [P, C.c_int]
        lib.cpu_aggregate.argtypes = [C.c_int32, I64, I32, C.POINTER(C.c_uint8), I32]
        lib.cpu_counters.argtypes = [P, I64]
        lib.cpu_set_threads.argtypes = [C.c_int]
        _lib = lib
    return _lib


def _ts(ts):
    out = TermSumC()
    terms = [t for t in ts.terms if t[0] != 0.0]
    if len(terms) > MAXT:
        raise ValueError("coefficient function has too many terms for the CPU backend")
    out.n_terms = len(terms)
    for i, (c, p, q, r) in enumerate(terms):
        out.c[i], out.p[i], out.q[i], out.r[i] = c, p, q, r
    return out


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def chebyshev_weights(m, lam_min=0.5, lam_max=2.0):
    """Richardson weights whose residual polynomial is the degree-m Chebyshev polynomial on
    [lam_min, lam_max] (the point-block-diagonally scaled species block is mass-matrix-like)."""
    theta, delta = 0.5 * (lam_max + lam_min), 0.5 * (lam_max - lam_min)
    k = np.arange(1, m + 1)
    return 1.0 / (theta + delta * np.cos((2 * k - 1) * np.pi / (2 * m)))


class CpuProblem:
    """One LFA model with a Poisson row on one mesh (the streamer family), resident in the C library."""

    def __init__(self, model):
        """``model``: an :class:`oracle.forms.LFAModel` (mesh, coefficients, tags, Dirichlet data)."""
        self.lib = load()
        self.model = model
        mesh = model.mesh
        if not model.poisson or any(w is not None for w in model.drift_w) or any(e is not None for e in model.ext_source):
            raise ValueError("the CPU backend covers LFA models with a Poisson row, field drift and no Expression sources")
        m = ModelC()
        m.ns, m.n_reactions, m.axisymmetric = model.ns, len(model.reactions), int(model.axisymmetric)
        m.charge_over_eps = elementary_charge / epsilon_0
        for s in range(model.ns):
            m.eq_type[s], m.Z[s] = model.eq_type[s], model.Z[s]
            m.mu[s], m.D[s] = _ts(model.mu[s]), _ts(model.D[s])
        for j, (k, P, nu) in enumerate(model.reactions):
            m.k[j] = _ts(k)
            for s in range(model.ns):
                m.power[j][s], m.net[j][s] = int(P[s]), int(nu[s])
        if model.bc_type is not None:
            m.n_tags = len(model.bc_type)
            for t, row in enumerate(model.bc_type):
                for s in range(model.ns):
                    m.bc_neumann[t][s] = int(row[s] == "Neumann")
        self.nv, self.neq = mesh.nv, model.neq
        coords = np.ascontiguousarray(mesh.coords, dtype=np.float64)
        cells = np.ascontiguousarray(mesh.cells, dtype=np.int32)
        tags = None if model.facet_tags is None else np.ascontiguousarray(model.facet_tags, dtype=np.int8)
        ddofs = np.ascontiguousarray(model.dirichlet_dofs, dtype=np.int32)
        dvals = np.ascontiguousarray(model.dirichlet_vals, dtype=np.float64)
        self._h = self.lib.cpu_create(mesh.nv, mesh.nc, _dp(coords), _ip(cells),
                                      tags.ctypes.data_as(C.POINTER(C.c_int8)) if tags is not None else None,
                                      C.byref(m), ddofs.size, _ip(ddofs), _dp(dvals))
        self.levels = None
        self._keep = []

    def close(self):
        if self._h:
            self.lib.cpu_destroy(self._h)
            self._h = None

    # -- state ------------------------------------------------------------------------------
    def set_state(self, U=None, Uold=None, Uold1=None):
        v = [None if a is None else np.ascontiguousarray(a, dtype=np.float64).ravel() for a in (U, Uold, Uold1)]
        self.lib.cpu_set_state(self._h, *[None if a is None else _dp(a) for a in v])

    def get_state(self):
        out = np.empty(self.nv * self.neq)
        self.lib.cpu_get_state(self._h, _dp(out))
        return out.reshape(self.nv, self.neq)

    # -- assembly ---------------------------------------------------------------------------
    def residual(self, dt, dt_old):
        self.lib.cpu_assemble(self._h, dt, dt_old, 0, 0)
        F = np.empty(self.nv * self.neq)
        self.lib.cpu_get_residual(self._h, _dp(F))
        return F

    def residual_jacobian(self, dt, dt_old):
        self.lib.cpu_assemble(self._h, dt, dt_old, 1, 0)
        F = np.empty(self.nv * self.neq)
        self.lib.cpu_get_residual(self._h, _dp(F))
        nnz = self.lib.cpu_nnz_blocks(self._h) * self.neq ** 2
        N = self.nv * self.neq
        indptr, indices, values = np.empty(N + 1, np.int64), np.empty(nnz, np.int32), np.empty(nnz)
        self.lib.cpu_jacobian_csr(self._h, indptr.ctypes.data_as(C.POINTER(C.c_int64)), _ip(indices), _dp(values))
        return F, sp.csr_matrix((values, indices, indptr), shape=(N, N))

    def _potential_block(self):
        self.lib.cpu_assemble(self._h, 1.0, 1.0, 1, 1)
        nb = self.lib.cpu_nnz_blocks(self._h)
        indptr, indices, values = np.empty(self.nv + 1, np.int64), np.empty(nb, np.int32), np.empty(nb)
        ip = self.neq - 1
        self.lib.cpu_block_csr(self._h, ip, ip, indptr.ctypes.data_as(C.POINTER(C.c_int64)), _ip(indices), _dp(values))
        return sp.csr_matrix((values, indices, indptr), shape=(self.nv, self.nv))

    # -- preconditioner set-up (cold path, once per mesh) --------------------------------------
    def setup_multigrid(self, theta=0.08, omega_p=4.0 / 3.0, max_coarse=2000, omega=0.85, chebyshev=6):
        """Smoothed-aggregation hierarchy of the constant potential block (Vanek, Mandel,
        Brezina 1996): greedy aggregation on |a_ij| >= theta sqrt(a_ii a_jj), prolongator smoothed
        by one damped-Jacobi step, Galerkin coarse operators, dense inverse on the coarsest level."""
        A = self._potential_block()
        fixed = np.zeros(self.nv, dtype=bool)
        ip = self.neq - 1
        d = self.model.dirichlet_dofs
        fixed[d[d % self.neq == ip] // self.neq] = True
        levels, free = [], ~fixed
        while A.shape[0] > max_coarse and len(levels) < 11:
            idx = np.nonzero(free)[0]
            Af = A[idx][:, idx].tocsr()
            dg = np.abs(Af.diagonal())
            rows = np.repeat(np.arange(Af.shape[0]), np.diff(Af.indptr))
            strong = (np.abs(Af.data) >= theta * np.sqrt(dg[rows] * dg[Af.indices])).astype(np.uint8)
            strong[Af.indices == rows] = 0
            agg = np.empty(Af.shape[0], dtype=np.int32)
            indptr = np.ascontiguousarray(Af.indptr, dtype=np.int64)
            indices = np.ascontiguousarray(Af.indices, dtype=np.int32)
            nagg = self.lib.cpu_aggregate(Af.shape[0], indptr.ctypes.data_as(C.POINTER(C.c_int64)), _ip(indices),
                                          strong.ctypes.data_as(C.POINTER(C.c_uint8)), _ip(agg))
            if nagg >= 0.8 * idx.size:
                break
            T = sp.csr_matrix((np.ones(idx.size), (idx, agg)), shape=(A.shape[0], nagg))
            DinvA = sp.diags(1.0 / A.diagonal()) @ A
            rho = np.abs(DinvA).sum(axis=1).max()
            P = sp.diags(free.astype(np.float64)) @ (T - (omega_p / rho) * (DinvA @ T))
            P = P.tocsr()
            P.eliminate_zeros()
            levels.append((A, P))
            A = (P.T @ A @ P).tocsr()
            free = np.ones(A.shape[0], dtype=bool)
        levels.append((A, None))
        self.levels = [a.shape[0] for a, _ in levels]
        keep = []

        def pack(M):
            M = sp.csr_matrix(M)
            M.sort_indices()
            ip_, ii, iv = (np.ascontiguousarray(M.indptr, np.int64), np.ascontiguousarray(M.indices, np.int32),
                           np.ascontiguousarray(M.data, np.float64))
            keep.extend([ip_, ii, iv])
            return CsrC(M.shape[0], M.shape[1], ip_.ctypes.data_as(C.POINTER(C.c_int64)), _ip(ii), _dp(iv))
        n = len(levels)
        Ac = (CsrC * n)(*[pack(a) for a, _ in levels])
        Pc = (CsrC * max(n - 1, 1))(*[pack(p) for _, p in levels[:-1]])
        Rc = (CsrC * max(n - 1, 1))(*[pack(p.T) for _, p in levels[:-1]])
        inv = np.ascontiguousarray(np.linalg.inv(levels[-1][0].toarray()))
        self.lib.cpu_set_amg(self._h, n, Ac, Pc, Rc, _dp(inv), omega)
        w = np.ascontiguousarray(chebyshev_weights(chebyshev))
        self.lib.cpu_set_chebyshev(self._h, w.size, _dp(w))
        return self.levels

    # -- solves ----------------------------------------------------------------------------
    def poisson_solve(self, rtol=1e-12, max_it=20000):
        its = C.c_int()
        rc = self.lib.cpu_poisson_solve(self._h, rtol, max_it, C.byref(its))
        if rc != 0:
            raise RuntimeError("initial Poisson solve did not converge")
        return its.value

    def newton_solve(self, dt, dt_old, rtol=1e-4, max_it=20, atol=1e-10, stol=1e-16, ksp_rtol=1e-5,
                     ksp_restart=30, ksp_max_it=10000):
        rep = ReportC()
        rc = self.lib.cpu_newton(self._h, dt, dt_old, rtol, atol, stol, max_it, ksp_rtol, ksp_restart, ksp_max_it,
                                 C.byref(rep))
        self.last_report = rep
        if rc != 0:
            raise RuntimeError({1: "maximum number of Newton iterations reached", 2: "NaN or Inf in the residual",
                                3: "linear solve (GMRES) did not converge"}[rc])
        return rep.iterations, rep.linear_iterations

    def counters(self):
        out = (C.c_int64 * 2)()
        self.lib.cpu_counters(self._h, out)
        return {"spmv": int(out[0]), "vcycles": int(out[1])}


def streamer_problem(n, grading=4.0):
    """The bench workload (BASELINE configs[3]) on an n x n graded "right" mesh."""
    xs = graded_axis(ost.BOX, n, grading) if grading != 1.0 else None
    mesh = rectangle_right(0.0, 0.0, ost.BOX, ost.BOX, n, n, xs=xs)
    model = ost.build(mesh)
    return CpuProblem(model), mesh


def run_streamer(prob, mesh, steps, dt_init=5e-12, dt_max=5e-12, dt_min=1e-15, ttol=1e-3, rtol=1e-4, max_it=20):
    """ICs, initial Poisson solve and the time loop of fedm-streamer.py:169-225, :304-340.
    Returns (state, StepState, t, statistics)."""
    U = np.zeros((mesh.nv, 3))
    U[:, 0], U[:, 1] = ost.initial_log_densities(mesh.coords)
    prob.set_state(U, U, U)
    prob.setup_multigrid()
    pits = prob.poisson_solve()
    U = prob.get_state()
    prob.set_state(U, U, U)
    st = controller.StepState(dt_init, 1e30, n_error=2)
    t, done = 0.0, 0
    count = {"newton": 0, "linear": 0}
    U_old, U_old1 = U.copy(), U.copy()

    def solve(Uw, dt, dt_old):
        prob.set_state(Uw, U_old, U_old1)
        its, lits = prob.newton_solve(dt, dt_old, rtol, max_it)
        count["newton"] += its
        count["linear"] += lits
        Uw[:] = prob.get_state()

    t0 = time.perf_counter()
    while done < steps:
        U_old1[:] = U_old
        U_old[:] = U
        t = controller.adaptive_solve(solve, U, U_old, t, st, ttol, dt_min, error_component=1)
        st.dt_old = st.dt
        st.dt = controller.adaptive_timestep(st.dt, st.max_error, ttol, dt_min, dt_max)
        st.max_error[2] = st.max_error[1]
        st.max_error[1] = st.max_error[0]
        done += 1
    elapsed = time.perf_counter() - t0
    return U, st, t, dict(seconds=elapsed, newton=count["newton"], linear=count["linear"], poisson_iterations=pits)


def usable_cores():
    """CPUs this process can actually run on at once: its affinity mask, capped by the cgroup CPU
    quota (a GPU box hands a one-GPU job a share of the host -- 16 of 256 cores -- while the
    affinity mask still lists every core; 128 spinning OpenMP threads on 16 cores take 20x longer
    than 16 threads).  Without a readable quota: at most 16."""
    avail = len(os.sched_getaffinity(0))
    quota = None
    try:
        txt = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if txt[0] != "max":
            quota = int(txt[0]) / int(txt[1])
    except (OSError, ValueError, IndexError):
        try:
            q = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
            per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is None:
        return min(avail, 16), avail, "no cgroup quota readable: capped at 16"
    return max(1, min(avail, int(quota))), avail, f"cgroup quota {quota:g} CPUs"


def bench(n, grading, steps, threads):
    """bench.py's cpu_baseline record on the n x n graded tensor-product mesh."""
    xs = graded_axis(ost.BOX, n, grading) if grading != 1.0 else None
    mesh = rectangle_right(0.0, 0.0, ost.BOX, ost.BOX, n, n, xs=xs)
    return bench_mesh(mesh.coords, mesh.cells, steps, threads, f"{n}x{n} graded mesh")


def bench_mesh(coords, cells, steps, threads, label):
    """bench.py's cpu_baseline record: `steps` accepted BDF2 steps of the streamer case on the
    given triangle mesh of the box (the device run's own mesh), timed after set-up (mesh, pattern,
    multigrid hierarchy, initial Poisson solve are untimed on the device side too)."""
    from .mesh import Mesh
    lib = load()
    cores, avail, why = usable_cores()
    used = cores if threads <= 0 else min(threads, avail)
    lib.cpu_set_threads(used)
    mesh = Mesh(coords, cells)
    prob = CpuProblem(ost.build(mesh))
    n = label
    _, _, t, stats = run_streamer(prob, mesh, steps)
    ndof = mesh.nv * 3
    counters = prob.counters()
    out = {"value": ndof * steps / stats["seconds"], "unit": "DOF-updates/s", "cores": used, "kind": "port",
           "timesteps_per_sec": steps / stats["seconds"], "ms_per_step": 1e3 * stats["seconds"] / steps,
           "newton_iterations_per_step": stats["newton"] / steps, "gmres_iterations_per_step": stats["linear"] / steps,
           "multigrid_levels": prob.levels, "dofs": ndof,
           "sample": f"{steps} accepted BDF2 steps of the same streamer case on the same mesh ({n}; "
                     f"{ndof} DOFs), timed after set-up like the device run; oracle/cpu/fedm_cpu.c: C + OpenMP, "
                     f"coloured element loop -> block CSR -> Newton -> flexible GMRES(30) with the same field "
                     f"split (Chebyshev(6) species sweeps + smoothed-aggregation V(1,1)) -- 'CPU restatement, not "
                     f"FEniCS'; {used} OpenMP threads ({why}; affinity mask {avail} of {os.cpu_count()} host CPUs)",
           "spmv_count": counters["spmv"]}
    prob.close()
    return out
