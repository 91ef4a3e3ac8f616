"""Device context: the Python face of libfedm_hip.so.

One :class:`DeviceProblem` holds a mesh, a model descriptor and the state
vectors in HBM and offers what the reference reaches through DOLFIN:

* ``residual`` / ``jacobian``   <- ``Problem.F`` / ``Problem.J``  (fedm/functions.py:188-202)
* ``newton_solve``              <- ``PETScSNESSolver.solve``      (fedm/functions.py:1047)
* ``field_error``               <- the two ``df.norm`` calls      (fedm/functions.py:1062-1064)

Numerical failure (non-convergence, NaN) raises ``RuntimeError`` exactly where
DOLFIN's ``error_on_nonconvergence`` would, so ``adaptive_solver``'s catch-all
retry (fedm/functions.py:1080) keeps working.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, quadrature
from .physical_constants import elementary_charge, epsilon_0
from .termsum import TermSum


@dataclass
class Reaction:
    k: TermSum                 # rate coefficient as a function of |E|
    power: Sequence[int]       # power-matrix row over the solved species
    net: Sequence[int]         # gain - loss row


@dataclass
class Model:
    """What fedm.functions' weak-form builders describe (see functions.py)."""
    n_species: int
    poisson: bool
    eq_type: Sequence[str]
    Z: Sequence[float]
    mu: Sequence[TermSum] = ()
    D: Sequence[TermSum] = ()
    drift_w: Sequence[Optional[Tuple[float, float]]] = ()
    reactions: Sequence[Reaction] = ()
    bc_kind: Sequence[Sequence[str]] = ()      # [tag-1][species]
    quadrature_degree: int = 2
    ext_source_degree: Sequence[int] = ()      # 0 = none, k = Expression(degree=k) source
    axisymmetric: bool = True
    log_representation: bool = True            # False: the unknowns are the densities (functions.py:350-368)

    @property
    def n_eq(self):
        return self.n_species + (1 if self.poisson else 0)

    def to_c(self):
        md = _lib.ModelDesc()
        ns = self.n_species
        md.linear_representation = 0 if self.log_representation else 1
        if not 1 <= ns <= _lib.MAX_SPECIES:
            raise ValueError(f"n_species must be 1..{_lib.MAX_SPECIES}")
        md.n_species, md.poisson, md.axisymmetric = ns, int(self.poisson), int(self.axisymmetric)
        mu = list(self.mu) or [TermSum.const(0.0)] * ns
        D = list(self.D) or [TermSum.const(0.0)] * ns
        w = list(self.drift_w) or [None] * ns
        for s in range(ns):
            md.eq_type[s] = _lib.EQ_TYPES[self.eq_type[s]]
            md.Z[s] = float(self.Z[s])
            TermSum.coerce(mu[s]).fill(md.mu[s])
            TermSum.coerce(D[s]).fill(md.D[s])
            if w[s] is not None:
                md.has_drift_w[s] = 1
                md.drift_w[s][0], md.drift_w[s][1] = float(w[s][0]), float(w[s][1])
        if len(self.reactions) > _lib.MAX_REACTIONS:
            raise ValueError(f"at most {_lib.MAX_REACTIONS} reactions")
        md.n_reactions = len(self.reactions)
        for j, rc in enumerate(self.reactions):
            TermSum.coerce(rc.k).fill(md.k[j])
            for s in range(ns):
                md.power[j][s] = int(rc.power[s])
                md.net[j][s] = int(rc.net[s])
        md.charge_over_eps = elementary_charge / epsilon_0
        md.n_tags = len(self.bc_kind)
        if md.n_tags > _lib.MAX_TAGS:
            raise ValueError(f"at most {_lib.MAX_TAGS} boundary tags")
        for t, row in enumerate(self.bc_kind):
            for s in range(ns):
                md.bc_kind[t][s] = _lib.BC_KINDS[row[s]]
        xq, wq = quadrature.triangle(self.quadrature_degree)
        if len(wq) > _lib.MAX_QP:
            raise ValueError("quadrature rule too large")
        md.n_qp = len(wq)
        for q in range(len(wq)):
            md.qp_x[q], md.qp_y[q], md.qp_w[q] = xq[q, 0], xq[q, 1], wq[q]
        tq, wt = quadrature.interval(self.quadrature_degree)
        md.n_fqp = len(wt)
        for q in range(len(wt)):
            md.fqp_t[q], md.fqp_w[q] = tq[q], wt[q]
        ext = list(self.ext_source_degree) or [0] * ns
        degs = {k for k in ext if k}
        if len(degs) > 1:
            raise ValueError("all Expression sources must share one degree")
        for s in range(ns):
            md.ext_nodes[s] = (ext[s] + 1) * (ext[s] + 2) // 2 if ext[s] else 0
        if degs:
            B, _ = quadrature.lagrange_interpolation_matrix(degs.pop(), xq)
            for q in range(B.shape[0]):
                for m in range(B.shape[1]):
                    md.ext_B[q][m] = B[q, m]
        return md


@dataclass
class GdModel:
    """LMEA model family (examples/glow_discharge/fedm-gd.py): energy equation + particle
    balances with nodal, semi-implicit coefficients + Poisson.  Species 0 is the gas."""
    n_species: int
    N0: float
    eq_type: Sequence[str]
    grad_diffusion: Sequence[bool]
    is_ion: Sequence[bool]
    sign: Sequence[float]
    vth: Sequence[float]
    electron_mass: float
    power: Sequence[Sequence[int]]
    net: Sequence[Sequence[int]]
    energy_loss: Sequence[float]
    ref: Sequence[Sequence[float]]             # [tag-1][species]
    gamma: Sequence[float]                     # [tag-1]
    we_secondary: float
    quadrature_degree: int = 4
    axisymmetric: bool = True
    poisson: bool = True
    # Energy_Source_term's Ei and mean_energy (fedm/functions.py:855, 906-909) for the sentinel losses 7.77e77 / 9.99e99
    # left in `energy_loss`: None (no sentinel may be present) or "unknown_ratio" = u[0] / u[n - 1] (fedm-gd.py:358)
    energy_Ei: float = 0.0
    mean_energy_form: Optional[str] = None

    @property
    def n_eq(self):
        return self.n_species + 1

    @property
    def n_reactions(self):
        return len(self.power)

    @property
    def n_fields(self):
        return 4 * self.n_species + 2 * self.n_reactions + 3

    def to_c(self):
        md = _lib.GdDesc()
        ns, nr = self.n_species, self.n_reactions
        if ns > _lib.GD_MAX_SPECIES - 1 or nr > _lib.GD_MAX_REACTIONS or len(self.ref) > _lib.MAX_TAGS:
            raise ValueError("LMEA model too large for the device descriptor")
        md.n_species, md.n_reactions, md.n_tags = ns, nr, len(self.ref)
        md.axisymmetric, md.N0 = int(self.axisymmetric), float(self.N0)
        md.charge_over_eps = elementary_charge / epsilon_0
        for i in range(ns):
            md.eq_type[i] = _lib.EQ_TYPES[self.eq_type[i]]
            md.grad_diffusion[i] = int(bool(self.grad_diffusion[i]))
            md.is_ion[i] = int(bool(self.is_ion[i]))
            md.sign[i] = float(self.sign[i])
            md.vth[i] = float(self.vth[i])
        md.vth_e_coef = 16.0 * elementary_charge / (3.0 * np.pi * self.electron_mass)
        for j in range(nr):
            md.energy_loss[j] = float(self.energy_loss[j])
            for i in range(ns):
                md.power[j][i] = int(self.power[j][i])
                md.net[j][i] = int(self.net[j][i])
        for t in range(len(self.ref)):
            md.gamma[t] = float(self.gamma[t])
            for i in range(ns):
                md.ref[t][i] = float(self.ref[t][i])
        md.we_secondary = float(self.we_secondary)
        sentinel = any(7e77 < v < 8e77 or 9e99 < v < 1e100 for v in self.energy_loss)
        if sentinel and self.mean_energy_form is None:
            raise ValueError("energy losses that depend on the mean energy need mean_energy_form")
        md.energy_Ei = float(self.energy_Ei)
        md.mean_energy_form = _lib.GD_ME_FORMS[self.mean_energy_form]
        xq, wq = quadrature.triangle(self.quadrature_degree)
        md.n_qp = len(wq)
        for q in range(len(wq)):
            md.qp_x[q], md.qp_y[q], md.qp_w[q] = xq[q, 0], xq[q, 1], wq[q]
        tq, wt = quadrature.interval(self.quadrature_degree)
        md.n_fqp = len(wt)
        for q in range(len(wt)):
            md.fqp_t[q], md.fqp_w[q] = tq[q], wt[q]
        return md


@dataclass
class NewtonReport:
    iterations: int
    converged: bool
    linear_iterations: int
    fnorm0: float
    fnorm: float


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _part1by1(v):
    v = v.astype(np.uint64) & np.uint64(0xFFFFFFFF)
    for shift, mask in ((16, 0x0000FFFF0000FFFF), (8, 0x00FF00FF00FF00FF),
                        (4, 0x0F0F0F0F0F0F0F0F), (2, 0x3333333333333333),
                        (1, 0x5555555555555555)):
        v = (v | (v << np.uint64(shift))) & np.uint64(mask)
    return v


def z_curve_order(coords):
    """Vertex order along a Z-curve over the rank-quantised coordinates: on a tensor-product mesh,
    whatever its grading, 64 consecutive vertices form an 8x8 block."""
    ranks = []
    for d in range(2):
        _, inv = np.unique(coords[:, d], return_inverse=True)
        ranks.append(inv.astype(np.uint64))
    key = _part1by1(ranks[0]) | (_part1by1(ranks[1]) << np.uint64(1))
    return np.argsort(key, kind="stable")


def _vertex_spacing(coords, cells):
    """Mean |dx|, |dy| of the edges at every vertex: the metric in which a patch should be round."""
    cells = np.asarray(cells)
    n = coords.shape[0]
    e = np.concatenate([cells[:, [0, 1]], cells[:, [1, 2]], cells[:, [2, 0]]])
    d = np.abs(coords[e[:, 0]] - coords[e[:, 1]])
    ends = np.concatenate([e[:, 0], e[:, 1]])
    cnt = np.bincount(ends, minlength=n).astype(np.float64)
    acc = np.stack([np.bincount(ends, weights=np.concatenate([d[:, k], d[:, k]]), minlength=n) for k in (0, 1)], axis=1)
    h = acc / np.maximum(cnt, 1.0)[:, None]
    h[h <= 0.0] = h[h > 0.0].mean() if np.any(h > 0.0) else 1.0
    return h


def bisection_order(coords, spacing=None, leaf=64):
    """Vertex order by recursive median bisection (a k-d tree whose leaves are the 64-vertex
    slices): each split halves the vertex set across its longer side, measured in local mesh
    spacings, and puts a multiple of 64 vertices to the left, so that every slice is one leaf -- a
    compact patch on a locally refined unstructured mesh, where the quantised Z-curve tears patches
    apart.  Inside a leaf the bisection goes on down to single vertices (a Z-like local order)."""
    n = coords.shape[0]
    h = np.ones_like(coords) if spacing is None else spacing
    order = np.arange(n)
    stack = [(0, n)]
    while stack:
        a, b = stack.pop()
        m = b - a
        if m <= leaf:
            continue
        idx = order[a:b]
        p = coords[idx]
        ext = (p.max(axis=0) - p.min(axis=0)) / h[idx].mean(axis=0)
        d = 0 if ext[0] >= ext[1] else 1
        n_left = ((m + leaf - 1) // leaf + 1) // 2 * leaf
        order[a:b] = idx[np.argpartition(p[:, d], n_left - 1)]
        stack.append((a, a + n_left))
        stack.append((a + n_left, b))
    n_leaves = (n + leaf - 1) // leaf
    o = np.concatenate([order, np.full(n_leaves * leaf - n, -1, dtype=order.dtype)])
    g = leaf
    while g > 1:                                   # all leaves at once, one level per pass
        grp = o.reshape(-1, g)
        valid = grp >= 0
        safe = np.maximum(grp, 0)
        P, H = coords[safe], h[safe]
        lo = np.where(valid[..., None], P, np.inf).min(axis=1)
        hi = np.where(valid[..., None], P, -np.inf).max(axis=1)
        hm = np.where(valid[..., None], H, 0.0).sum(axis=1) / np.maximum(valid.sum(axis=1), 1)[:, None]
        with np.errstate(invalid="ignore", divide="ignore"):
            ext = (hi - lo) / hm
        d = np.where(np.nan_to_num(ext[:, 0], nan=0.0) >= np.nan_to_num(ext[:, 1], nan=0.0), 0, 1)
        key = np.where(valid, np.take_along_axis(P, d[:, None, None], axis=2)[..., 0], np.inf)
        o = np.take_along_axis(grp, np.argsort(key, axis=1, kind="stable"), axis=1).reshape(-1)
        g //= 2
    return o[o >= 0]


def cell_visits(order, cells, n_vertices, leaf=64):
    """Cell evaluations of the patch assembly under a vertex order: a cell is evaluated once by
    every slice that owns one of its vertices."""
    inv = np.empty(n_vertices, dtype=np.int64)
    inv[order] = np.arange(order.size)
    s = np.sort(inv[cells] // leaf, axis=1)
    return int(cells.shape[0] + (s[:, 1] != s[:, 0]).sum() + (s[:, 2] != s[:, 1]).sum())


def locality_order(coords, cells=None):
    """Internal vertex numbering of the device path (returns ``order``: new -> old).

    64 consecutive vertices are one slice = one assembly patch = one wavefront of matrix rows, so
    the order should make them compact: few halo vertices and redundant cell evaluations per patch,
    the neighbours of a matrix slice within a few cache lines, multigrid aggregates contiguous.
    Two candidates -- the Z-curve over rank-quantised coordinates (ideal on tensor-product meshes,
    whatever their grading) and recursive bisection in the metric of the local mesh spacing (for
    unstructured, locally refined meshes) -- and the one with fewer redundant cell evaluations is
    taken; without ``cells`` the Z-curve.  Deterministic: every rank computes the same order."""
    z = z_curve_order(coords)
    if cells is None or len(cells) == 0:
        return z
    cells = np.asarray(cells)
    kd = bisection_order(coords, _vertex_spacing(coords, cells))
    nv = coords.shape[0]
    return z if cell_visits(z, cells, nv) <= cell_visits(kd, cells, nv) else kd


def pattern_stats(coords, cells, reorder=True):
    """Host-side preprocessing alone (``fedm_pattern_stats``; runs without a GPU): slices, patch
    sizes and the LDS-accumulator clashes of the patch cell order, for the mesh as the device would
    number it."""
    lib = _lib.load()
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    order = locality_order(coords, cells) if reorder else np.arange(coords.shape[0])
    inv = np.empty(coords.shape[0], dtype=np.int64)
    inv[order] = np.arange(coords.shape[0])
    cdev = np.ascontiguousarray(coords[order])
    kdev = np.ascontiguousarray(inv[cells], dtype=np.int32)
    mesh = _lib.MeshDesc()
    mesh.n_vertices, mesh.n_cells = coords.shape[0], cells.shape[0]
    mesh.coords = _dp(cdev)
    mesh.cells = kdev.ctypes.data_as(C.POINTER(C.c_int32))
    out = (C.c_int64 * 12)()
    rc = lib.fedm_pattern_stats(C.byref(mesh), out)
    if rc != 0:
        raise RuntimeError(f"fedm_pattern_stats failed ({rc}): {_lib.last_error()}")
    keys = ("n_slices", "max_patch_cells", "max_patch_width", "max_patch_verts", "cell_visits",
            "owned_pairs", "bank_clashes", "nnz_blocks", "stored_blocks", "halo_vertices", "n_colours",
            "emission_blocks")
    return dict(zip(keys, (int(v) for v in out)))


def fieldsplit_tiles_stats(coords, cells, slices_per_tile=8, layers=3, reorder=True):
    """The tile tables of the species sweeps (csrc/fs_tiles.hip) built on the host alone
    (``fedm_fieldsplit_tiles_stats``; runs without a GPU), for the mesh as the device would number it, with the
    library's self-check of them (``violations`` must be 0)."""
    lib = _lib.load()
    coords = np.ascontiguousarray(coords, dtype=np.float64)
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    order = locality_order(coords, cells) if reorder else np.arange(coords.shape[0])
    inv = np.empty(coords.shape[0], dtype=np.int64)
    inv[order] = np.arange(coords.shape[0])
    cdev = np.ascontiguousarray(coords[order])
    kdev = np.ascontiguousarray(inv[cells], dtype=np.int32)
    mesh = _lib.MeshDesc()
    mesh.n_vertices, mesh.n_cells = coords.shape[0], cells.shape[0]
    mesh.coords = _dp(cdev)
    mesh.cells = kdev.ctypes.data_as(C.POINTER(C.c_int32))
    out = (C.c_int64 * 10)()
    rc = lib.fedm_fieldsplit_tiles_stats(C.byref(mesh), int(slices_per_tile), int(layers), out)
    if rc != 0:
        raise RuntimeError(f"fedm_fieldsplit_tiles_stats failed ({rc}): {_lib.last_error()}")
    keys = ("n_tiles", "row_width", "max_vertices", "max_rows", "total_rows", "total_vertices", "bytes", "violations",
            "lds_bytes_species_kernel", "lds_bytes_multigrid_kernel")
    return dict(zip(keys, (int(v) for v in out)))


class DeviceProblem:
    """Mesh + model + state resident on one MI355X."""

    def __init__(self, coords, cells, model: Model, facet_tags=None,
                 dirichlet_dofs=(), dirichlet_vals=(), device=0, reorder=True, n_owned=None,
                 identity_vertices=None, halo_depth=1):
        self.lib = _lib.load()
        self.model = model
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.nv, self.nc = self.coords.shape[0], self.cells.shape[0]
        if self.coords.ndim != 2 or self.coords.shape[1] != 2 or self.cells.ndim != 2 or self.cells.shape[1] != 3:
            raise ValueError("DeviceProblem: coords must be (n_vertices, 2) and cells (n_cells, 3)")
        if self.nc and (self.cells.min() < 0 or self.cells.max() >= self.nv):
            raise ValueError("DeviceProblem: cell vertex index out of range")
        self.n_eq = model.n_eq
        self.n = self.nv * self.n_eq
        self._tags = None if facet_tags is None else np.ascontiguousarray(facet_tags, dtype=np.int8)
        # internal vertex numbering (the caller keeps seeing its own numbering)
        # multi-GPU: vertices [0, n_owned) are owned, the rest are ghosts (kept in place)
        self.n_owned = self.nv if n_owned is None else int(n_owned)
        self._order = np.arange(self.nv)
        if reorder:
            # (cells among the owned vertices: across GPUs the ghosts keep their place at the end)
            own_cells = self.cells[(self.cells < self.n_owned).all(axis=1)]
            self._order[:self.n_owned] = locality_order(self.coords[:self.n_owned], own_cells)
        self._inv = np.empty(self.nv, dtype=np.int64)
        self._inv[self._order] = np.arange(self.nv)
        neq = self.n_eq
        self._dof_new_of_old = (self._inv[:, None] * neq + np.arange(neq)[None, :]).ravel()
        ddofs = np.asarray(dirichlet_dofs, dtype=np.int64)
        self._ddofs = np.ascontiguousarray(self._dof_new_of_old[ddofs] if ddofs.size else ddofs,
                                           dtype=np.int32)
        self._dvals = np.ascontiguousarray(dirichlet_vals, dtype=np.float64)
        self._coords_dev = np.ascontiguousarray(self.coords[self._order])
        self._cells_dev = np.ascontiguousarray(self._inv[self.cells], dtype=np.int32)
        if isinstance(model, GdModel) and reorder and os.environ.get("FEDM_GD_CELL_ORDER", "1") != "0":
            # LMEA: the element kernels take 64 consecutive cells a workgroup and the gather kernel reads, for
            # consecutive matrix rows, the blocks of the cells around them -- cells in the order of their vertices
            # (nothing the caller passes or gets back is indexed by cell besides the facet tags)
            by_vertex = np.argsort(self._cells_dev.min(axis=1), kind="stable")
            self._cells_dev = np.ascontiguousarray(self._cells_dev[by_vertex])
            if self._tags is not None:
                self._tags = np.ascontiguousarray(self._tags.reshape(self.nc, 3)[by_vertex])
        md = model.to_c()
        mesh = _lib.MeshDesc()
        mesh.n_vertices, mesh.n_cells = self.nv, self.nc
        mesh.coords = _dp(self._coords_dev)
        mesh.cells = self._cells_dev.ctypes.data_as(C.POINTER(C.c_int32))
        mesh.facet_tags = (self._tags.ctypes.data_as(C.POINTER(C.c_int8))
                           if self._tags is not None else None)
        mesh.n_dirichlet = self._ddofs.size
        mesh.dirichlet_dofs = self._ddofs.ctypes.data_as(C.POINTER(C.c_int32))
        mesh.dirichlet_vals = _dp(self._dvals)
        mesh.n_owned_vertices = self.n_owned
        # deep halos (partition.local_mesh(depth > 1)): the ghost vertices with identity rows (ghosts keep
        # their place in the internal numbering, so the caller's ids are the device's)
        self.halo_depth = int(halo_depth)
        if identity_vertices is not None and self.halo_depth > 1:
            self._identity = np.ascontiguousarray(identity_vertices, dtype=np.int32)
            if self._identity.size and (self._identity.min() < self.n_owned or self._identity.max() >= self.nv):
                raise ValueError("DeviceProblem: identity_vertices must be ghost vertices")
            mesh.n_identity_vertices = self._identity.size
            mesh.identity_vertices = self._identity.ctypes.data_as(C.POINTER(C.c_int32))
            mesh.halo_depth = self.halo_depth
        handle = C.c_void_p()
        create = self.lib.fedm_ctx_create_gd if isinstance(model, GdModel) else self.lib.fedm_ctx_create
        rc = create(C.byref(mesh), C.byref(md), int(device), C.byref(handle))
        if rc != 0:
            raise RuntimeError(f"fedm_ctx_create failed ({rc}): {_lib.last_error()}")
        self._h = handle

    # -- lifetime ----------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.fedm_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError(f"{what} failed ({rc}): {_lib.last_error()}")
        if rc > 0:
            raise RuntimeError(f"{what}: {_lib.DIVERGED.get(rc, rc)}")

    # -- state --------------------------------------------------------------
    def _vec(self, a):
        """caller's dof order -> device order"""
        if a is None:
            return None
        a = np.asarray(a, dtype=np.float64).reshape(-1)
        if a.size != self.n:
            raise ValueError(f"state vector must have {self.n} entries, got {a.size}")
        return np.ascontiguousarray(a.reshape(self.nv, self.n_eq)[self._order]).reshape(-1)

    def _back(self, a):
        """device order -> caller's dof order"""
        return np.ascontiguousarray(a.reshape(self.nv, self.n_eq)[self._inv]).reshape(-1)

    def set_state(self, u_new=None, u_old=None, u_old1=None):
        vs = [self._vec(v) for v in (u_new, u_old, u_old1)]
        self._check(self.lib.fedm_set_state(self._h, *[_dp(v) if v is not None else None for v in vs]),
                    "fedm_set_state")

    def get_state(self):
        out = np.empty(self.n)
        self._check(self.lib.fedm_get_state(self._h, _dp(out)), "fedm_get_state")
        return self._back(out).reshape(self.nv, self.n_eq)

    def shift_state(self):
        self._check(self.lib.fedm_shift_state(self._h), "fedm_shift_state")

    def reset_state(self):
        self._check(self.lib.fedm_reset_state(self._h), "fedm_reset_state")

    def snapshot_state(self):
        """u, u_old, u_old1 into a device-side copy (fedm_state_snapshot)."""
        self._check(self.lib.fedm_state_snapshot(self._h), "fedm_state_snapshot")

    def restore_state(self):
        self._check(self.lib.fedm_state_restore(self._h), "fedm_state_restore")

    def set_step(self, dt, dt_old):
        self._check(self.lib.fedm_set_step(self._h, float(dt), float(dt_old)), "fedm_set_step")

    def set_dirichlet_values(self, vals):
        v = np.ascontiguousarray(vals, dtype=np.float64)
        if v.size != self._ddofs.size:
            raise ValueError("wrong number of Dirichlet values")
        self._check(self.lib.fedm_set_dirichlet_values(self._h, _dp(v)), "fedm_set_dirichlet_values")

    def set_gd_fields(self, fields):
        """Nodal coefficient fields of the LMEA family, (n_fields, n_vertices) in the order
        of FEDM_GD_N_FIELDS (include/fedm_hip.h), caller's vertex numbering."""
        f = np.asarray(fields, dtype=np.float64)
        if f.shape != (self.model.n_fields, self.nv):
            raise ValueError(f"expected fields of shape {(self.model.n_fields, self.nv)}")
        f = np.ascontiguousarray(f[:, self._order])
        self._check(self.lib.fedm_gd_set_fields(self._h, _dp(f)), "fedm_gd_set_fields")

    def get_gd_fields(self):
        out = np.empty((self.model.n_fields, self.nv))
        self._check(self.lib.fedm_gd_get_fields(self._h, _dp(out)), "fedm_gd_get_fields")
        return np.ascontiguousarray(out[:, self._inv])

    def gd_prep_setup(self, tables, programs):
        """Install the on-device per-step refresh of the LMEA coefficient fields.
        ``tables``: list of (x, y) arrays; ``programs``: one dict per field row with keys
        kind ('keep' | 'table' | 'scaled_row' | 'me_old' | 'me' | 'ue_old') and, as needed,
        table, arg ('energy' | 'redfield'), scale, src_row."""
        import scipy.sparse as sp
        from . import amg
        x = self._coords_dev[self._cells_dev]
        d1, d2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
        det = np.abs(d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0])
        vals = det[:, None, None] * ((np.ones((3, 3)) + np.eye(3)) / 24.0)[None]
        c = self._cells_dev.astype(np.int64)
        rows = np.broadcast_to(c[:, :, None], vals.shape).ravel()
        cols = np.broadcast_to(c[:, None, :], vals.shape).ravel()
        M = sp.coo_matrix((vals.ravel(), (rows, cols)), shape=(self.nv, self.nv)).tocsr()
        keep = []
        mass = amg._csr_struct(M, keep)
        ptr = np.zeros(len(tables) + 1, dtype=np.int32)
        ptr[1:] = np.cumsum([len(t[0]) for t in tables])
        tx = np.ascontiguousarray(np.concatenate([np.asarray(t[0], float) for t in tables]) if tables else np.zeros(1))
        ty = np.ascontiguousarray(np.concatenate([np.asarray(t[1], float) for t in tables]) if tables else np.zeros(1))
        if len(programs) != self.model.n_fields:
            raise ValueError("one program per field row")
        progs = (_lib.GdFieldProg * len(programs))()
        for r, p in enumerate(programs):
            progs[r].kind = _lib.GDP[p["kind"]]
            progs[r].table = int(p.get("table", 0))
            progs[r].arg = _lib.GDP_ARG[p.get("arg", "energy")]
            progs[r].src_row = int(p.get("src_row", 0))
            progs[r].scale = float(p.get("scale", 1.0))
        self._check(self.lib.fedm_gd_prep_setup(self._h, C.byref(mass), len(tables),
                                                ptr.ctypes.data_as(C.POINTER(C.c_int32)), _dp(tx), _dp(ty),
                                                progs), "fedm_gd_prep_setup")

    def gd_prep_step(self):
        self._check(self.lib.fedm_gd_prep_step(self._h), "fedm_gd_prep_step")

    def gd_update_mean_energy(self):
        self._check(self.lib.fedm_gd_update_mean_energy(self._h), "fedm_gd_update_mean_energy")

    def get_state_old(self):
        out = np.empty(self.n)
        self._check(self.lib.fedm_get_state_old(self._h, _dp(out)), "fedm_get_state_old")
        return self._back(out).reshape(self.nv, self.n_eq)

    def set_ext_source_program(self, species, ops, consts, n_params):
        """Expression source evaluated on the device: the postfix program of
        :func:`fedm_amd.forms.expression_program` (``fedm_ext_source_program``)."""
        ops = np.ascontiguousarray(ops, dtype=np.int32).reshape(-1, 2)
        consts = np.ascontiguousarray(consts, dtype=np.float64)
        self._check(self.lib.fedm_ext_source_program(
            self._h, int(species), int(ops.shape[0]), ops.ctypes.data_as(C.POINTER(C.c_int32)),
            int(consts.size), _dp(consts if consts.size else np.zeros(1)), int(n_params)), "fedm_ext_source_program")

    def eval_ext_source(self, species, params):
        """Fill the species' source table with the program's values for these parameter values."""
        p = np.ascontiguousarray(params if len(params) else [0.0], dtype=np.float64)
        self._check(self.lib.fedm_ext_source_eval(self._h, int(species), _dp(p)), "fedm_ext_source_eval")

    def set_ext_source(self, species, nodal):
        v = np.ascontiguousarray(nodal, dtype=np.float64)
        self._check(self.lib.fedm_set_ext_source(self._h, int(species), _dp(v)), "fedm_set_ext_source")

    # -- Problem.F / Problem.J ------------------------------------------------
    def residual(self, download=True):
        F = np.empty(self.n) if download else None
        fn = C.c_double()
        self._check(self.lib.fedm_residual(self._h, _dp(F) if download else None, C.byref(fn)),
                    "fedm_residual")
        return (self._back(F), fn.value) if download else fn.value

    def jacobian(self):
        self._check(self.lib.fedm_jacobian(self._h), "fedm_jacobian")

    def jacobian_csr(self):
        import scipy.sparse as sp
        nnz = self.lib.fedm_jacobian_nnz(self._h)
        indptr = np.empty(self.n + 1, dtype=np.int64)
        indices = np.empty(nnz, dtype=np.int32)
        values = np.empty(nnz)
        self._check(self.lib.fedm_jacobian_csr(
            self._h, indptr.ctypes.data_as(C.POINTER(C.c_int64)),
            indices.ctypes.data_as(C.POINTER(C.c_int32)), _dp(values)), "fedm_jacobian_csr")
        J = sp.csr_matrix((values, indices, indptr), shape=(self.n, self.n))
        m = self._dof_new_of_old
        return J[m][:, m].tocsr()

    def spmv(self, x):
        x = self._vec(x)
        y = np.empty(self.n)
        self._check(self.lib.fedm_spmv(self._h, _dp(x), _dp(y)), "fedm_spmv")
        return self._back(y)

    def comm_roundtrip(self, vec, red=()):
        """One halo exchange of a DOF vector and one all-reduce through the context's transport
        (``fedm_debug_comm_roundtrip``): returns (vector with the ghost entries received, sums)."""
        x = self._vec(vec).copy()
        r = np.ascontiguousarray(red if len(red) else [0.0], dtype=np.float64).copy()
        self._check(self.lib.fedm_debug_comm_roundtrip(self._h, _dp(x), _dp(r), int(len(red))),
                    "fedm_debug_comm_roundtrip")
        return self._back(x), r[:len(red)]

    # -- multi-GPU transport ---------------------------------------------------------
    def _plan_arrays(self, lm):
        nb = np.ascontiguousarray(lm.neighbours, dtype=np.int32)
        sp_ = np.ascontiguousarray(lm.send_ptr, dtype=np.int32)
        si = np.ascontiguousarray(self._inv[lm.send_idx], dtype=np.int32)   # device numbering
        rp = np.ascontiguousarray(lm.recv_ptr, dtype=np.int32)
        self._plan_keep = (nb, sp_, si, rp)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        return len(nb), ip(nb), ip(sp_), ip(si), ip(rp)

    def init_comm_rccl(self, lm, unique_id: bytes, rank, n_ranks):
        """RCCL transport (ncclSend/ncclRecv halo groups + ncclAllReduce on the library's
        stream).  ``unique_id``: 128 bytes from :func:`rccl_unique_id` on rank 0."""
        uid = C.create_string_buffer(unique_id, 128)
        self._check(self.lib.fedm_comm_init_rccl(self._h, *self._plan_arrays(lm), uid,
                                                 int(rank), int(n_ranks)), "fedm_comm_init_rccl")

    def init_comm_torch(self, lm, group=None):
        """Host-staged transport through torch.distributed (gloo): the library copies the
        packed halo / the reduction scalars to pinned host memory and calls back here.
        Same algorithm as the RCCL path, used to test it where ranks share one GPU."""
        import torch
        import torch.distributed as dist
        n_send, n_ghost = int(lm.send_ptr[-1]), int(lm.recv_ptr[-1])

        def allreduce(buf, n, _user):
            a = np.ctypeslib.as_array(buf, shape=(n,))
            t = torch.from_numpy(a)
            dist.all_reduce(t, group=group)

        def exchange(send, recv, width, _user):
            s = np.ctypeslib.as_array(send, shape=(max(n_send, 1), width))
            r = np.ctypeslib.as_array(recv, shape=(max(n_ghost, 1), width))
            reqs, bufs = [], []
            for k, q in enumerate(lm.neighbours):
                out = torch.from_numpy(np.ascontiguousarray(s[lm.send_ptr[k]:lm.send_ptr[k + 1]]))
                reqs.append(dist.isend(out, int(q), group=group))
                buf = torch.empty((int(lm.recv_ptr[k + 1] - lm.recv_ptr[k]), width), dtype=torch.float64)
                bufs.append(buf)
                reqs.append(dist.irecv(buf, int(q), group=group))
            for q in reqs:
                q.wait()
            for k, buf in enumerate(bufs):
                r[lm.recv_ptr[k]:lm.recv_ptr[k + 1]] = buf.numpy()

        self._cb_keep = (_lib.ALLREDUCE_FN(allreduce), _lib.EXCHANGE_FN(exchange))
        self._check(self.lib.fedm_comm_init_callbacks(
            self._h, *self._plan_arrays(lm), self._cb_keep[0], self._cb_keep[1], None,
            dist.get_rank(group), dist.get_world_size(group)), "fedm_comm_init_callbacks")

    def sync_ghosts(self):
        self._check(self.lib.fedm_sync_ghosts(self._h), "fedm_sync_ghosts")

    def time_comm(self, kind, repeats=50):
        """ms per halo exchange (kind 0: block vector, 1: scalar) or all-reduce of 32 doubles (2)."""
        ms = C.c_double()
        self._check(self.lib.fedm_time_comm(self._h, int(kind), int(repeats), C.byref(ms)), "fedm_time_comm")
        return ms.value

    def comm_stats(self):
        """Transport kind and counters of the multi-GPU plumbing (``fedm_comm_stats``)."""
        out = (C.c_int64 * 10)()
        self._check(self.lib.fedm_comm_stats(self._h, out), "fedm_comm_stats")
        kind = {0: "none", 1: "host-staged callbacks", 2: "rccl"}[int(out[0])]
        return dict(transport=kind, ranks=int(out[1]), halo_exchanges=int(out[2]), allreduces=int(out[3]),
                    failed=bool(out[4]), neighbours=int(out[5]), interior_patches=int(out[6]),
                    boundary_patches=int(out[7]), halo_bytes=int(out[8]), allreduce_bytes=int(out[9]))

    # -- linear-solver set-up ---------------------------------------------------
    def block_csr(self, cr, cc):
        import scipy.sparse as sp
        nnz = self.lib.fedm_block_nnz(self._h)
        indptr = np.empty(self.nv + 1, dtype=np.int64)
        indices = np.empty(nnz, dtype=np.int32)
        values = np.empty(nnz)
        self._check(self.lib.fedm_block_csr(
            self._h, int(cr), int(cc), indptr.ctypes.data_as(C.POINTER(C.c_int64)),
            indices.ctypes.data_as(C.POINTER(C.c_int32)), _dp(values)), "fedm_block_csr")
        return sp.csr_matrix((values, indices, indptr), shape=(self.nv, self.nv))

    def setup_multigrid(self, theta=0.08, nu=2, omega=0.67, max_coarse=2000, max_sparse_levels=None,
                        poly_degree=None, poly_fraction=8.0, hard_poly_degree=None):
        """Build (host, once) and install the multigrid hierarchy of the constant potential
        block; afterwards Newton uses GMRES + field split (block Jacobi on the species,
        one V-cycle on the potential) and poisson_solve uses V-cycle-preconditioned CG.
        `poly_degree`: Chebyshev polynomial smoother of that many sweeps per leg instead of `nu`
        damped-Jacobi sweeps.  `hard_poly_degree`: such a cycle installed NEXT to the V(nu,nu)
        one, used with the alternative species sweeps of :meth:`set_fieldsplit` (one GPU)."""
        from . import amg
        if not self.model.poisson:
            raise ValueError("the model has no potential equation")
        self._check(self.lib.fedm_jacobian_poisson_only(self._h), "fedm_jacobian_poisson_only")
        ip = self.n_eq - 1
        K = self.block_csr(ip, ip)
        fixed = np.zeros(self.nv, dtype=bool)
        d = self._ddofs[self._ddofs % self.n_eq == ip] // self.n_eq
        fixed[d] = True            # device numbering, like K itself
        fixed[self.n_owned:] = True  # ghost rows are identity rows of the local block
        levels = amg.build_hierarchy(K, theta=theta, max_coarse=max_coarse, fixed=fixed,
                                     coords=self._coords_dev, max_sparse_levels=max_sparse_levels)
        self._last_hierarchy = levels
        if len(levels) < 2:                 # the whole block fits the dense coarsest solve: nothing to smooth
            poly_degree = hard_poly_degree = None
        if poly_degree:
            self.multigrid_levels = amg.install_poly(self._h, levels, poly_degree, poly_fraction)
        else:
            self.multigrid_levels = amg.install(self._h, levels, nu=nu, omega=omega)
        if hard_poly_degree:
            amg.install_poly(self._h, levels, hard_poly_degree, poly_fraction, alternative=True)
        return self.multigrid_levels

    def setup_multigrid_distributed(self, lm, group=None, theta=0.08, nu=1, omega=0.67, max_coarse=2000,
                                    prolongator_damping=4.0 / 3.0):
        """Several GPUs.  The potential block's multigrid is ONE global smoothed-aggregation
        hierarchy whose finest level is distributed and whose coarser levels are replicated:

        * aggregates are formed rank by rank (they never cross a partition boundary);
        * the prolongator is smoothed with the undecomposed operator: an owned vertex next to the
          boundary also interpolates from the neighbour's aggregates (the aggregate ids of the
          ghost vertices come from their owners), so constants are reproduced across ranks;
        * the level-1 operator is the sum of the ranks' contributions P_r^T K_r P (owned rows of K
          reach into ghost columns, whose prolongator rows come from the neighbours); every rank
          builds the hierarchy below it from the same matrix;
        * on the device the finest level is smoothed as a distributed operator (ghost values
          exchanged inside the V-cycle); its restriction writes straight into the global level-1
          vector, which is all-reduced once per cycle.

        With rank-local hierarchies (block Jacobi over ranks) GMRES needs 3x (2 ranks) to 4.5x
        (4 ranks) the single-GPU iterations.  The exchanges at set-up go through the process group."""
        import scipy.sparse as sp
        import torch.distributed as dist
        from . import amg
        if not self.model.poisson:
            raise ValueError("the model has no potential equation")
        self._check(self.lib.fedm_jacobian_poisson_only(self._h), "fedm_jacobian_poisson_only")
        ip = self.n_eq - 1
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        n_own, nv = lm.n_owned, self.nv
        to_dev, to_loc = self._inv, self._order            # local -> device index and back

        def from_neighbours(per_vertex):
            """Values of this rank's ghost vertices from their owners (list per neighbour);
            per_vertex is indexed by the partition's local vertex number."""
            send = {int(q): per_vertex[lm.send_idx[lm.send_ptr[k]:lm.send_ptr[k + 1]]]
                    for k, q in enumerate(lm.neighbours)}
            got = [None] * world
            dist.all_gather_object(got, send, group=group)
            return [got[int(q)][rank] for q in lm.neighbours]

        def gather(obj):
            out = [None] * world
            dist.all_gather_object(out, obj, group=group)
            return out

        # everything below in the partition's local numbering (owned vertices first)
        K = sp.csr_matrix(self.block_csr(ip, ip))[to_dev][:, to_dev]
        fixed = np.zeros(nv, dtype=bool)
        fixed[to_loc[self._ddofs[self._ddofs % self.n_eq == ip] // self.n_eq]] = True
        fixed[n_own:] = True
        diag = K.diagonal()
        for k, blk in enumerate(from_neighbours(diag)):     # ghost rows: their owners' diagonal
            diag[n_own + lm.recv_ptr[k]:n_own + lm.recv_ptr[k + 1]] = blk
        K = (K + sp.diags(diag - K.diagonal())).tocsr()
        step = amg.tentative_prolongator(K, theta, ~fixed, self.coords)
        if step is None:
            raise RuntimeError("multigrid coarsening of the finest level stalled")
        T, cxy = step                                        # n_local x n1 (ghost / fixed rows empty)
        n1 = T.shape[1]
        offset = np.concatenate([[0], np.cumsum(gather(int(n1)))]).astype(np.int64)
        n_g = int(offset[-1])
        # tentative prolongator with global columns, ghost rows filled from their owners
        agg = np.full(nv, -1, dtype=np.int64)
        Tc = T.tocoo()
        agg[Tc.row] = Tc.col + offset[rank]
        for k, blk in enumerate(from_neighbours(agg)):
            agg[n_own + lm.recv_ptr[k]:n_own + lm.recv_ptr[k + 1]] = blk
        has = agg >= 0
        T_ext = sp.csr_matrix((np.ones(int(has.sum())), (np.nonzero(has)[0], agg[has])), shape=(nv, n_g))
        # smoothed prolongator rows of the owned vertices: (I - w/rho D^-1 K) T with the whole K row
        DinvK = sp.diags(1.0 / diag[:n_own]) @ K[:n_own]
        rho = max(gather(float(np.abs(DinvK).sum(axis=1).max())))
        free_own = sp.diags((~fixed[:n_own]).astype(np.float64))
        P_own = (free_own @ (T_ext[:n_own] - (prolongator_damping / rho) * (DinvK @ T_ext))).tocsr()
        P_own.eliminate_zeros()
        # ... and of the ghost vertices, from their owners, for the Galerkin product
        P_ext = sp.vstack([P_own] + [sp.csr_matrix(blk) for blk in from_neighbours(P_own)]).tocsr() \
            if lm.n_ghost else P_own
        contribution = (P_own.T @ (K[:n_own] @ P_ext)).tocsr()            # n_g x n_g, sparse
        pieces = gather((contribution, cxy))
        A1 = pieces[0][0]
        for piece, _ in pieces[1:]:
            A1 = A1 + piece
        A1 = (0.5 * (A1 + A1.T)).tocsr()
        coords1 = np.vstack([c for _, c in pieces])
        # device: finest level (device numbering) with the global level-1 space as its coarse space
        R_dev = sp.vstack([P_own, sp.csr_matrix((nv - n_own, n_g))]).tocsr()[to_loc].T.tocsr()
        if self.halo_depth > 1:
            # deep halos: the correction P x_c is formed on the ghost layers too (their prolongator rows came
            # from the owners for the Galerkin product), so that the post-smoothing of the owned rows finds
            # exact neighbours without an exchange; the restriction sums OWNED rows only (each fine row once
            # over all ranks).  The outermost layer (identity rows) gets no correction.
            keep = np.ones(nv)
            keep[np.asarray(self._identity, dtype=np.int64)] = 0.0
            P_dev = (sp.diags(keep) @ P_ext).tocsr()[to_loc]
        else:
            P_dev = R_dev.T.tocsr()
        K_dev = K[to_loc][:, to_loc]
        local_sizes = amg.install(self._h, [(K_dev, P_dev), (A1, None)], nu=nu, omega=omega, dense_coarse=False,
                                  restrictions=[R_dev])
        levels_g = amg.build_hierarchy(A1, theta=theta, max_coarse=max_coarse, coords=coords1)
        global_sizes = amg.install_global(self._h, levels_g, n_g, 0, nu=nu, omega=omega)
        self.multigrid_levels = [local_sizes[0], f"{n1} of {n_g} global"] + global_sizes[1:]
        return self.multigrid_levels

    def set_fieldsplit(self, weights=(0.8, 0.8, 0.8), hard_weights=None, switch_above=5.0, back_below=3.5):
        """Richardson weights of the species-block sweeps; :func:`chebyshev_weights` gives the
        optimal ones for a spectrum interval of Duu^-1 Juu.  `hard_weights`: a second set (a cheaper
        one with the default species-first order, a higher degree with the potential-first order) used
        while Newton solves need >= `switch_above` Krylov steps per Newton iteration (until one
        needs <= `back_below` again)."""
        w = np.ascontiguousarray(weights, dtype=np.float64)
        self._check(self.lib.fedm_set_fieldsplit(self._h, int(w.size), _dp(w)), "fedm_set_fieldsplit")
        if hard_weights is not None:
            a = np.ascontiguousarray(hard_weights, dtype=np.float64)
            self._check(self.lib.fedm_set_fieldsplit_alternative(self._h, int(a.size), _dp(a),
                                                                 float(switch_above), float(back_below)),
                        "fedm_set_fieldsplit_alternative")

    def fieldsplit_policy(self):
        """Which preconditioner set ran (``fedm_fieldsplit_policy``): policy, active set, solves under each."""
        out = (C.c_int64 * 4)()
        self._check(self.lib.fedm_fieldsplit_policy(self._h, out), "fedm_fieldsplit_policy")
        return dict(policy="measured (wall time per Newton iteration)" if out[0] else "Krylov counts",
                    alternative_active=bool(out[1]), solves_main_set=int(out[2]), solves_alternative_set=int(out[3]))

    def clear_multigrid(self):
        self.lib.fedm_amg_clear(self._h)

    # -- solves --------------------------------------------------------------
    def newton_solve(self, rtol=1e-9, max_it=50, atol=1e-10, stol=1e-16,
                     ksp_restart=30, ksp_rtol=1e-5, ksp_atol=1e-50, ksp_max_it=10000):
        # (watch_component: the field whose relative change adaptive_solver asks for right after
        # the solve -- its two sums then ride on the final residual check's publication)
        watch = getattr(self, "watch_component", None)
        o = _lib.NewtonOpts(rtol, atol, stol, max_it, ksp_restart, ksp_rtol, ksp_atol, ksp_max_it,
                            0 if watch is None else int(watch) + 1)
        r = _lib.NewtonReport()
        rc = self.lib.fedm_newton_solve(self._h, C.byref(o), C.byref(r))
        self.last_report = NewtonReport(r.iterations, bool(r.converged), r.linear_iterations,
                                        r.fnorm0, r.fnorm)
        self._check(rc, "fedm_newton_solve")
        return r.iterations, True

    def poisson_solve(self, rtol=1e-10, max_it=20000):
        its = C.c_int()
        self._check(self.lib.fedm_poisson_solve(self._h, float(rtol), int(max_it), C.byref(its)),
                    "fedm_poisson_solve")
        return its.value

    def field_error(self, component):
        e = C.c_double()
        self._check(self.lib.fedm_field_error(self._h, int(component), C.byref(e)), "fedm_field_error")
        return e.value

    def plane_masks(self):
        """(kept, zero): bit masks over the n_eq x n_eq value planes of a Jacobian block that the
        assembly keeps between assemblies / that the SpMV skips (``fedm_plane_masks``)."""
        kept, zero = C.c_uint32(), C.c_uint32()
        self._check(self.lib.fedm_plane_masks(self._h, C.byref(kept), C.byref(zero)), "fedm_plane_masks")
        return kept.value, zero.value

    def set_assembly(self, kind):
        """'patch' (LDS patches, default) or 'colour' (global colouring, bitwise reproducible)."""
        code = {"colour": 0, "patch": 1}[kind]
        self._check(self.lib.fedm_set_assembly(self._h, code), "fedm_set_assembly")

    def set_preconditioner_side(self, side):
        """'right' (flexible GMRES, true residual norm; default for LFA models) or 'left'
        (preconditioned residual norm; default for LMEA models)."""
        code = {"left": 0, "right": 1}[side]
        self._check(self.lib.fedm_set_preconditioner_side(self._h, code), "fedm_set_preconditioner_side")

    def set_fieldsplit_order(self, order):
        """'lower' (default: species first; the true residual tracks the error of the iterate) or
        'upper' (V-cycle on the potential block first, species sweeps on the residual minus
        J_u,phi z_phi: the residual test passes in half the Krylov steps late in a streamer run, but
        the potential is left at V-cycle accuracy -- include/fedm_hip.h); order of the
        block-triangular split when it is on the right of the operator.  Takes effect with the next
        Jacobian assembly."""
        code = {"lower": 0, "upper": 1}[order]
        self._check(self.lib.fedm_set_fieldsplit_order(self._h, code), "fedm_set_fieldsplit_order")

    # -- measurement ----------------------------------------------------------
    def species_planes_check(self):
        """Test hook: the field split's planes as they stand against the separate pass over the Jacobian as it stands:
        (relative difference of Duu^-1, absolute of the half-precision species planes, relative of the coupling plane,
        whether the last set-up was the one fused into the assembly)."""
        out = (C.c_double * 4)()
        self._check(self.lib.fedm_debug_species_planes_check(self._h, out), "fedm_debug_species_planes_check")
        return float(out[0]), float(out[1]), float(out[2]), bool(out[3])

    def time_kernel(self, kind, repeats=20):
        ms = C.c_double()
        self._check(self.lib.fedm_time_kernel(self._h, int(kind), int(repeats), C.byref(ms)),
                    "fedm_time_kernel")
        return ms.value

    def profile(self, enable=True):
        """True / 1: time the assembly kernels only (the solver is untouched).  2: also time the
        Jacobian SpMV and the V-cycle -- GMRES then launches kernel by kernel instead of
        replaying its per-iteration graphs, so use it for a separate profiling pass."""
        self._check(self.lib.fedm_profile(self._h, int(enable)), "fedm_profile")

    def profile_read(self):
        """{kind: (total ms, launches)} of the kernels timed since profile(True)."""
        names = {0: "assembly_FJ", 1: "spmv", 2: "assembly_F", 3: "vcycle"}
        out = {}
        for k, name in names.items():
            ms, cnt = C.c_double(), C.c_int64()
            self._check(self.lib.fedm_profile_read(self._h, k, C.byref(ms), C.byref(cnt)),
                        "fedm_profile_read")
            out[name] = (ms.value, cnt.value)
        return out

    def gd_assembly_kernel_name(self):
        """The LMEA F + J assembly kernels in use (bench.py's glow-discharge roofline block)."""
        return ("gd_jacobian_rows_kernel<.., BALL> + gd_gather_dest_rows_kernel (hand-derived element blocks of all cells, the "
                "three column vertices side by side at one wave per SIMD; summed per stored matrix position and row; F + J)")

    def sizes(self):
        v = [C.c_int64() for _ in range(6)]
        self.lib.fedm_sizes(self._h, *[C.byref(x) for x in v])
        keys = ("n_vertices", "n_cells", "n_eq", "nnz_blocks", "stored_blocks", "n_colours")
        out = dict(zip(keys, (x.value for x in v)))
        kept, zero = self.plane_masks()
        out["kept_planes"], out["zero_planes"] = bin(kept).count("1"), bin(zero).count("1")
        info = (C.c_int64 * 9)()
        self._check(self.lib.fedm_pattern_info(self._h, info), "fedm_pattern_info")
        out.update(zip(("n_slices", "max_patch_cells", "max_patch_width", "max_patch_verts", "cell_visits",
                        "halo_vertices"), (int(v) for v in info[:6])))
        out["assembly_variant"] = ("global colouring", "lds-patches/unrolled", "lds-patches", "lds-patches/one-pass")[int(info[6])]
        out["patch_threads"] = int(info[7])
        out["model_structure"] = "compiled in" if info[8] else "run time"
        return out

    def assembly_variant(self):
        return self.sizes()["assembly_variant"]

    def fieldsplit_apply(self, t):
        """z = Minv t with the field split of the current Jacobian (test hook; call jacobian() first)."""
        t = np.ascontiguousarray(t, dtype=np.float64).reshape(-1)
        assert t.size == self.n
        z = np.empty(self.n)
        self._check(self.lib.fedm_debug_fieldsplit_apply(self._h, _dp(t), _dp(z)), "fedm_debug_fieldsplit_apply")
        return z

    def configure_fieldsplit_tiles(self, on, slices_per_tile=0, layers=0, threads=0, multigrid=True):
        """Test hook: species sweeps on tiles (several per launch) or one launch each; multigrid=False keeps the
        finest multigrid level's sweeps as kernels of their own."""
        mode = (1 if on else 0) | (0 if multigrid else 2)
        self._check(self.lib.fedm_debug_fieldsplit_tiles(self._h, mode, int(slices_per_tile), int(layers),
                                                         int(threads)), "fedm_debug_fieldsplit_tiles")

    def fieldsplit_tiles(self):
        """How the species sweeps of the field split run here: None = one launch per sweep; else the tiles of
        csrc/fs_tiles.hip (several sweeps per launch, the vertex layers around a tile of slices in LDS)."""
        info = (C.c_int64 * 10)()
        rc = self.lib.fedm_fieldsplit_tiles_info(self._h, info)
        if rc < 0:
            self._check(rc, "fedm_fieldsplit_tiles_info")
        if rc == 0:
            return None
        return dict(zip(("n_tiles", "slices_per_tile", "layers", "row_width", "max_vertices", "max_rows", "bytes",
                         "threads", "total_rows", "total_vertices"), (int(v) for v in info[:10])))


def rccl_unique_id():
    """128-byte ncclUniqueId (call on one rank, broadcast to the others)."""
    buf = C.create_string_buffer(128)
    lib = _lib.load()
    if lib.fedm_comm_unique_id(buf) != 0:
        raise RuntimeError(f"fedm_comm_unique_id failed: {_lib.last_error()}")
    return buf.raw


def chebyshev_weights(m, lam_min=0.5, lam_max=2.0):
    """Richardson weights whose residual polynomial is the degree-m Chebyshev polynomial on
    [lam_min, lam_max] (defaults: the diagonally scaled P1 mass matrix in 2-D)."""
    theta, delta = 0.5 * (lam_max + lam_min), 0.5 * (lam_max - lam_min)
    k = np.arange(1, m + 1)
    return 1.0 / (theta + delta * np.cos((2 * k - 1) * np.pi / (2 * m)))
