"""Set-up of the smoothed-aggregation multigrid hierarchy for the potential block.

The Poisson stiffness block of FEDM's coupled system,
``2*pi*r*inner(grad(u), grad(v))*dx`` (fedm/functions.py:401) with Dirichlet
rows, does not change during a run, so its hierarchy is built once per mesh on
the host (cold path, numpy/scipy + a native greedy aggregation) and uploaded;
every V-cycle then runs on the device as sliced-ELL SpMV kernels.  The
reference leaves this work to MUMPS (examples) or PETSc's default PC (test
harness, fedm_streamer.py:32); see DESIGN.md "Linear solver".
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib


def aggregate(A, theta):
    """Greedy aggregation on the strength graph |a_ij| >= theta*sqrt(a_ii a_jj)."""
    A = A.tocsr()
    n = A.shape[0]
    d = np.abs(A.diagonal())
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    strong = (np.abs(A.data) >= theta * np.sqrt(d[rows] * d[A.indices])).astype(np.uint8)
    strong[A.indices == rows] = 0
    agg = np.empty(n, dtype=np.int32)
    nagg = C.c_int32()
    fn = _lib.load().fedm_amg_aggregate
    indptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    rc = fn(C.c_int32(n), indptr.ctypes.data_as(C.POINTER(C.c_int64)),
            indices.ctypes.data_as(C.POINTER(C.c_int32)),
            strong.ctypes.data_as(C.POINTER(C.c_uint8)),
            agg.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nagg))
    if rc != 0:
        raise RuntimeError("fedm_amg_aggregate failed")
    return agg, nagg.value


def _z_order(xy):
    from .device import locality_order
    return locality_order(xy)


def tentative_prolongator(A, theta, free, xy=None):
    """One aggregation step: the piecewise-constant prolongator T (n x n_aggregates, zero rows
    for the fixed vertices) and the aggregates' centroids (or None).  Returns None when the
    coarsening stalls.  With coordinates the nodes are visited in a lexicographic sweep and the
    aggregates are numbered along a Z-curve of their centroids."""
    n = A.shape[0]
    idx = np.nonzero(free)[0]
    if xy is not None:                   # lexicographic sweep: by y, then x
        idx = idx[np.lexsort((xy[idx, 0], xy[idx, 1]))]
    Af = A[idx][:, idx].tocsr()
    agg_f, nagg = aggregate(Af, theta)
    if nagg >= 0.8 * idx.size:
        return None
    xy_next = None
    if xy is not None:
        cnt = np.bincount(agg_f, minlength=nagg).astype(np.float64)
        cxy = np.stack([np.bincount(agg_f, weights=xy[idx, d], minlength=nagg) / cnt
                        for d in range(2)], axis=1)
        order = _z_order(cxy)            # new -> old aggregate id
        relabel = np.empty(nagg, dtype=np.int64)
        relabel[order] = np.arange(nagg)
        agg_f = relabel[agg_f]
        xy_next = cxy[order]
    T = sp.csr_matrix((np.ones(idx.size), (idx, agg_f)), shape=(n, nagg))
    return T, xy_next


def build_hierarchy(A, theta=0.08, omega=4.0 / 3.0, max_coarse=600, max_levels=12,
                    fixed=None, coords=None, max_sparse_levels=None):
    """Levels [(A_l, P_l)] of a smoothed-aggregation hierarchy; the last A is coarsest.

    ``fixed``: boolean mask of identity rows (Dirichlet / padding); they are kept out of
    the aggregates (their prolongator rows are zero) so the coarse problems stay SPD.
    ``max_sparse_levels``: once that many sparse levels exist, further coarsening steps are
    composed into the last prolongator (P <- P P_next) instead of adding levels: every level
    costs ~5 latency-bound kernel launches per cycle on the device.
    ``coords`` (n,2), optional: nodes are visited in a lexicographic sweep (compact, regular
    aggregates whatever the numbering) and the aggregates of every level are numbered along
    a Z-curve of their centroids, so that restriction/prolongation gathers stay within a
    few cache lines on the device."""
    A = sp.csr_matrix(A)
    levels = []
    free = np.ones(A.shape[0], dtype=bool) if fixed is None else ~np.asarray(fixed, dtype=bool)
    xy = None if coords is None else np.asarray(coords, dtype=np.float64)
    while A.shape[0] > max_coarse and len(levels) < max_levels - 1:
        step = tentative_prolongator(A, theta, free, xy)
        if step is None:                    # coarsening stalled
            break
        T, xy_next = step
        d = A.diagonal()
        DinvA = sp.diags(1.0 / d) @ A
        rho = np.abs(DinvA).sum(axis=1).max()          # Gershgorin bound on rho(D^-1 A)
        P = (T - (omega / rho) * (DinvA @ T)).tocsr()
        P = sp.diags(free.astype(np.float64)) @ P       # fixed rows interpolate nothing
        P.eliminate_zeros()
        Ac = (P.T @ A @ P).tocsr()
        if max_sparse_levels is not None and len(levels) >= max_sparse_levels:
            A_prev, P_prev = levels[-1]
            levels[-1] = (A_prev, (P_prev @ P).tocsr())
        else:
            levels.append((A, P.tocsr()))
        A = Ac
        free = np.ones(A.shape[0], dtype=bool)
        if xy is not None:
            xy = xy_next
    levels.append((A, None))
    return levels


def _csr_struct(M, keep):
    M = sp.csr_matrix(M)
    M.sort_indices()
    indptr = np.ascontiguousarray(M.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(M.indices, dtype=np.int32)
    values = np.ascontiguousarray(M.data, dtype=np.float64)
    keep.extend([indptr, indices, values])
    return _lib.Csr(M.shape[0], M.shape[1], indptr.ctypes.data_as(C.POINTER(C.c_int64)),
                    indices.ctypes.data_as(C.POINTER(C.c_int32)),
                    values.ctypes.data_as(C.POINTER(C.c_double)))


def _pack(levels, dense_coarse=True, restrictions=None):
    """restrictions: R per level where it is not the prolongator's transpose (several GPUs with deep halos:
    the prolongator also has rows for ghost vertices, the restriction sums owned rows only)."""
    keep = []
    n = len(levels)
    A = (_lib.Csr * n)(*[_csr_struct(a, keep) for a, _ in levels])
    P = (_lib.Csr * max(n - 1, 1))(*[_csr_struct(p, keep) for _, p in levels[:-1]])
    Rs = [p.T for _, p in levels[:-1]]
    for l, r in enumerate(restrictions or []):
        Rs[l] = r
    R = (_lib.Csr * max(n - 1, 1))(*[_csr_struct(r, keep) for r in Rs])
    coarse = None
    if dense_coarse:
        n_coarse = levels[-1][0].shape[0]
        if n_coarse > 8192:   # a dense inverse of that size is the wrong tool (n^2 doubles on host and device)
            raise RuntimeError(f"multigrid coarsening stalled at {n_coarse} rows (levels "
                               f"{[a.shape[0] for a, _ in levels]}): adjust theta / max_coarse")
        coarse = np.ascontiguousarray(np.linalg.inv(levels[-1][0].toarray()))
        keep.append(coarse)
    return n, A, P, R, coarse, keep


def install(handle, levels, nu=2, omega=0.67, dense_coarse=True, restrictions=None):
    """Upload a hierarchy built by :func:`build_hierarchy` into a device context.
    ``dense_coarse=False``: no inverse of the last level (a global hierarchy takes over there,
    :func:`install_global`)."""
    lib = _lib.load()
    n, A, P, R, coarse, keep = _pack(levels, dense_coarse, restrictions)
    cptr = coarse.ctypes.data_as(C.POINTER(C.c_double)) if coarse is not None else None
    rc = lib.fedm_amg_setup(handle, n, A, P, R, cptr, int(nu), float(omega))
    if rc != 0:
        raise RuntimeError(f"fedm_amg_setup failed ({rc}): {_lib.last_error()}")
    return [a.shape[0] for a, _ in levels]


def chebyshev_smoother_weights(levels, degree, fraction=8.0, safety=1.05, power_iterations=40):
    """Richardson weights (reciprocals of the Chebyshev roots on [lambda_max / fraction, lambda_max]
    of Dinv A, lambda_max from a power iteration) for every level but the coarsest: array
    [n_levels - 1, degree]."""
    out = np.empty((len(levels) - 1, degree))
    rng = np.random.default_rng(0)
    k = np.arange(1, degree + 1)
    for l, (A, _) in enumerate(levels[:-1]):
        dinv = 1.0 / A.diagonal()
        x = rng.standard_normal(A.shape[0])
        lam = 1.0
        for _ in range(power_iterations):
            y = dinv * (A @ x)
            lam = float(np.linalg.norm(y) / np.linalg.norm(x))
            x = y / np.linalg.norm(y)
        hi = safety * lam
        lo = hi / fraction
        out[l] = 1.0 / (0.5 * (hi + lo) + 0.5 * (hi - lo) * np.cos((2 * k - 1) * np.pi / (2 * degree)))
    return out


def install_poly(handle, levels, degree=2, fraction=8.0, alternative=False, weights=None):
    """The hierarchy of :func:`build_hierarchy` with a Chebyshev polynomial smoother of `degree`
    sweeps per leg (``fedm_amg_setup_poly``); `alternative`: next to the installed V-cycle, for
    the Newton solves that need many Krylov steps."""
    lib = _lib.load()
    n, A, P, R, coarse, keep = _pack(levels, True)
    if weights is None:
        weights = chebyshev_smoother_weights(levels, degree, fraction)
    w = np.ascontiguousarray(np.broadcast_to(np.asarray(weights, dtype=np.float64), (len(levels) - 1, degree)))
    rc = lib.fedm_amg_setup_poly(handle, n, A, P, R, coarse.ctypes.data_as(C.POINTER(C.c_double)), int(degree),
                                 w.ctypes.data_as(C.POINTER(C.c_double)), 1 if alternative else 0)
    if rc != 0:
        raise RuntimeError(f"fedm_amg_setup_poly failed ({rc}): {_lib.last_error()}")
    return [a.shape[0] for a, _ in levels]


def install_global(handle, levels, n_global, offset, nu=2, omega=0.67):
    """Several GPUs: the replicated hierarchy below the rank-local finest level."""
    lib = _lib.load()
    n, A, P, R, coarse, keep = _pack(levels, True)
    rc = lib.fedm_amg_set_global_hierarchy(handle, int(n_global), int(offset), n, A, P, R,
                                           coarse.ctypes.data_as(C.POINTER(C.c_double)), int(nu), float(omega))
    if rc != 0:
        raise RuntimeError(f"fedm_amg_set_global_hierarchy failed ({rc}): {_lib.last_error()}")
    return [a.shape[0] for a, _ in levels]
