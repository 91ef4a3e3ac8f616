"""Time-of-flight verification case (oracle; test infra).

Restates the reference's test harness
tests/integrated_tests/time_of_flight/fedm_tof.py:18-179 (same constants,
same loop, same output) and the full-size example
examples/time_of_flight/fedm-tof.py on top of :mod:`oracle.forms`.
One balance equation in the log variable, constant drift velocity
``w = (0, wez)``, constant diffusion, Expression source (degree 2),
no Poisson row, no Dirichlet rows.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .forms import LFAModel
from .lagrange import interpolation_matrix, lattice, p1_basis
from .mesh import rectangle_right
from .newton import newton_solve
from .quadrature import triangle_rule

DOLFIN_EPS = 3.0e-16
WEZ, DE, ALPHA_E = 1.7e5, 0.12, 5009.51          # fedm_tof.py:51-53


def log_density(x, t, eps=0.0):
    """u_analytical, fedm_tof.py:113 (eps=DOLFIN_EPS variant :124)."""
    r, z = x[..., 0], x[..., 1]
    val = np.exp(-((z - WEZ * t) ** 2 + r ** 2) / (4.0 * DE * t) + ALPHA_E * WEZ * t) \
        / (4.0 * DE * t * np.pi) ** 1.5
    return np.log(val + eps)


def source(x, t):
    """f, fedm_tof.py:122."""
    r, z = x[..., 0], x[..., 1]
    return np.exp(-((z - WEZ * t) ** 2 + r ** 2) / (4.0 * DE * t) + ALPHA_E * WEZ * t) \
        * (WEZ * ALPHA_E) / (8.0 * np.pi ** 1.5 * (DE * t) ** 1.5)


def cell_nodes(mesh, k):
    """Physical coordinates of the P_k lattice nodes of every cell, (Nc,n,2)."""
    lam = lattice(k)
    phi = p1_basis(lam)                                   # (n,3)
    return np.einsum("na,cad->cnd", phi, mesh.coords[mesh.cells])


def mass_matrix(mesh):
    """Cartesian P1 mass matrix (``u*v*dx`` of ``project``/``norm``)."""
    x = mesh.coords[mesh.cells]
    d1, d2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
    det = np.abs(d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0])
    ref = (np.ones((3, 3)) + np.eye(3)) / 24.0
    vals = det[:, None, None] * ref[None]
    c = mesh.cells.astype(np.int64)
    rows = np.broadcast_to(c[:, :, None], vals.shape).ravel()
    cols = np.broadcast_to(c[:, None, :], vals.shape).ravel()
    return sp.coo_matrix((vals.ravel(), (rows, cols)), shape=(mesh.nv, mesh.nv)).tocsc(), det


def project_exp(mesh, M, det, values_at_q, degree):
    """``project(exp(w), V)``: M n = int exp(w) v dx with the given rule."""
    xq, wq = triangle_rule(degree)
    phi = p1_basis(xq)                                    # (nq,3)
    rhs_e = np.einsum("q,cq,qa->ca", wq, np.exp(values_at_q), phi) * det[:, None]
    b = np.bincount(mesh.cells.ravel(), weights=rhs_e.ravel(), minlength=mesh.nv)
    return spla.splu(M).solve(b)


def run(nx=40, ny=40, box_width=2.5e-4, box_height=5e-4, t0=2.5e-9, T_final=2.6e-9,
        dt_init=1e-12, t_output=2.6e-9, rtol=1e-10, max_it=50,
        qdeg_time=8, qdeg_flux=8, qdeg_source=8, solver=None, model_hook=None):
    """Returns dict(n_num, n_exact, relative_error, u, h_max, steps).

    ``solver(model, U, Uold, Uold1, dt, dt_old)`` may replace the oracle's
    Newton solve (used by the GPU parity tests to run the same harness).
    Quadrature: the script sets no degree (fedm_tof.py:22-25), so UFL estimates
    it.  The three integrals of fedm/functions.py:357-366 share one measure and
    empty metadata, UFL sums their integrands before estimating, and the
    largest term wins: exp(u) [1+2] * u_part [3: ``trp1**2.0`` is a float power,
    heuristic +2, times u_old] * v [1] * r [1] = 8 -> FIAT's collapsed
    Gauss-Jacobi rule with 5x5 points.  Pinned by the golden field
    (tests/golden/tof_golden.npz): degrees 6 or 7, or per-term degrees, miss the
    reference's own 1e-5 tolerance by 1-2 orders of magnitude; 8 meets it and
    reproduces the logged relative error to 1e-12.
    """
    mesh = rectangle_right(0.0, 0.0, box_width, box_height, nx, ny)
    model = LFAModel(mesh, n_species=1, poisson=False,
                     eq_type=["drift-diffusion-reaction"], Z=[-1.0],
                     D=[DE], drift_w=[(0.0, WEZ)],
                     qdeg_time=qdeg_time, qdeg_flux=qdeg_flux, qdeg_source=qdeg_source,
                     qdeg_ext=qdeg_source)
    if model_hook is not None:
        model_hook(model)
    x = mesh.coords
    U_old = log_density(x, t0)[:, None].copy()
    U_old1 = U_old.copy()
    U = log_density(x, t0, DOLFIN_EPS)[:, None].copy()
    p2 = cell_nodes(mesh, 2)

    t, dt, dt_old = t0, dt_init, 1e30
    out = {}
    steps = 0
    M, det = mass_matrix(mesh)
    while abs(t - T_final) / T_final > 1e-6:
        U_old1[:] = U_old
        U_old[:] = U
        t += dt
        model.set_ext_source(0, 2, source(p2, t))
        if solver is None:
            newton_solve(model, U, U_old, U_old1, dt, dt_old, rtol, max_it)
        else:
            solver(model, U, U_old, U_old1, dt, dt_old)
        steps += 1
        if abs(t - t_output) / t_output <= 1e-6:
            # project(exp(u_analytical)), degree-3 Expression: est. degree 3+2+1 = 6
            xq6, _ = triangle_rule(6)
            ua = log_density(cell_nodes(mesh, 3), t) @ interpolation_matrix(3, xq6).T
            n_exact = project_exp(mesh, M, det, ua, 6)
            # project(exp(u_new)): est. degree 1+2+1 = 4
            xq4, _ = triangle_rule(4)
            un = U[mesh.cells, 0] @ p1_basis(xq4).T
            n_num = project_exp(mesh, M, det, un, 4)
            e = n_num - n_exact
            out = dict(n_num=n_num, n_exact=n_exact,
                       relative_error=float(np.sqrt(e @ (M @ e)) / np.sqrt(n_exact @ (M @ n_exact))))
        if t > (t0 + dt_init):
            dt_old = dt
    out.update(u=U[:, 0].copy(), h_max=mesh.hmax(), steps=steps, mesh=mesh)
    return out
