"""CPU oracle: 1-D time of flight (BASELINE configs[0], "plumbing, no GPU").

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates
examples/time_of_flight_1D/fedm-tof_1d.py: one drift-diffusion-reaction balance equation in the
log variable on ``IntervalMesh(4000, 0, 1e-3)`` with **P2** elements (8001 dofs, :87,:98),
constant drift ``w`` and diffusion ``D`` (:46-48, interpolated -> constants), the manufactured
source ``f`` interpolated into P2 (degree-2 Expression, :113), ``2 pi r = 1`` (functions.py:251),
no boundary terms, variable-step BDF2 with two BDF1 start-up steps (:168-171), Newton with
rtol 1e-10 (:22-24) and a direct solve, the relative L2 error of the density against the
analytic pulse at the output times (:155-160).

PARITY UNPINNED for this configuration: the reference ships no golden for the 1-D case (there is
no integrated test for it); the check is the method of exact solutions the example itself uses.
Quadrature degree 10 follows the UFL estimate pinned on the 2-D golden (sum of the factor
degrees, exp() + 2, float power + 2) applied to P2.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from .quadrature import interval_rule

W_DRIFT, D_E, ALPHA, X0, L_GAUSS = 1.7e5, 0.12, 5009.51, 3e-4, 0.00004   # fedm-tof_1d.py:46-50
DOLFIN_EPS = 3.0e-16


def log_density(x, t, eps=0.0):
    """u_analytical, fedm-tof_1d.py:104 (and the initial guess with + DOLFIN_EPS, :119)."""
    s = 1.0 + 4.0 * D_E * t / L_GAUSS ** 2
    return np.log(np.exp(-((x - X0 - W_DRIFT * t) / L_GAUSS) ** 2 / s + ALPHA * W_DRIFT * t) / np.sqrt(s) + eps)


def source(x, t):
    """f, fedm-tof_1d.py:113."""
    s = 1.0 + 4.0 * D_E * t / L_GAUSS ** 2
    return np.exp(-((x - X0 - W_DRIFT * t) / L_GAUSS) ** 2 / s + ALPHA * W_DRIFT * t) * (W_DRIFT * ALPHA) / np.sqrt(s)


class P2Interval:
    """P2 Lagrange on a uniform interval mesh: dofs = vertices 0..n, then cell midpoints."""

    def __init__(self, n_cells, length, qdeg=10):
        self.n, self.h = n_cells, length / n_cells
        self.ndof = 2 * n_cells + 1
        self.x = np.concatenate([np.linspace(0.0, length, n_cells + 1),
                                 (np.arange(n_cells) + 0.5) * self.h])
        c = np.arange(n_cells)
        self.cell_dofs = np.stack([c, c + 1, n_cells + 1 + c], axis=1)          # left, right, mid
        xi, w = interval_rule(qdeg)
        self.wq = w * self.h
        self.phi = np.stack([(1 - xi) * (1 - 2 * xi), xi * (2 * xi - 1), 4 * xi * (1 - xi)], axis=1)   # [q, 3]
        self.dphi = np.stack([4 * xi - 3, 4 * xi - 1, 4 - 8 * xi], axis=1) / self.h
        self.rows = np.repeat(self.cell_dofs, 3, axis=1).ravel()
        self.cols = np.tile(self.cell_dofs, (1, 3)).ravel()

    def at_q(self, nodal):
        v = nodal[self.cell_dofs]                       # [cell, 3]
        return v @ self.phi.T, v @ self.dphi.T          # value, derivative at [cell, q]

    def mass(self):
        Me = np.einsum("q,qa,qb->ab", self.wq, self.phi, self.phi)
        vals = np.broadcast_to(Me, (self.n, 3, 3)).ravel()
        return sp.csr_matrix((vals, (self.rows, self.cols)), shape=(self.ndof, self.ndof))


def residual_jacobian(sp2, u, uold, uold1, f_nodal, dt, dt_old, jac=True):
    """fedm/functions.py:350-359 (log representation, BDF2) with the flux of fedm-tof_1d.py:112."""
    tr = dt / dt_old
    trp1 = 1.0 + tr
    c_new = (1.0 + 2.0 * tr) / trp1
    uq, duq = sp2.at_q(u)
    uo, _ = sp2.at_q(uold)
    uo1, _ = sp2.at_q(uold1)
    fq, _ = sp2.at_q(f_nodal)
    n = np.exp(uq)
    u_part = (uq * (1.0 + 2.0 * tr) - trp1 ** 2 * uo + tr ** 2 * uo1) / trp1
    flux = -D_E * n * duq + W_DRIFT * n                                          # Gamma
    W = sp2.wq[None, :]
    R_loc = np.einsum("cq,qa->ca", W * (n * u_part / dt - fq), sp2.phi) - np.einsum("cq,qa->ca", W * flux, sp2.dphi)
    R = np.zeros(sp2.ndof)
    np.add.at(R, sp2.cell_dofs.ravel(), R_loc.ravel())
    if not jac:
        return R, None
    # d/du_b:  n phi_b (u_part + c_new)/dt phi_a  -  dGamma phi_a',  dGamma = n phi_b (w - D u') - D n phi_b'
    t1 = W * n * (u_part + c_new) / dt
    t2 = W * n * (W_DRIFT - D_E * duq)
    t3 = W * D_E * n
    J_loc = (np.einsum("cq,qa,qb->cab", t1, sp2.phi, sp2.phi) - np.einsum("cq,qa,qb->cab", t2, sp2.dphi, sp2.phi)
             + np.einsum("cq,qa,qb->cab", t3, sp2.dphi, sp2.dphi))
    J = sp.csr_matrix((J_loc.ravel(), (sp2.rows, sp2.cols)), shape=(sp2.ndof, sp2.ndof))
    return R, J


def newton(sp2, u, uold, uold1, f_nodal, dt, dt_old, rtol=1e-10, atol=1e-10, max_it=50):
    """PETSc SNES newtonls/basic (functions.py:1047) with a row-equilibrated direct solve."""
    R, J = residual_jacobian(sp2, u, uold, uold1, f_nodal, dt, dt_old)
    r0 = np.linalg.norm(R)
    for it in range(1, max_it + 1):
        s = 1.0 / np.maximum(np.abs(J).max(axis=1).toarray().ravel(), 1e-300)
        du = spla.spsolve((sp.diags(s) @ J).tocsc(), -(s * R))
        u = u + du
        R, J = residual_jacobian(sp2, u, uold, uold1, f_nodal, dt, dt_old)
        rn = np.linalg.norm(R)
        if rn < atol or rn <= rtol * r0 or np.linalg.norm(du) < 1e-16 * np.linalg.norm(u):
            return u, it
    raise RuntimeError("tof1d: Newton did not converge")


def relative_error(sp2, M, lu, u, t):
    """errornorm(project(exp(u)), project(exp(u_analytical)), 'l2') / norm(., 'l2'), :155-158:
    both densities are L2-projected into P2 first, then compared in the L2 function norm."""
    def project(values_at_q):
        b = np.zeros(sp2.ndof)
        np.add.at(b, sp2.cell_dofs.ravel(), np.einsum("cq,qa->ca", sp2.wq[None, :] * values_at_q, sp2.phi).ravel())
        return lu.solve(b)
    uq, _ = sp2.at_q(u)
    n_num = project(np.exp(uq))
    # the analytic density is evaluated at the quadrature points (the reference interpolates the
    # degree-3 Expression cell-wise first: an O(h^4) difference)
    xi, _ = interval_rule(10)
    xq = np.arange(sp2.n)[:, None] * sp2.h + xi[None, :] * sp2.h
    n_ex = project(np.exp(log_density(xq, t)))
    d = n_num - n_ex
    return float(np.sqrt(d @ (M @ d)) / np.sqrt(n_ex @ (M @ n_ex)))


def run(n_cells=4000, length=1e-3, dt=1e-11, n_steps=300, output_every=10, qdeg=10):
    """The time loop of fedm-tof_1d.py:137-172.  Returns [(t, relative error), ...] at the output
    times and the final nodal log density."""
    sp2 = P2Interval(n_cells, length, qdeg)
    M = sp2.mass()
    lu = spla.splu(M.tocsc())
    t = 0.0
    uold = uold1 = log_density(sp2.x, 0.0)
    u = log_density(sp2.x, 0.0, DOLFIN_EPS)
    dt_old = 1e30
    errors = []
    for step in range(1, n_steps + 1):
        uold1, uold = uold, u
        t = step * dt
        u, _ = newton(sp2, u, uold, uold1, source(sp2.x, t), dt, dt_old)
        if step % output_every == 0:
            errors.append((t, relative_error(sp2, M, lu, u, t)))
        if t > dt * (1.0 + 1e-12):              # BDF1 for the first two steps, :168-169
            dt_old = dt
    return errors, u, sp2
