"""Time-of-flight verification case on the device path.

Counterpart of examples/time_of_flight/fedm-tof.py (160x320 mesh, 500 steps)
and of its test harness tests/integrated_tests/time_of_flight/fedm_tof.py
(40x40, 100 steps): one electron balance equation in the log variable with a
constant drift velocity, constant diffusion and the analytic source as a
degree-2 Expression.  No quadrature degree is set there, so UFL's estimate (8)
applies -- see DESIGN.md.
"""
import numpy as np

from ..device import DeviceProblem, Model
from ..mesh import RectangleMesh
from ..termsum import TermSum

WEZ, DE, ALPHA_E = 1.7e5, 0.12, 5009.51      # fedm-tof.py:51-53
DOLFIN_EPS = 3.0e-16


def model():
    return Model(n_species=1, poisson=False, eq_type=["drift-diffusion-reaction"], Z=[-1.0],
                 D=[TermSum.const(DE)], drift_w=[(0.0, WEZ)],
                 quadrature_degree=8, ext_source_degree=[2])


def analytic_log_density(x, t, eps=0.0):
    """u_analytical, fedm-tof.py:107 (and the initial guess with DOLFIN_EPS, :120)."""
    r, z = x[..., 0], x[..., 1]
    n = np.exp(-((z - WEZ * t) ** 2 + r ** 2) / (4.0 * DE * t) + ALPHA_E * WEZ * t) \
        / (4.0 * DE * t * np.pi) ** 1.5
    return np.log(n + eps)


def source(x, t):
    """f, fedm-tof.py:116."""
    r, z = x[..., 0], x[..., 1]
    return np.exp(-((z - WEZ * t) ** 2 + r ** 2) / (4.0 * DE * t) + ALPHA_E * WEZ * t) \
        * (WEZ * ALPHA_E) / (8.0 * np.pi ** 1.5 * (DE * t) ** 1.5)


SOURCE_STRING = ('exp(-(pow(x[1]-w*t, 2)+pow(x[0], 2))/(4.0*D*t)+alpha*w*t)*(w*alpha)'
                 '/(8*pow(pi,1.5)*pow(D*t, 1.5))')          # fedm-tof.py:116


def p2_nodes(coords, cells):
    """Coordinates of the 6 P2 lattice nodes of every cell, (Nc,6,2), lattice order."""
    lam = np.array([(i / 2, j / 2) for j in range(3) for i in range(3 - j)])
    phi = np.stack([1 - lam[:, 0] - lam[:, 1], lam[:, 0], lam[:, 1]], axis=1)
    return np.einsum("na,cad->cnd", phi, coords[cells])


def device_problem(nx, ny, box_width, box_height, device=0):
    mesh = RectangleMesh((0.0, 0.0), (box_width, box_height), nx, ny)
    return DeviceProblem(mesh.coords, mesh.cells, model(), device=device), mesh


def _mass_matrix(mesh):
    import scipy.sparse as sp
    x = mesh.coords[mesh.cells]
    d1, d2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
    det = np.abs(d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0])
    vals = det[:, None, None] * ((np.ones((3, 3)) + np.eye(3)) / 24.0)[None]
    c = mesh.cells.astype(np.int64)
    rows = np.broadcast_to(c[:, :, None], vals.shape).ravel()
    cols = np.broadcast_to(c[:, None, :], vals.shape).ravel()
    n = mesh.num_vertices()
    return sp.coo_matrix((vals.ravel(), (rows, cols)), shape=(n, n)).tocsc(), det


def _project_exp(mesh, M, det, values_at_q, wq, phi):
    """project(exp(w), V) (post-processing, fedm-tof.py:155-156): host-side mass solve."""
    import scipy.sparse.linalg as spla
    rhs = np.einsum("q,cq,qa->ca", wq, np.exp(values_at_q), phi) * det[:, None]
    b = np.bincount(mesh.cells.ravel(), weights=rhs.ravel(), minlength=mesh.num_vertices())
    return spla.splu(M).solve(b)


def run_harness(nx=40, ny=40, box_width=2.5e-4, box_height=5e-4, t0=2.5e-9, T_final=2.6e-9,
                dt_init=1e-12, t_output=2.6e-9, relative_tolerance=1e-10, maximum_iterations=50,
                ksp_rtol=1e-5, device=0):
    """The reference's ToF test harness (tests/integrated_tests/time_of_flight/fedm_tof.py)
    with the Newton solves on the device.  Returns n_num, n_exact, relative_error."""
    from .. import quadrature
    prob, mesh = device_problem(nx, ny, box_width, box_height, device)
    x = mesh.coords
    u_old = analytic_log_density(x, t0)
    u_new = analytic_log_density(x, t0, DOLFIN_EPS)
    prob.set_state(u_new, u_old, u_old)
    # the source as the reference gives it, a C++ Expression string (fedm-tof.py:116), evaluated on the
    # device at the P2 lattice nodes before every solve (fedm_ext_source_program / _eval)
    from .. import forms
    f = forms.Expression(SOURCE_STRING, D=DE, w=WEZ, alpha=ALPHA_E, t=t0, pi=np.pi, degree=2)
    ops, consts, names = forms.expression_program(f)
    prob.set_ext_source_program(0, ops, consts, len(names))
    import time
    t, dt, dt_old = t0, dt_init, 1e30
    out, steps, newton, linear = {}, 0, 0, 0
    step_seconds = 0.0
    while abs(t - T_final) / T_final > 1e-6:
        step_start = time.perf_counter()
        prob.shift_state()
        t += dt
        f.t = t
        prob.eval_ext_source(0, [getattr(f, n) for n in names])
        prob.set_step(dt, dt_old)
        prob.newton_solve(rtol=relative_tolerance, max_it=maximum_iterations, ksp_rtol=ksp_rtol)
        newton += prob.last_report.iterations
        linear += prob.last_report.linear_iterations
        steps += 1
        step_seconds += time.perf_counter() - step_start      # (the output projections below apart)
        if abs(t - t_output) / t_output <= 1e-6:
            U = prob.get_state()[:, 0]
            M, det = _mass_matrix(mesh)
            xq6, wq6 = quadrature.triangle(6)
            B3, lam3 = quadrature.lagrange_interpolation_matrix(3, xq6)
            phi3 = np.stack([1 - lam3[:, 0] - lam3[:, 1], lam3[:, 0], lam3[:, 1]], axis=1)
            p3 = np.einsum("na,cad->cnd", phi3, mesh.coords[mesh.cells])
            p1 = lambda xq: np.stack([1 - xq[:, 0] - xq[:, 1], xq[:, 0], xq[:, 1]], axis=1)
            n_exact = _project_exp(mesh, M, det, analytic_log_density(p3, t) @ B3.T, wq6, p1(xq6))
            xq4, wq4 = quadrature.triangle(4)
            n_num = _project_exp(mesh, M, det, U[mesh.cells] @ p1(xq4).T, wq4, p1(xq4))
            e = n_num - n_exact
            out = dict(n_num=n_num, n_exact=n_exact,
                       relative_error=float(np.sqrt(e @ (M @ e)) / np.sqrt(n_exact @ (M @ n_exact))))
        if t > (t0 + dt_init):
            dt_old = dt
    out.update(steps=steps, newton_iterations=newton, linear_iterations=linear, h_max=mesh.hmax(),
               loop_seconds=step_seconds)
    return out
