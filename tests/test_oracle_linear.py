"""The non-logarithmic representation (`weak_form_balance_equation(..., log_representation=False)`,
fedm/functions.py:350-368; `Flux(..., logarithm_representation=False)`, :219-237) in the oracle.

PARITY UNPINNED: no example, test or golden of the reference uses this form.  What can be checked
without one: the assembled Jacobian is the derivative of the assembled residual, and the two
representations are the same equations -- with n = exp(u) every term of the linear form except the
time derivative equals the logarithmic one (the time terms differ by construction: BDF2 of n
against n times BDF2 of ln n)."""
import numpy as np
import pytest

from oracle import streamer as ost
from oracle.forms import LFAModel
from oracle.mesh import graded_axis, mark_boundaries, rectangle_right


def _model(mesh, log):
    tags = mark_boundaries(mesh, ost.BOUNDARIES)
    m = LFAModel(mesh, n_species=2, poisson=True, eq_type=["reaction", "drift-diffusion-reaction"],
                 Z=[1.0, -1.0], mu=[0.0, ost.MU_E], D=[0.0, ost.D_E], reactions=[(ost.K_ION, [0, 1], [1, 1])],
                 facet_tags=tags, bc_type=ost.BC_TYPE, qdeg=2, log_representation=log)
    z = mesh.coords[:, 1]
    cath, anod = np.nonzero(np.abs(z) < 3e-16)[0], np.nonzero(np.abs(z - ost.BOX) < 3e-16)[0]
    m.set_dirichlet(np.concatenate([cath, anod]) * 3 + 2,
                    np.concatenate([np.zeros(cath.size), np.full(anod.size, ost.U_W)]))
    return m


def _state(mesh, seed=3):
    rng = np.random.default_rng(seed)
    x, y = mesh.coords[:, 0] / ost.BOX, mesh.coords[:, 1] / ost.BOX
    lnn = np.zeros((mesh.nv, 2))
    lnn[:, 0] = 30.0 + 2.0 * np.sin(5 * x) * np.cos(3 * y) + 0.05 * rng.standard_normal(mesh.nv)
    lnn[:, 1] = 29.0 + 2.5 * np.cos(4 * x) * np.sin(6 * y) + 0.05 * rng.standard_normal(mesh.nv)
    phi = ost.U_W * y + 40.0 * np.sin(3 * x) * np.sin(np.pi * y)
    return lnn, phi


def test_linear_jacobian_is_the_derivative_of_the_residual():
    n = 10
    mesh = rectangle_right(0.0, 0.0, ost.BOX, ost.BOX, n, n, xs=graded_axis(ost.BOX, n, 2.0))
    m = _model(mesh, log=False)
    lnn, phi = _state(mesh)
    U = np.column_stack([np.exp(lnn), phi])
    Uo, Uo1 = U * (1.0 + 1e-3), U * (1.0 - 2e-3)
    dt, dt_old = 5e-12, 4e-12
    F, J = m.residual_jacobian(U, Uo, Uo1, dt, dt_old)
    rng = np.random.default_rng(0)
    V = rng.standard_normal(U.shape) * np.array([U[:, 0].mean(), U[:, 1].mean(), 1.0]) * 1e-3
    eps = 0.1     # V is 1e-3 of the state: a 1e-4 relative perturbation (smaller steps drown in the rounding of rows
    # that cancel to 1e-13 of their terms)
    Fp = m.residual(U + eps * V, Uo, Uo1, dt, dt_old)
    Fm = m.residual(U - eps * V, Uo, Uo1, dt, dt_old)
    fd, jv = (Fp - Fm) / (2 * eps), J @ V.ravel()
    scale = np.abs(jv).reshape(-1, 3).max(axis=0)
    assert (np.abs(fd - jv).reshape(-1, 3) / scale).max() < 1e-6


def test_linear_and_logarithmic_forms_agree_away_from_the_time_derivative():
    n = 8
    mesh = rectangle_right(0.0, 0.0, ost.BOX, ost.BOX, n, n)
    lin, log = _model(mesh, False), _model(mesh, True)
    lnn, phi = _state(mesh, seed=5)
    lnn *= 0.0
    lnn += np.array([30.0, 29.0])            # constant densities: P1 interpolation of n and exp(P1 of ln n) coincide
    Ulog = np.column_stack([lnn, phi])
    Ulin = np.column_stack([np.exp(lnn), phi])
    # steady state in time (u = u_old = u_old1): the time terms vanish in both forms
    Flog = log.residual(Ulog, Ulog, Ulog, 5e-12, 5e-12)
    Flin = lin.residual(Ulin, Ulin, Ulin, 5e-12, 5e-12)
    scale = np.abs(Flog).reshape(-1, 3).max(axis=0)
    assert (np.abs(Flin - Flog).reshape(-1, 3) / scale).max() < 1e-10
