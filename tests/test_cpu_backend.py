"""oracle/cpu/fedm_cpu.c (the C/OpenMP CPU baseline that bench.py times) against the numpy oracle.

The C backend restates the device library's algorithm (coloured element loop, block CSR, Newton,
flexible GMRES with the field split); the numpy oracle (oracle/forms.py, oracle/newton.py) restates
the reference's forms with a direct solve.  Two independent CPU statements of the same path must
agree: element tensors to rounding, time steps to the Newton tolerance."""
import warnings

import numpy as np
import pytest

from oracle import cpu_backend as cb
from oracle import streamer as ost
from oracle.mesh import graded_axis, rectangle_right


@pytest.fixture(scope="module", autouse=True)
def _built():
    cb.build()


def _perturbed(mesh, model):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        U0 = ost.initial_state(model)
    x, y = mesh.coords[:, 0] / ost.BOX, mesh.coords[:, 1] / ost.BOX
    rng = np.random.default_rng(7)
    U = U0.copy()
    U[:, 0] += 0.05 * np.sin(7.0 * x) * np.cos(5.0 * y)
    U[:, 1] += 0.05 * np.cos(3.0 * x) * np.sin(4.0 * y)
    U[:, 2] += 30.0 * np.sin(3.0 * x) * np.sin(np.pi * y)
    return U, U0 + 0.01 * rng.standard_normal(U0.shape), U0 + 0.02 * rng.standard_normal(U0.shape)


def test_residual_and_jacobian_match_the_numpy_oracle():
    n = 20
    mesh = rectangle_right(0.0, 0.0, ost.BOX, ost.BOX, n, n, xs=graded_axis(ost.BOX, n, 2.0))
    model = ost.build(mesh)
    U, Uo, Uo1 = _perturbed(mesh, model)
    prob = cb.CpuProblem(model)
    prob.set_state(U, Uo, Uo1)
    F, J = prob.residual_jacobian(5e-12, 4e-12)
    F_only = prob.residual(5e-12, 4e-12)
    Fo, Jo = model.residual_jacobian(U, Uo, Uo1, 5e-12, 4e-12)
    prob.close()
    assert np.abs(F - Fo).max() <= 1e-11 * np.abs(Fo).max()
    assert np.array_equal(F, F_only)
    d = abs(J - Jo)
    rowmax = abs(Jo).max(axis=1).toarray().ravel()
    assert (d.max(axis=1).toarray().ravel() <= 1e-11 * rowmax + 1e-300).all()


@pytest.mark.parametrize("n,levels", [(24, 1), (64, 2)])
def test_time_steps_match_the_direct_solve_oracle(n, levels):
    """Three accepted BDF2 steps: Newton (rtol 1e-4) + flexible GMRES with the field split in C
    against Newton + equilibrated sparse LU in numpy.  n = 64 runs the multigrid V-cycle with a
    sparse level; n = 24 solves the potential block with the dense inverse alone."""
    prob, mesh = cb.streamer_problem(n, 2.0)
    U, st, t, stats = cb.run_streamer(prob, mesh, 3)
    assert len(prob.levels) == levels
    prob.close()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Uo, sto, to, _ = ost.run(mesh=mesh, max_steps=3)
    assert t == pytest.approx(to, rel=1e-12)
    assert (np.abs(U - Uo).max(axis=0) <= 1e-6 * np.abs(Uo).max(axis=0)).all()
    assert np.allclose(np.array(st.log), np.array(sto.log), rtol=2e-4)
    assert 6 <= stats["newton"] <= 9 and 0 < stats["linear"] <= 60


def test_bench_record_has_the_contract_fields():
    rec = cb.bench(32, 4.0, 2, 2)
    assert rec["kind"] == "port" and rec["cores"] == 2 and rec["unit"] == "DOF-updates/s"
    assert rec["value"] > 0 and rec["dofs"] == 33 * 33 * 3
    assert "not FEniCS" in rec["sample"]
