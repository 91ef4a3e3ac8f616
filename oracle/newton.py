"""Newton iteration with PETSc-SNES "newtonls/basic" stopping rules (oracle; test infra).

Restates what ``nonlinear_solver.solve(problem, u_new.vector())`` does at
fedm/functions.py:1047 (and examples/time_of_flight/fedm-tof.py:152).  The
solver itself is un-vendored third-party code (DOLFIN 2019.1.0
``PETScSNESSolver`` over PETSc SNES); its published algorithm is: full Newton
steps (line search "basic"), converged when ``|F| < atol`` or
``|F| <= rtol*|F0|`` or ``|dx| < stol*|x|``; failure raises (DOLFIN's
``error_on_nonconvergence``), which ``adaptive_solver`` catches at
fedm/functions.py:1080.  DOLFIN defaults atol=1e-10, stol=1e-16; the scripts
set rtol and max_it (fedm-streamer.py:27-28,295-297; fedm-tof.py:24-25).
The sparse direct solve (``splu``) stands in for MUMPS; like MUMPS' default
(ICNTL(8)=77) the rows are equilibrated first -- the log-variable rows scale
with exp(u) over >60 orders of magnitude (time-of-flight floor at 3e-16 next to
a 1e13 pulse) and an unscaled LU loses the update there.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


def direct_solve(J, b):
    """Row-equilibrated sparse LU solve of J x = b."""
    J = J.tocsr()
    rmax = np.maximum.reduceat(np.abs(J.data), J.indptr[:-1])
    scale = 1.0 / np.where(rmax > 0.0, rmax, 1.0)
    return spla.splu((sp.diags(scale) @ J).tocsc()).solve(scale * b)


class NewtonDiverged(RuntimeError):
    pass


def newton_solve(model, U, Uold, Uold1, dt, dt_old, rtol, max_it,
                 atol=1e-10, stol=1e-16, linear_solve=None, report=None):
    """Solve F(U)=0 in place.  Returns (iterations, converged)."""
    shape = U.shape
    fnorm0 = None
    its = 0
    history = []
    while True:
        F, J = model.residual_jacobian(U, Uold, Uold1, dt, dt_old)
        fnorm = float(np.linalg.norm(F))
        history.append(fnorm)
        if not np.isfinite(fnorm):
            raise NewtonDiverged("residual norm is not finite")
        if its == 0:
            fnorm0 = fnorm
            if fnorm < atol:
                break
        else:
            if fnorm < atol or fnorm <= rtol * fnorm0:
                break
            if snorm < stol * float(np.linalg.norm(U)):
                break
        if its >= max_it:
            raise NewtonDiverged(f"no convergence in {max_it} Newton iterations")
        if linear_solve is None:
            delta = direct_solve(J, -F)
        else:
            delta = linear_solve(J, -F)
        snorm = float(np.linalg.norm(delta))
        U += delta.reshape(shape)
        its += 1
    if report is not None:
        report["residual_history"] = history
    return its, True
