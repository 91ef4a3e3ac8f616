"""Time-step controllers and the accept/reject loop (oracle; test infra).

Restates fedm/functions.py:915-951 (``adaptive_timestep`` and the PI.3.4 /
H211b variants) and fedm/functions.py:1037-1127 (``adaptive_solver``) together
with the caller's bookkeeping at
examples/streamer_discharge/fedm-streamer.py:304-340.
"""
import numpy as np

DOLFIN_EPS = 3.0e-16


def adaptive_timestep(dt, error, tol=1e-4, dt_min=1e-13, dt_max=1e-9):
    dt *= ((error[1] / error[0]) ** 0.075 * (tol / error[0]) ** 0.175
           * (error[1] ** 2 / (error[0] * error[2])) ** 0.01)
    return max(min(dt, dt_max), dt_min)


def adaptive_timestep_PI34(dt, error, tol=1e-4, dt_min=1e-13, dt_max=1e-9):
    dt *= (0.8 * tol / error[0]) ** (0.3 / 3) * (0.8 * error[1] / error[0]) ** (0.4 / 3)
    return max(min(dt, dt_max), dt_min)


def adaptive_timestep_H211b(dt, dt_old, error, tol=1e-4, dt_min=1e-13, dt_max=1e-9):
    dt *= ((0.8 * tol / error[0]) ** (1 / 12) * (0.8 * tol / error[1]) ** (1 / 12)
           * (dt / dt_old) ** (-1 / 4))
    return max(min(dt, dt_max), dt_min)


def field_error(new, old):
    """df.norm(new - old + DOLFIN_EPS) / df.norm(old + DOLFIN_EPS), functions.py:1062-1064."""
    return float(np.linalg.norm(new - old + DOLFIN_EPS) / np.linalg.norm(old + DOLFIN_EPS))


class StepState:
    """dt / dt_old / error history carried between steps."""

    def __init__(self, dt, dt_old=1e30, n_error=1):
        self.dt = dt
        self.dt_old = dt_old
        self.error = [0.0] * n_error
        self.max_error = [1, 1, 1]
        self.log = []                 # rows (error, dt_old, dt), one per attempt


def adaptive_solve(solve, U, U_old, t, st, ttol, dt_min, error_component):
    """One accepted step.  ``solve(U, dt, dt_old)`` advances U in place or raises.

    Iterative form of the reference's recursion (fedm/functions.py:1037-1127)."""
    while True:
        try:
            t += st.dt
            solve(U, st.dt, st.dt_old)
            st.error[0] = field_error(U[:, error_component], U_old[:, error_component])
            st.log.append((st.error[0], st.dt_old, st.dt))
            st.max_error[0] = max(st.error)
            if st.error[0] >= ttol:
                raise _ErrorGreaterThanTTOL()
            return t
        except Exception as exc:               # the reference catches everything (:1080)
            t -= st.dt
            if isinstance(exc, _ErrorGreaterThanTTOL):
                st.dt *= 0.5 * ttol / st.max_error[0]
            else:
                st.dt *= 0.5
            if st.dt < dt_min:
                raise SystemExit("Minimum time-step size reached, program is terminating.")
            U[:] = U_old


class _ErrorGreaterThanTTOL(Exception):
    pass
