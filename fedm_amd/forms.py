"""Light stand-ins for the DOLFIN objects a FEDM script passes around.

Only what the hot path's callers touch: the time-step ``Expression``
(``dt.time_step``), the mixed ``Function`` handles (``u_new``, ``u_old``,
``u_old1``: here views of the state vectors resident in HBM) and the
``FunctionAssigner`` (a no-op: per-field views are cut from the mixed state on
demand).
"""
import math

import numpy as np


class Expression:
    """``Expression("time_step", time_step=..., degree=0)`` and friends: a bag of
    named parameters (``.time_step``, ``.t`` ...), as FEDM scripts use it for scalars."""

    def __init__(self, code=None, degree=0, **params):
        self.code = code
        self.degree = degree
        for k, v in params.items():
            setattr(self, k, v)


class Function:
    """A nodal array on the host (coefficients, post-processing fields)."""

    def __init__(self, space=None, values=None):
        self.space = space
        n = getattr(space, "dim", None)
        self._v = np.zeros(n() if callable(n) else (n or 0)) if values is None else np.asarray(values, float)

    def vector(self):
        return self._v


TrialFunction = TestFunction = Function


class DeviceState:
    """u_new / u_old / u_old1 of a :class:`fedm_amd.device.DeviceProblem`."""

    def __init__(self, device, which):
        self.device, self.which = device, which

    def vector(self):
        return self

    def assign(self, other):
        pair = (self.which, getattr(other, "which", None))
        if pair == ("new", "old"):
            self.device.reset_state()                      # functions.py:1103
        elif pair == ("old", "new") or pair == ("old1", "old"):
            raise RuntimeError("use DeviceProblem.shift_state(): u_old1<-u_old, u_old<-u_new")
        else:
            self.device.set_state(**{"u_" + self.which: np.asarray(other)})

    def array(self):
        if self.which != "new":
            raise RuntimeError("only the current state can be downloaded")
        return self.device.get_state()


class FunctionAssigner:
    """``assigner.assign(var_list, u)``: per-field views are cut on demand; no copy."""

    def __init__(self, *spaces):
        self.spaces = spaces

    def assign(self, receiving, assigning):
        return None


def exp(x):
    return x.exp() if hasattr(x, "exp") else math.exp(x)
