"""Light stand-ins for the DOLFIN/UFL objects a FEDM script passes around.

DOLFIN cannot exist on the GPU box, so the arguments that are UFL objects in the
reference (``u``, ``v``, ``dx``, ``ds``, ``Gamma``, ``f``, ``E``, ``normal`` ...) are small
descriptor objects here.  They support exactly the algebra the example scripts use
(examples/streamer_discharge/fedm-streamer.py:220-271): ``exp(u[i])``, products with
coefficients that depend on ``E_m``, sums of such terms, ``-grad(u[k])``,
``sqrt(inner(E, E))``.  ``fedm_amd.functions`` turns them into the model descriptor of
the device kernels; nothing here does arithmetic on fields.
"""
import math
from numbers import Real

import numpy as np

from .termsum import TermSum

# DOLFIN's global parameter dictionary (fedm-streamer.py:19-23); only the quadrature degree
# is meaningful on the device path.
parameters = {"form_compiler": {"optimize": True, "cpp_optimize": True, "quadrature_degree": -1},
              "std_out_all_processes": False, "krylov_solver": {"nonzero_initial_guess": True}}
DOLFIN_EPS = 3.0e-16
pi = math.pi


def near(a, b, eps=DOLFIN_EPS):
    return abs(a - b) <= eps


# ---------------------------------------------------------------------------------------
# scalars, nodal functions, device state handles
# ---------------------------------------------------------------------------------------
class Expression:
    """``Expression("time_step", time_step=..., degree=0)`` and friends: a bag of named
    parameters (``.time_step``, ``.t`` ...), as FEDM scripts use it for scalars.  Spatial
    expressions are given as ``python=callable(x)`` (no C++ JIT here)."""

    def __init__(self, code=None, degree=0, python=None, **params):
        self.code = code
        self.degree = degree
        self.python = python
        for k, v in params.items():
            setattr(self, k, v)

    def __call__(self, x):
        if self.python is None:
            raise NotImplementedError("C++ expression strings are not compiled here; pass python=")
        return self.python(np.asarray(x))


class Constant:
    def __init__(self, value):
        self.value = value

    def __float__(self):
        return float(self.value)


class Function:
    """A nodal array on the host (coefficients, post-processing fields)."""

    def __init__(self, space=None, values=None):
        self.space = space
        n = space.mesh.num_vertices() if hasattr(space, "mesh") else 0
        self._v = np.zeros(n) if values is None else np.asarray(values, dtype=float)

    def vector(self):
        return self._v

    def assign(self, other):
        self._v = np.array(other.vector() if hasattr(other, "vector") else other, dtype=float)


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
        elif pair == ("old1", "old") or pair == ("old", "new"):
            # u_old1.assign(u_old); u_old.assign(u_new) always come as a pair
            # (fedm-streamer.py:306-307): the rotation happens on the second call
            if pair == ("old", "new"):
                self.device.shift_state()
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


# ---------------------------------------------------------------------------------------
# spaces, unknowns, measures
# ---------------------------------------------------------------------------------------
class FiniteElement:
    def __init__(self, family="Lagrange", cell=None, degree=1):
        if degree != 1 or family not in ("Lagrange", "P", "CG"):
            raise NotImplementedError("the device path implements P1 Lagrange elements")
        self.family, self.degree = family, degree


class MixedElement(list):
    pass


class SubSpace:
    def __init__(self, space, component):
        self.space, self.component = space, component


class FunctionSpace:
    def __init__(self, mesh, element, degree=None):
        self.mesh = mesh
        self.n_eq = len(element) if isinstance(element, (list, MixedElement)) else 1

    def sub(self, i):
        return SubSpace(self, i)


class Unknown:
    """Component ``i`` of the mixed trial function."""

    def __init__(self, space, index):
        self.space, self.index = space, index


class Test:
    def __init__(self, space, index):
        self.space, self.index = space, index


def TrialFunction(space):
    return [Unknown(space, i) for i in range(space.n_eq)] if space.n_eq > 1 else Unknown(space, 0)


def TestFunctions(space):
    return [Test(space, i) for i in range(space.n_eq)]


def TestFunction(space):
    return Test(space, 0)


class Measure:
    def __init__(self, kind, domain=None, subdomain_data=None, tag=None):
        self.kind, self.domain, self.subdomain_data, self.tag = kind, domain, subdomain_data, tag

    def __call__(self, tag):
        return Measure(self.kind, self.domain, self.subdomain_data, tag)


dx = Measure("dx")
ds = Measure("ds")


class FacetNormal:
    def __init__(self, mesh):
        self.mesh = mesh


class DirichletBC:
    """``DirichletBC(ME.sub(k), value, inside)`` with ``inside(x, on_boundary)`` as in
    fedm-streamer.py:189-200,233."""

    def __init__(self, space, value, inside):
        self.component = space.component if isinstance(space, SubSpace) else 0
        self.space = space.space if isinstance(space, SubSpace) else space
        self.value, self.inside = value, inside

    def rows(self, mesh, n_eq):
        cell, local = mesh.exterior_facets()
        ends = np.array([[1, 2], [0, 2], [0, 1]])[local]
        verts = np.unique(mesh.cells[cell[:, None], ends])
        hit = [v for v in verts if self.inside(mesh.coords[v], True)]
        dofs = np.array(hit, dtype=np.int64) * n_eq + self.component
        val = self.value
        vals = np.array([float(val(mesh.coords[v])) if callable(val) else float(val) for v in hit])
        return dofs, vals


# ---------------------------------------------------------------------------------------
# the little algebra of source terms:  sum_k  coef_k(E_m) * prod_i exp(u_i)^p_ki
# ---------------------------------------------------------------------------------------
class Rate:
    """coef(|E|) * prod_i n_i^powers[i]"""

    def __init__(self, coef, powers):
        self.coef = TermSum.coerce(coef)
        self.powers = {i: p for i, p in powers.items() if p}

    def exp(self):
        raise ValueError("exp() of a density product is not a supported expression")


class RateSum:
    def __init__(self, terms=()):
        self.terms = list(terms)

    @staticmethod
    def coerce(x):
        if isinstance(x, RateSum):
            return x
        if isinstance(x, Rate):
            return RateSum([x])
        if isinstance(x, (Real, TermSum)):
            ts = TermSum.coerce(x)
            return RateSum([Rate(ts, {})] if ts.terms else [])
        if isinstance(x, Function) and not np.any(x.vector()):
            return RateSum([])                  # a zero Function used as accumulator (:164,246)
        if isinstance(x, Constant):
            return RateSum.coerce(float(x))
        raise TypeError(f"cannot use {type(x).__name__} in a source-term expression")

    def __add__(self, o):
        return RateSum(self.terms + RateSum.coerce(o).terms)

    __radd__ = __add__

    def __neg__(self):
        return RateSum([Rate(-t.coef, t.powers) for t in self.terms])

    def __sub__(self, o):
        return self + (-RateSum.coerce(o))

    def __rsub__(self, o):
        return RateSum.coerce(o) - self

    def __mul__(self, o):
        o = RateSum.coerce(o)
        out = []
        for a in self.terms:
            for b in o.terms:
                p = dict(a.powers)
                for i, e in b.powers.items():
                    p[i] = p.get(i, 0) + e
                out.append(Rate(a.coef * b.coef, p))
        return RateSum(out)

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = TermSum.coerce(float(o) if isinstance(o, Constant) else o)
        return RateSum([Rate(t.coef / o, t.powers) for t in self.terms])

    def __pow__(self, n):
        n = int(n)
        out = RateSum.coerce(1.0)
        for _ in range(n):
            out = out * self
        return out


def _rate_ops(cls):
    """Let Density take part in the RateSum algebra."""
    for name in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__",
                 "__truediv__", "__neg__", "__pow__"):
        def op(self, *a, _n=name):
            return getattr(RateSum([Rate(1.0, {self.index: 1})]), _n)(*a)
        setattr(cls, name, op)
    return cls


@_rate_ops
class Density:
    """exp(u_i)"""

    def __init__(self, index):
        self.index = index


# TermSum * RateSum must dispatch to RateSum
TermSum.__array_priority__ = 1000
_ts_mul = TermSum.__mul__


def _ts_mul_dispatch(self, o):
    if isinstance(o, (RateSum, Density)):
        return RateSum.coerce(self) * (o if isinstance(o, RateSum) else RateSum([Rate(1.0, {o.index: 1})]))
    return _ts_mul(self, o)


TermSum.__mul__ = _ts_mul_dispatch
TermSum.__rmul__ = _ts_mul_dispatch


class GradOf:
    def __init__(self, unknown, sign=1.0):
        self.unknown, self.sign = unknown, sign

    def __neg__(self):
        return GradOf(self.unknown, -self.sign)


class FieldSquared:
    def __init__(self, phi_index):
        self.phi_index = phi_index


def grad(u):
    if not isinstance(u, Unknown):
        raise NotImplementedError("grad() of the potential unknown only")
    return GradOf(u)


def inner(a, b):
    if isinstance(a, GradOf) and isinstance(b, GradOf) and a.unknown is b.unknown:
        return FieldSquared(a.unknown.index)
    raise NotImplementedError("inner() is defined for the electric field with itself")


dot = inner


def sqrt(x):
    if isinstance(x, FieldSquared):
        return TermSum.field()                   # E_m
    if isinstance(x, TermSum):
        return x ** 0.5
    return math.sqrt(x)


def exp(x):
    if isinstance(x, Unknown):
        return Density(x.index)
    if isinstance(x, TermSum):
        return x.exp()
    return math.exp(x)


def action(form, u):
    return form


def derivative(form, u, du=None):
    return ("jacobian of", form)
