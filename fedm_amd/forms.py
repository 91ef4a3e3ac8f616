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
# Opaque symbolic nodes.  The LMEA scripts (examples/glow_discharge/fedm-gd.py:205-385) combine
# nodal Functions, unknowns and constants into expressions -- the semi-implicit mean energy, the
# electron thermal velocity, 5/3 of a coefficient, the ion flux through a wall.  The device
# kernels implement those expressions natively, so the facade only has to RECOGNISE them: a Sym
# records the operation and its operands, `fedm_amd.lmea` matches the structure and evaluates the
# numeric parts (``evaluate``), nothing here does arithmetic on fields.
# ---------------------------------------------------------------------------------------
class _Ops:
    def __add__(self, o):
        return Sym("add", self, o)

    def __radd__(self, o):
        return self if (isinstance(o, Real) and o == 0) else Sym("add", o, self)

    def __sub__(self, o):
        return Sym("sub", self, o)

    def __rsub__(self, o):
        return Sym("sub", o, self)

    def __mul__(self, o):
        return Sym("mul", self, o)

    def __rmul__(self, o):
        return Sym("mul", o, self)

    def __truediv__(self, o):
        return Sym("div", self, o)

    def __rtruediv__(self, o):
        return Sym("div", o, self)

    def __neg__(self):
        return Sym("neg", self)

    def __pow__(self, n):
        return Sym("pow", self, n)

    def __abs__(self):
        return Sym("abs", self)


class Sym(_Ops):
    def __init__(self, op, *args):
        self.op, self.args = op, args

    def __float__(self):
        return float(evaluate(self))

    def leaves(self, kind=None):
        out = []
        for a in self.args:
            if isinstance(a, Sym):
                out.extend(a.leaves(kind))
            elif kind is None or isinstance(a, kind):
                out.append(a)
        return out


def evaluate(x, env=None):
    """Numeric value of an expression whose leaves are numbers, Constants, parameter Expressions
    and nodal Functions (-> arrays); ``env`` maps objects (by id) to substitute values.  Unknowns
    and cell-wise operations are not values: TypeError."""
    env = env or {}
    if id(x) in env:
        return env[id(x)]
    if isinstance(x, Real):
        return float(x)
    if isinstance(x, np.ndarray):
        return x
    if isinstance(x, Constant):
        return float(x.value)
    if isinstance(x, (Function, FunctionComponent)):
        return np.asarray(x.vector(), dtype=float)
    if isinstance(x, Expression):
        return x.value()
    if isinstance(x, Sym):
        a = [evaluate(v, env) for v in x.args] if x.op not in ("pow",) else [evaluate(x.args[0], env), x.args[1]]
        ops = {"add": lambda: a[0] + a[1], "sub": lambda: a[0] - a[1], "mul": lambda: a[0] * a[1],
               "div": lambda: a[0] / a[1], "neg": lambda: -a[0], "pow": lambda: a[0] ** a[1],
               "abs": lambda: np.abs(a[0]), "exp": lambda: np.exp(a[0]), "sqrt": lambda: np.sqrt(a[0]),
               "log": lambda: np.log(a[0])}
        if x.op in ops:
            return ops[x.op]()
    raise TypeError(f"cannot evaluate {type(x).__name__}{'(' + x.op + ')' if isinstance(x, Sym) else ''} numerically")


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
            try:
                return np.full(np.shape(x)[:-1], float(self.code))      # Expression('3.0', degree=1)
            except (TypeError, ValueError):
                raise NotImplementedError("C++ expression strings are not compiled here; pass python=") from None
        import inspect
        n_args = len(inspect.signature(self.python).parameters)
        return self.python(np.asarray(x)) if n_args == 1 else self.python(np.asarray(x), self)

    def value(self):
        """Value of a parameter-only expression (``Expression('u_p', u_p=5.0)``,
        ``Expression('U0*(1-exp(-t/1e-9))', U0=..., t=..., python=lambda x, e: ...)``)."""
        if self.python is not None:
            return float(np.asarray(self(np.zeros((1, 2)))).ravel()[0])
        if self.code in self.__dict__:
            return float(self.__dict__[self.code])
        return float(self.code)

    def __mul__(self, o):
        return Sym("mul", self, o)

    def __rmul__(self, o):
        return Sym("mul", o, self)


class Constant(_Ops):
    def __init__(self, value):
        self.value = value

    def __float__(self):
        return float(self.value)


class Function(_Ops):
    """A nodal array on the host (coefficients, post-processing fields).  On a mixed space
    (``Function(ME)``) it is the handle of a device state once a Problem is bound to it; its
    components ``f[i]`` are descriptors."""

    def __init__(self, space=None, values=None):
        self.space = space
        n = space.mesh.num_vertices() if hasattr(space, "mesh") else 0
        self.n_eq = getattr(space, "n_eq", 1)
        self._v = np.zeros(n) if values is None else np.asarray(values, dtype=float)
        self.state = None          # DeviceState once bound (mixed functions)
        self.name = None

    def vector(self):
        return self.state if self.state is not None else self._v

    def rename(self, name, label=None):
        self.name = name

    def __getitem__(self, i):
        return FunctionComponent(self, i)

    def __add__(self, o):
        # a zero Function used as an accumulator of source terms (fedm-streamer.py:164,246)
        if isinstance(o, (RateSum, Rate, Density)):
            return NotImplemented
        if isinstance(o, (Sym, Unknown)) and not np.any(self._v) and o_is_source(o):
            return RateSum.coerce(o)                  # zero Function + densities written with bare unknowns
        return Sym("add", self, o)

    def assign(self, other):
        if self.state is not None:
            return self.state.assign(getattr(other, "state", other))
        if isinstance(other, Expression) and self.space is not None:
            self._v = np.array(other(self.space.mesh.coords), dtype=float)
        elif isinstance(other, (Constant, Real)):
            self._v = np.full_like(self._v, float(other))
        elif isinstance(other, Sym):
            self._v = np.array(evaluate(other), dtype=float) + np.zeros_like(self._v)
        else:
            self._v = np.array(other.vector() if hasattr(other, "vector") else other, dtype=float)


class FunctionComponent(_Ops):
    """``u_old[i]`` of a mixed Function."""

    def __init__(self, function, index):
        self.function, self.index = function, index

    def vector(self):
        raise TypeError("a component of a mixed Function lives on the device; use the assigner")


def interpolate(expr, space):
    """``interpolate(Expression(...), V)`` / ``interpolate(Constant(c), V)``: nodal values."""
    f = Function(space)
    f.assign(expr)
    return f


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
        if self.which == "new":
            return self.device.get_state()
        if self.which == "old":
            return self.device.get_state_old()
        raise RuntimeError("u_old1 cannot be downloaded")


class FunctionAssigner:
    """``FunctionAssigner(receiving_space, assigning_space)``.

    * ``assigner.assign([f_0, ..., f_k], u)`` with ``u`` a device state (or a mixed Function bound
      to one): downloads that state once and fills the host Functions (entries that are None are
      skipped; with ``receiving=None`` nothing moves -- the LFA drivers keep everything on the device);
    * ``rev_assigner.assign(u, [f_0, ..., f_k])``: the reverse, one upload."""

    def __init__(self, *spaces):
        self.spaces = spaces

    def assign(self, receiving, assigning):
        state = lambda x: x if isinstance(x, DeviceState) else getattr(x, "state", None)
        if receiving is None:
            return None
        if isinstance(receiving, (list, tuple)):
            src = state(assigning)
            if src is None:
                if isinstance(assigning, Function):        # mixed Function not bound yet: remember the parts
                    assigning.parts = list(receiving)
                    return None
                raise TypeError("FunctionAssigner: the assigning function is not a device state")
            U = src.array()
            for i, f in enumerate(receiving):
                if f is not None:
                    f.vector()[:] = U[:, i]
            return None
        dst = state(receiving)
        if dst is None:                                    # rev_assigner before the Problem exists
            receiving.parts = list(assigning)
            return None
        dst.assign(np.stack([np.asarray(f.vector(), dtype=float) for f in assigning], axis=1))
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


class Unknown(_Ops):
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
    """coef(|E|) * prod_i n_i^powers[i].  ``bare``: the species whose density entered as the
    unknown itself (``u[i]``, non-logarithmic representation) rather than as ``exp(u[i])``."""

    def __init__(self, coef, powers, bare=()):
        self.coef = TermSum.coerce(coef)
        self.powers = {i: p for i, p in powers.items() if p}
        self.bare = frozenset(i for i in bare if i in self.powers)

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
        if isinstance(x, Unknown):                    # the density itself: non-logarithmic representation
            return RateSum([Rate(1.0, {x.index: 1}, bare=(x.index,))])
        if isinstance(x, Density):
            return RateSum([Rate(1.0, {x.index: 1})])
        if isinstance(x, Sym):                        # products / sums of unknowns and numbers written before
            a = x.args                                # it was known that they form a source term
            if x.op == "add":
                return RateSum.coerce(a[0]) + RateSum.coerce(a[1])
            if x.op == "sub":
                return RateSum.coerce(a[0]) - RateSum.coerce(a[1])
            if x.op == "mul":
                return RateSum.coerce(a[0]) * RateSum.coerce(a[1])
            if x.op == "neg":
                return -RateSum.coerce(a[0])
            if x.op == "div" and isinstance(a[1], (Real, Constant, TermSum)) or (x.op == "div" and isinstance(a[1], Sym) and not a[1].leaves((Unknown, Function))):
                return RateSum.coerce(a[0]) / a[1]
            if x.op == "pow":
                return RateSum.coerce(a[0]) ** a[1]
        raise TypeError(f"cannot use {type(x).__name__} in a source-term expression")

    def __add__(self, o):
        return RateSum(self.terms + RateSum.coerce(o).terms)

    __radd__ = __add__

    def __neg__(self):
        return RateSum([Rate(-t.coef, t.powers, t.bare) for t in self.terms])

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
                out.append(Rate(a.coef * b.coef, p, a.bare | b.bare))
        return RateSum(out)

    __rmul__ = __mul__

    def __truediv__(self, o):
        o = TermSum.coerce(float(o) if isinstance(o, (Constant, Sym)) else o)
        return RateSum([Rate(t.coef / o, t.powers, t.bare) for t in self.terms])

    def __pow__(self, n):
        n = int(n)
        out = RateSum.coerce(1.0)
        for _ in range(n):
            out = out * self
        return out


def o_is_source(x):
    """True when a symbolic expression consists of unknowns and numbers only (a source term of the
    non-logarithmic representation)."""
    try:
        RateSum.coerce(x)
        return True
    except TypeError:
        return False


def _rate_ops(cls):
    """Let Density take part in the RateSum algebra."""
    for name in ("__add__", "__radd__", "__sub__", "__rsub__", "__mul__", "__rmul__",
                 "__truediv__", "__neg__", "__pow__"):
        def op(self, *a, _n=name):
            try:
                if any(isinstance(x, (Function, FunctionComponent, Sym)) for x in a):
                    raise TypeError("nodal operand")
                return getattr(RateSum([Rate(1.0, {self.index: 1})]), _n)(*a)
            except TypeError:            # an operand that is no function of |E|: nodal coefficients (LMEA)
                sym_op = {"__add__": "add", "__radd__": "add", "__sub__": "sub", "__rsub__": "rsub",
                          "__mul__": "mul", "__rmul__": "mul", "__truediv__": "div", "__pow__": "pow"}[_n]
                if sym_op == "rsub":
                    return Sym("sub", a[0], self)
                return Sym(sym_op, self, *a)
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
    if isinstance(o, (RateSum, Density, Unknown)):
        return RateSum.coerce(self) * RateSum.coerce(o)
    if isinstance(o, Sym) and o_is_source(o):
        return RateSum.coerce(self) * RateSum.coerce(o)
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
    if isinstance(u, Function):
        return Sym("grad", u)                    # of a nodal Function: cell-wise constant (project())
    if not isinstance(u, Unknown):
        raise NotImplementedError("grad() of the potential unknown or of a nodal Function")
    return GradOf(u)


def inner(a, b):
    from .functions import FluxDesc
    if isinstance(a, GradOf) and isinstance(b, GradOf) and a.unknown is b.unknown:
        return FieldSquared(a.unknown.index)
    if isinstance(a, FluxDesc) and isinstance(b, FacetNormal):
        return Sym("normal_flux", a)             # Gamma . n on a wall (fedm-gd.py:351)
    if isinstance(a, FluxDesc) and isinstance(b, GradOf):
        return Sym("flux_dot_field", a, b)       # Gamma . E: Joule heating (fedm-gd.py:359)
    if isinstance(a, Sym) and isinstance(b, Sym):
        return Sym("dot", a, b)
    raise NotImplementedError("inner()/dot() of these operands is not part of the device models")


dot = inner


def sqrt(x):
    if isinstance(x, FieldSquared):
        return TermSum.field()                   # E_m
    if isinstance(x, TermSum):
        return x ** 0.5
    if isinstance(x, (Sym, Function)):
        return Sym("sqrt", x)
    return math.sqrt(x)


def exp(x):
    if isinstance(x, Unknown):
        return Density(x.index)
    if isinstance(x, TermSum):
        return x.exp()
    if isinstance(x, (Sym, Function, FunctionComponent)):
        return Sym("exp", x)
    return math.exp(x)


def project(expr, space=None, solver_type=None):
    """L2 projection onto P1 of ``c * sqrt(dot(grad(f), grad(f)))``-type expressions of nodal
    Functions (fedm-gd.py:309,432: the reduced electric field): the gradient of a P1 Function is
    constant per cell; consistent mass matrix, factorised once per mesh (host, post-processing
    size -- the device-resident pipeline is `fedm_gd_prep_step`)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    funcs = [f for f in expr.leaves(Function)] if isinstance(expr, Sym) else []
    if not funcs:
        raise NotImplementedError("project() of an expression without nodal Functions")
    mesh = funcs[0].space.mesh
    x = mesh.coords[mesh.cells]
    d1, d2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
    det = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]

    def cellwise(e):
        if isinstance(e, Sym) and e.op == "grad":
            P = np.asarray(e.args[0].vector(), dtype=float)[mesh.cells]
            gx = (P[:, 0] * (x[:, 1, 1] - x[:, 2, 1]) + P[:, 1] * (x[:, 2, 1] - x[:, 0, 1])
                  + P[:, 2] * (x[:, 0, 1] - x[:, 1, 1])) / det
            gy = (P[:, 0] * (x[:, 2, 0] - x[:, 1, 0]) + P[:, 1] * (x[:, 0, 0] - x[:, 2, 0])
                  + P[:, 2] * (x[:, 1, 0] - x[:, 0, 0])) / det
            return np.stack([gx, gy], axis=1)
        if isinstance(e, Sym) and e.op == "dot":
            return np.einsum("cd,cd->c", cellwise(e.args[0]), cellwise(e.args[1]))
        if isinstance(e, Sym):
            a = [cellwise(v) for v in e.args]
            return {"neg": lambda: -a[0], "sqrt": lambda: np.sqrt(a[0]), "mul": lambda: a[0] * a[1],
                    "div": lambda: a[0] / a[1], "add": lambda: a[0] + a[1], "sub": lambda: a[0] - a[1]}[e.op]()
        if isinstance(e, Function):
            raise NotImplementedError("project(): nodal values outside grad() are not supported")
        return evaluate(e)
    f = cellwise(expr)
    lu = getattr(mesh, "_p1_mass_lu", None)
    n = mesh.num_vertices()
    if lu is None:
        vals = np.abs(det)[:, None, None] * ((np.ones((3, 3)) + np.eye(3)) / 24.0)[None]
        c = mesh.cells.astype(np.int64)
        rows = np.broadcast_to(c[:, :, None], vals.shape).ravel()
        cols = np.broadcast_to(c[:, None, :], vals.shape).ravel()
        lu = mesh._p1_mass_lu = spla.splu(sp.coo_matrix((vals.ravel(), (rows, cols)), shape=(n, n)).tocsc())
    rhs = np.bincount(mesh.cells.ravel(), weights=np.repeat(f * np.abs(det) / 6.0, 3), minlength=n)
    return Function(space or funcs[0].space, values=lu.solve(rhs))


def action(form, u):
    """``F = action(F, u_new)``: the form evaluated at u_new; the facade notes which mixed Function
    that is (it becomes the handle of the device's current state in Problem())."""
    if isinstance(u, Function):
        try:
            form.u_new = u
        except AttributeError:
            pass
    return form


def derivative(form, u, du=None):
    return ("jacobian of", form)
