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
_CPP_FUNCTIONS = {"exp": np.exp, "log": np.log, "sqrt": np.sqrt, "pow": np.power, "sin": np.sin, "cos": np.cos,
                  "tan": np.tan, "fabs": np.abs, "abs": np.abs, "tanh": np.tanh, "atan": np.arctan}
_CPP_CONSTANTS = {"pi": math.pi, "DOLFIN_EPS": DOLFIN_EPS, "DOLFIN_PI": math.pi}


def _parse_cpp_expression(code):
    """The checked syntax tree of a C++ expression string of the arithmetic subset (see
    compile_cpp_expression); NotImplementedError for anything else."""
    import ast
    text = code.replace("std::", "").strip()
    try:
        tree = ast.parse(text, mode="eval").body
    except SyntaxError as exc:
        raise NotImplementedError(f"Expression string outside the supported arithmetic subset: {code!r}") from exc

    def integer_valued(node):
        """An operand that a C++ compiler types `int`: integer literals and arithmetic on them."""
        if isinstance(node, ast.Constant):
            return isinstance(node.value, int) and not isinstance(node.value, bool)
        if isinstance(node, ast.UnaryOp):
            return integer_valued(node.operand)
        if isinstance(node, ast.BinOp):
            return integer_valued(node.left) and integer_valued(node.right)
        return False

    def check(node):
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Div) and integer_valued(node.left) \
                and integer_valued(node.right):
            # DOLFIN's JIT divides two ints the C++ way (1/2 == 0, 3/2 == 1); evaluated here it would
            # silently be 0.5 and 1.5 -- a different source term or initial condition
            raise NotImplementedError(f"integer division in Expression string {code!r}: C++ truncates it; "
                                      "write one operand as a floating-point literal (1.0/2)")
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)) and not isinstance(node.value, bool):
            return
        if isinstance(node, ast.Name):
            return
        if isinstance(node, ast.Subscript) and isinstance(node.value, ast.Name) and node.value.id == "x":
            idx = node.slice
            if isinstance(idx, ast.Constant) and isinstance(idx.value, int):
                return
        if isinstance(node, ast.BinOp) and isinstance(node.op, (ast.Add, ast.Sub, ast.Mult, ast.Div)):
            check(node.left)
            check(node.right)
            return
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            check(node.operand)
            return
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and node.func.id in _CPP_FUNCTIONS \
                and not node.keywords and len(node.args) == (2 if node.func.id == "pow" else 1):
            for a in node.args:
                check(a)
            return
        raise NotImplementedError(f"Expression string outside the supported arithmetic subset: {code!r}")
    check(tree)
    return tree


def expression_program(expr):
    """The postfix program of a spatial Expression for the device (``fedm_ext_source_program``):
    ``(ops int32[n][2], consts float64[], parameter names)``.  Names that the Expression carries as
    attributes are parameters (their values are read when the program runs, so ``f.t = t`` works),
    ``pi`` / ``DOLFIN_EPS`` otherwise constants."""
    import ast
    from ._lib import EXPR_OPS as OP, EXPR_MAX_OPS, EXPR_MAX_PARAMS, EXPR_STACK
    if expr.python is not None or not isinstance(expr.code, str):
        raise NotImplementedError("only Expression strings have a device program")
    tree = _parse_cpp_expression(expr.code)
    ops, consts, params = [], [], []

    def const(v):
        v = float(v)
        if v not in consts:
            consts.append(v)
        ops.append((OP["const"], consts.index(v)))

    def emit(node):
        if isinstance(node, ast.Constant):
            const(node.value)
        elif isinstance(node, ast.Name):
            if hasattr(expr, node.id) and node.id not in ("code", "degree", "python"):
                if node.id not in params:
                    params.append(node.id)
                ops.append((OP["param"], params.index(node.id)))
            elif node.id in _CPP_CONSTANTS:
                const(_CPP_CONSTANTS[node.id])
            else:
                raise NameError(f"Expression parameter '{node.id}' is not set")
        elif isinstance(node, ast.Subscript):
            if node.slice.value not in (0, 1):
                raise NotImplementedError("x[i] with i = 0, 1")
            ops.append((OP["x"], int(node.slice.value)))
        elif isinstance(node, ast.BinOp):
            emit(node.left)
            emit(node.right)
            ops.append((OP[{ast.Add: "add", ast.Sub: "sub", ast.Mult: "mul", ast.Div: "div"}[type(node.op)]], 0))
        elif isinstance(node, ast.UnaryOp):
            emit(node.operand)
            if isinstance(node.op, ast.USub):
                ops.append((OP["neg"], 0))
        else:
            for a in node.args:
                emit(a)
            ops.append((OP[node.func.id], 0))
    emit(tree)
    depth = peak = 0
    for op, _ in ops:
        depth += 1 if op <= OP["param"] else -1 if op <= OP["pow"] else 0
        peak = max(peak, depth)
    if len(ops) > EXPR_MAX_OPS or len(params) > EXPR_MAX_PARAMS or peak > EXPR_STACK:
        raise NotImplementedError("Expression too long for the device evaluator")
    return np.array(ops, dtype=np.int32).reshape(-1, 2), np.array(consts, dtype=np.float64), params


def run_expression_program(ops, consts, params, x):
    """Host interpreter of a device expression program (for tests): the value at points x[..., 2]."""
    from ._lib import EXPR_OPS as OP
    unary = {OP["neg"]: np.negative, OP["exp"]: np.exp, OP["log"]: np.log, OP["sqrt"]: np.sqrt, OP["sin"]: np.sin,
             OP["cos"]: np.cos, OP["tan"]: np.tan, OP["fabs"]: np.abs, OP["tanh"]: np.tanh, OP["atan"]: np.arctan}
    binary = {OP["add"]: np.add, OP["sub"]: np.subtract, OP["mul"]: np.multiply, OP["div"]: np.divide,
              OP["pow"]: np.power}
    x = np.asarray(x, dtype=float)
    stack = []
    for op, arg in np.asarray(ops).reshape(-1, 2):
        if op == OP["const"]:
            stack.append(float(consts[arg]))        # (scalars stay scalars: numpy's x**2.0 fast path as in the AST walk)
        elif op == OP["x"]:
            stack.append(x[..., arg])
        elif op == OP["param"]:
            stack.append(float(params[arg]))
        elif op in binary:
            b = stack.pop()
            stack[-1] = binary[op](stack[-1], b)
        else:
            stack[-1] = unary[op](stack[-1])
    (value,) = stack
    return value + np.zeros(x.shape[:-1])


def compile_cpp_expression(code):
    """A DOLFIN C++ expression string (``'std::log(exp(-(pow(x[1]-w*t, 2)+pow(x[0], 2))/(4.0*D*t))...'``,
    examples/time_of_flight/fedm-tof.py:107,116,120) as a function ``f(x, owner)`` of the point array
    ``x[..., dim]`` and the object that carries the named parameters.  The arithmetic subset of the
    language is parsed with Python's ``ast`` and walked by hand (never ``eval``): numbers, ``x[i]``,
    parameters, ``+ - * /``, unary minus and the calls in ``_CPP_FUNCTIONS``."""
    import ast
    tree = _parse_cpp_expression(code)

    def run(node, x, owner):
        if isinstance(node, ast.Constant):
            return float(node.value)
        if isinstance(node, ast.Name):
            if hasattr(owner, node.id) and node.id not in ("code", "degree", "python"):
                value = getattr(owner, node.id)
                if isinstance(value, Function):
                    # a Function as parameter (fedm-gd.py:258: Expression('std::log(a) + b', a=mean_energy,
                    # b=u_oldV[...])): its nodal values, i.e. the expression is evaluated at the vertices
                    nodal = np.asarray(value.vector(), dtype=float)
                    if nodal.shape != np.shape(x)[:-1]:
                        raise NotImplementedError("an Expression with Function parameters is evaluated at the "
                                                  "mesh vertices only (interpolate)")
                    return nodal
                return float(value)
            if node.id in _CPP_CONSTANTS:
                return _CPP_CONSTANTS[node.id]
            raise NameError(f"Expression parameter '{node.id}' is not set")
        if isinstance(node, ast.Subscript):
            return x[..., node.slice.value]
        if isinstance(node, ast.BinOp):
            a, b = run(node.left, x, owner), run(node.right, x, owner)
            return a + b if isinstance(node.op, ast.Add) else a - b if isinstance(node.op, ast.Sub) \
                else a * b if isinstance(node.op, ast.Mult) else a / b
        if isinstance(node, ast.UnaryOp):
            v = run(node.operand, x, owner)
            return -v if isinstance(node.op, ast.USub) else v
        return _CPP_FUNCTIONS[node.func.id](*[run(a, x, owner) for a in node.args])
    return lambda x, owner: run(tree, np.asarray(x, dtype=float), owner) + np.zeros(np.shape(x)[:-1])


class Expression:
    """``Expression("time_step", time_step=..., degree=0)`` and friends: a bag of named
    parameters (``.time_step``, ``.t`` ...), as FEDM scripts use it for scalars.  Spatial
    expressions: ``python=callable(x)`` or a C++ string of the arithmetic subset that
    :func:`compile_cpp_expression` understands (no JIT here)."""

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
                pass
            compiled = self.__dict__.get("_compiled")
            if compiled is None:
                compiled = self.__dict__["_compiled"] = compile_cpp_expression(self.code)
            return compiled(x, self)
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
        try:
            return float(self.code)
        except (TypeError, ValueError):      # an arithmetic string of the parameters: 'U0*(1-exp(-t/1e-9))'
            return float(np.asarray(self(np.zeros((1, 2)))).ravel()[0])

    def __mul__(self, o):
        return Sym("mul", self, o)

    def __rmul__(self, o):
        return Sym("mul", o, self)


class Constant(_Ops):
    """``Constant(c)`` or a vector constant ``Constant(('0', w))`` (fedm-tof.py:111)."""

    def __init__(self, value):
        self.value = tuple(float(v) for v in value) if isinstance(value, (tuple, list)) else value

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
        elif isinstance(other, Constant) and isinstance(other.value, tuple):
            self._v = np.tile(np.asarray(other.value, dtype=float), (self.space.mesh.num_vertices(), 1))
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
        if isinstance(element, str) and (element not in ("P", "Lagrange", "CG") or degree not in (None, 1)):
            raise NotImplementedError("the device path implements P1 Lagrange elements")
        self.mesh = mesh
        self.n_eq = len(element) if isinstance(element, (list, MixedElement)) else 1

    def sub(self, i):
        return SubSpace(self, i)


def RectangleMesh(*args, **kwargs):
    """``RectangleMesh(Point(0, 0), Point(w, h), nx, ny[, diagonal])``: DOLFIN's numbering
    (:func:`fedm_amd.mesh.RectangleMesh`; ``x_lines`` / ``y_lines`` grade it)."""
    from .mesh import RectangleMesh as build
    return build(*args, **kwargs)


def Mesh(path):
    """``Mesh('mesh.xml')``: a DOLFIN XML mesh file (fedm-streamer.py:117)."""
    from .mesh_io import read_dolfin_xml
    return read_dolfin_xml(path)


def XDMFFile(*args, **kwargs):
    """``XDMFFile(path)`` with ``write_checkpoint`` (fedm-gd.py:265-270; ``XDMFFile.Encoding.HDF5``)."""
    from .mesh_io import XDMFFile as writer
    return writer(*args, **kwargs)


class _XdmfEncoding:
    HDF5, ASCII = "HDF5", "ASCII"


XDMFFile.Encoding = _XdmfEncoding


class File:
    """``File('x.pvd') << (function, t)`` (VTU series) and ``File('x.pvd') << boundary_mesh_function``
    (fedm-streamer.py:122: the facet tags, written as a text table next to the name -- a facet
    MeshFunction has no VTU representation here)."""

    def __init__(self, path):
        from pathlib import Path
        self.path = Path(path)
        self._pvd = None

    def __lshift__(self, item):
        if isinstance(item, np.ndarray):
            self.path.parent.mkdir(parents=True, exist_ok=True)
            np.savetxt(self.path.with_suffix(".txt"), np.asarray(item).reshape(item.shape[0], -1), fmt="%d",
                       header="facet tags [cell][local facet]")
            return self
        if self._pvd is None:
            from .mesh_io import PVDFile
            self._pvd = PVDFile(self.path)
        self._pvd << item
        return self


def Point(*xy):
    """``Point(x, y)`` of ``RectangleMesh(Point(0, 0), Point(w, h), nx, ny)`` (fedm-tof.py:89)."""
    return tuple(float(v) for v in xy)


def VectorFunctionSpace(mesh, family, degree):
    """P1 vector fields (fedm-tof.py:98: the constant drift velocity): nodal arrays [vertex][dim]."""
    space = FunctionSpace(mesh, family, degree)
    space.value_dim = 2
    return space


class _Communicator:
    pass


class MPI:
    """``MPI.rank(MPI.comm_world)`` / ``MPI.max(MPI.comm_world, v)`` of the scripts: the ranks of
    torch.distributed when a process group exists, one rank otherwise."""
    comm_world = _Communicator()

    @staticmethod
    def rank(comm=None):
        from .utils import _rank
        return _rank()

    @staticmethod
    def size(comm=None):
        try:
            import torch.distributed as dist
            return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        except ImportError:
            return 1

    @staticmethod
    def max(comm, value):
        if MPI.size() == 1:
            return value
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(value)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])


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

    def apply(self, tensor):
        """``bc.apply(A)`` (unit rows) / ``bc.apply(b)`` (boundary values) on host-assembled tensors."""
        dofs, vals = self.rows(tensor.mesh, 1)
        if isinstance(tensor, HostMatrix):
            tensor.identity_rows(dofs)
        else:
            tensor[dofs] = vals

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
    if isinstance(u, (Function, Sym)):
        # of a nodal Function: cell-wise constant (project()); of a product like D*exp(u): a flux
        # written by hand, recognised by compile_forms (fedm-tof.py:115)
        return Sym("grad", u)
    if not isinstance(u, Unknown):
        raise NotImplementedError("grad() of the potential unknown, of a nodal Function or of D*exp(u)")
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
    if isinstance(x, (Sym, Function, FunctionComponent, Expression)):
        return Sym("exp", x)
    return math.exp(x)


def _p1_mass(mesh):
    """Consistent P1 mass matrix of the mesh and its factorisation (kept on the mesh)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    cached = getattr(mesh, "_p1_mass", None)
    if cached is None:
        x = mesh.coords[mesh.cells]
        d1, d2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
        det = np.abs(d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0])
        vals = det[:, None, None] * ((np.ones((3, 3)) + np.eye(3)) / 24.0)[None]
        c = mesh.cells.astype(np.int64)
        rows = np.broadcast_to(c[:, :, None], vals.shape).ravel()
        cols = np.broadcast_to(c[:, None, :], vals.shape).ravel()
        n = mesh.num_vertices()
        M = sp.coo_matrix((vals.ravel(), (rows, cols)), shape=(n, n)).tocsc()
        cached = mesh._p1_mass = (M, spla.splu(M), det)
    return cached


def _nodal(f):
    """Nodal values of a P1 Function on the host; a Function bound to a device state is downloaded."""
    if getattr(f, "state", None) is not None:
        U = f.state.array()
        return U[:, 0] if U.ndim == 2 and U.shape[1] == 1 else U
    return np.asarray(f.vector(), dtype=float)


def _project_exp(arg, space):
    """``project(exp(w), V)`` of a nodal P1 Function or of a spatial Expression (fedm-tof.py:155-156),
    the way FFC sets it up: an Expression of degree d is interpolated at the P_d lattice nodes of
    every cell, and the quadrature degree is UFL's estimate (d + 2) + 1."""
    from . import quadrature
    mesh = space.mesh
    M, lu, det = _p1_mass(mesh)
    if isinstance(arg, Expression):
        deg = int(arg.degree) + 3
        xq, wq = quadrature.triangle(deg)
        B, lam = quadrature.lagrange_interpolation_matrix(int(arg.degree), xq)
        phi_nodes = np.stack([1 - lam[:, 0] - lam[:, 1], lam[:, 0], lam[:, 1]], axis=1)
        nodes = np.einsum("na,cad->cnd", phi_nodes, mesh.coords[mesh.cells])
        at_q = np.asarray(arg(nodes)) @ B.T
    else:
        xq, wq = quadrature.triangle(4)
        phi_q = np.stack([1 - xq[:, 0] - xq[:, 1], xq[:, 0], xq[:, 1]], axis=1)
        at_q = _nodal(arg)[mesh.cells] @ phi_q.T
    phi = np.stack([1 - xq[:, 0] - xq[:, 1], xq[:, 0], xq[:, 1]], axis=1)
    rhs = np.einsum("q,cq,qa->ca", wq, np.exp(at_q), phi) * det[:, None]
    b = np.bincount(mesh.cells.ravel(), weights=rhs.ravel(), minlength=mesh.num_vertices())
    return Function(space, values=lu.solve(b))


def errornorm(u, uh, norm_type="l2"):
    """L2 norm of the difference of two P1 Functions (fedm-tof.py:157)."""
    if str(norm_type).lower() != "l2":
        raise NotImplementedError("errornorm: L2 only")
    M = _p1_mass(u.space.mesh)[0]
    e = _nodal(u) - _nodal(uh)
    return float(np.sqrt(e @ (M @ e)))


def norm(u, norm_type="l2"):
    """L2 norm of a P1 Function (fedm-tof.py:157)."""
    if str(norm_type).lower() != "l2":
        raise NotImplementedError("norm: L2 only")
    M = _p1_mass(u.space.mesh)[0]
    v = _nodal(u)
    return float(np.sqrt(v @ (M @ v)))


def project(expr, space=None, solver_type=None):
    """L2 projection onto P1 of ``c * sqrt(dot(grad(f), grad(f)))``-type expressions of nodal
    Functions (fedm-gd.py:309,432: the reduced electric field): the gradient of a P1 Function is
    constant per cell; consistent mass matrix, factorised once per mesh (host, post-processing
    size -- the device-resident pipeline is `fedm_gd_prep_step`).  Also ``exp(w)`` of one nodal
    Function or spatial Expression (fedm-tof.py:155-156)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    if isinstance(expr, Sym) and expr.op == "exp" and isinstance(expr.args[0], (Function, Expression)) \
            and space is not None:
        return _project_exp(expr.args[0], space)
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


# ---------------------------------------------------------------------------------------
# lhs / rhs / assemble / bc.apply / solve for the scalar Poisson problem the scripts solve once
# before the time loop (fedm-streamer.py:203-215, fedm-gd.py:287-300): host side, scipy -- a
# one-off solve of post-processing size; the Newton systems of the time loop are the device's.
# ---------------------------------------------------------------------------------------
class _FormSide:
    def __init__(self, piece, side):
        self.piece, self.side = piece, side


def lhs(form):
    return _FormSide(form, "lhs")


def rhs(form):
    return _FormSide(form, "rhs")


class HostMatrix:
    """Assembled P1 matrix (scipy CSR) with DOLFIN's ``bc.apply(A)`` semantics."""

    def __init__(self, csr, mesh):
        self.csr, self.mesh = csr.tolil(), mesh

    def identity_rows(self, dofs):
        for d in dofs:
            self.csr.rows[d], self.csr.data[d] = [int(d)], [1.0]


class HostVector(np.ndarray):
    """Assembled P1 vector; carries its mesh for ``bc.apply(b)``."""
    mesh = None


def _at_quadrature(expr, mesh, phi_q):
    """Values [cell][point] of an expression of nodal P1 Functions, Constants, numbers and parameter
    Expressions at the quadrature points of every cell (the Functions interpolated first, then the
    arithmetic: what FFC generates for exp(u_old) in a form)."""
    if isinstance(expr, Real):
        return float(expr)
    if isinstance(expr, Constant):
        return float(expr.value)
    if isinstance(expr, Function):
        return np.asarray(expr.vector(), dtype=float)[mesh.cells] @ phi_q.T
    if isinstance(expr, Expression):
        return expr.value()
    if isinstance(expr, Sym):
        if expr.op == "pow":
            return _at_quadrature(expr.args[0], mesh, phi_q) ** expr.args[1]
        a = [_at_quadrature(v, mesh, phi_q) for v in expr.args]
        ops = {"add": lambda: a[0] + a[1], "sub": lambda: a[0] - a[1], "mul": lambda: a[0] * a[1],
               "div": lambda: a[0] / a[1], "neg": lambda: -a[0], "abs": lambda: np.abs(a[0]),
               "exp": lambda: np.exp(a[0]), "sqrt": lambda: np.sqrt(a[0]), "log": lambda: np.log(a[0])}
        if expr.op in ops:
            return ops[expr.op]()
    raise NotImplementedError(f"assemble(): {type(expr).__name__} in a host-assembled source term")


def assemble(form, tensor=None):
    """``assemble(lhs(F))`` / ``assemble(rhs(F))`` of a ``weak_form_Poisson_equation`` on a scalar P1
    space: ``2 pi r grad(u).grad(v) dx`` and ``2 pi r f v dx`` (``r`` the coordinate Expression, or the
    default ``0.5/pi`` of plane problems), the source at the points of the degree set in
    ``parameters["form_compiler"]["quadrature_degree"]`` (2 if unset)."""
    import scipy.sparse as sp
    from . import quadrature
    from .functions import PoissonEq, _axisymmetric
    if not isinstance(form, _FormSide) or not isinstance(form.piece, PoissonEq):
        raise NotImplementedError("assemble(): lhs/rhs of weak_form_Poisson_equation on a scalar space")
    piece = form.piece
    mesh = piece.u.space.mesh
    x = mesh.coords[mesh.cells]
    d1, d2 = x[:, 1] - x[:, 0], x[:, 2] - x[:, 0]
    det = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]
    axis = _axisymmetric(piece.r)
    two_pi_r = 2.0 * np.pi * x[:, :, 0] if axis else np.ones(x.shape[:2])      # at the vertices, [cell][a]
    n = mesh.num_vertices()
    c = mesh.cells.astype(np.int64)
    if form.side == "lhs":
        G = np.stack([np.stack([x[:, 1, 1] - x[:, 2, 1], x[:, 2, 0] - x[:, 1, 0]], axis=1),
                      np.stack([x[:, 2, 1] - x[:, 0, 1], x[:, 0, 0] - x[:, 2, 0]], axis=1),
                      np.stack([x[:, 0, 1] - x[:, 1, 1], x[:, 1, 0] - x[:, 0, 0]], axis=1)], axis=1) / det[:, None, None]
        weight = 0.5 * np.abs(det) * two_pi_r.mean(axis=1)                       # r is linear: exact
        vals = np.einsum("cad,cbd->cab", G, G) * weight[:, None, None]
        rows = np.broadcast_to(c[:, :, None], vals.shape).ravel()
        cols = np.broadcast_to(c[:, None, :], vals.shape).ravel()
        return HostMatrix(sp.coo_matrix((vals.ravel(), (rows, cols)), shape=(n, n)).tocsr(), mesh)
    degree = parameters["form_compiler"]["quadrature_degree"]
    xq, wq = quadrature.triangle(degree if degree and degree > 0 else 2)
    phi_q = np.stack([1 - xq[:, 0] - xq[:, 1], xq[:, 0], xq[:, 1]], axis=1)       # [point][a]
    f_q = np.broadcast_to(_at_quadrature(piece.f, mesh, phi_q), (c.shape[0], len(wq)))
    r_q = two_pi_r @ phi_q.T
    elem = np.einsum("q,cq,cq,qa->ca", wq, f_q, r_q, phi_q) * np.abs(det)[:, None]
    b = np.bincount(c.ravel(), weights=elem.ravel(), minlength=n).view(HostVector)
    b.mesh = mesh
    return b


def solve(A, x, b, *solver_arguments):
    """``solve(A, Phi.vector(), b[, 'mumps'])``: sparse direct solve on the host, into ``x``."""
    import scipy.sparse.linalg as spla
    if not isinstance(A, HostMatrix):
        raise NotImplementedError("solve(): a matrix from assemble(lhs(...))")
    x[...] = spla.spsolve(A.csr.tocsc(), np.asarray(b, dtype=float))
    return x


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
