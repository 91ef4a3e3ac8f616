"""Normal form of field-dependent coefficients:  f(E) = sum_i c_i E^p_i exp(q_i E^r_i).

The streamer deck gives mobility, diffusion and ionisation coefficients as
one-line Python/UFL strings in ``E_m`` (the magnitude of the electric field),
e.g. ``2.3987*E_m**(-0.26)`` (transport_coefficients/e_Nb.dat:12) or
``(1.1944e6 + 4.3666e26 * E_m**(-3))*exp(-2.73e7/E_m)-340.75`` (alpha.dat:12),
which the reference ``eval``s into UFL (fedm-streamer.py:237-239).  Here the
string is parsed with a restricted ``ast`` grammar (never ``eval``) and
expanded into a sum of generalised monomials, which the element kernel
evaluates -- value and d/dE -- once per cell.
"""
import ast
import math
from numbers import Real


class TermSum:
    __array_priority__ = 1000

    def __init__(self, terms=()):
        merged = {}
        for c, p, q, r in terms:
            c, p, q, r = float(c), float(p), float(q), float(r)
            if q == 0.0:
                r = 0.0
            key = (p, q, r)
            merged[key] = merged.get(key, 0.0) + c
        self.terms = [(c, p, q, r) for (p, q, r), c in merged.items() if c != 0.0]

    # -- constructors ---------------------------------------------------------
    @classmethod
    def const(cls, c):
        return cls([(float(c), 0.0, 0.0, 0.0)])

    @classmethod
    def field(cls):
        """The symbol E_m itself."""
        return cls([(1.0, 1.0, 0.0, 0.0)])

    @classmethod
    def coerce(cls, v):
        if isinstance(v, TermSum):
            return v
        if isinstance(v, Real):
            return cls.const(v)
        if isinstance(v, str):
            return parse(v)
        raise TypeError(f"cannot interpret {v!r} as a coefficient of |E|")

    # -- queries ----------------------------------------------------------------
    def is_const(self):
        return all(p == 0.0 and q == 0.0 for _, p, q, _ in self.terms)

    def const_value(self):
        if not self.is_const():
            raise ValueError("coefficient depends on |E|")
        return sum(c for c, *_ in self.terms)

    def __call__(self, E):
        return sum(c * E ** p * math.exp(q * E ** r if q else 0.0) for c, p, q, r in self.terms)

    def derivative(self, E):
        return sum(c * E ** p * math.exp(q * E ** r if q else 0.0) * (p + r * q * E ** r) / E
                   for c, p, q, r in self.terms if (p or q))

    def fill(self, cstruct):
        from . import _lib
        if len(self.terms) > _lib.MAX_TERMS:
            raise ValueError(f"coefficient has {len(self.terms)} terms, at most {_lib.MAX_TERMS}")
        cstruct.n_terms = len(self.terms)
        for i, (c, p, q, r) in enumerate(self.terms):
            # (+ 0.0: a negative zero left by the algebra becomes +0.0 -- the same model, the same bytes)
            cstruct.c[i], cstruct.p[i], cstruct.q[i], cstruct.r[i] = c + 0.0, p + 0.0, q + 0.0, r + 0.0

    # -- algebra ------------------------------------------------------------------
    def __add__(self, o):
        return TermSum(self.terms + TermSum.coerce(o).terms)

    __radd__ = __add__

    def __neg__(self):
        return TermSum([(-c, p, q, r) for c, p, q, r in self.terms])

    def __sub__(self, o):
        return self + (-TermSum.coerce(o))

    def __rsub__(self, o):
        return TermSum.coerce(o) - self

    def __mul__(self, o):
        if not isinstance(o, (TermSum, Real)):
            return NotImplemented
        o = TermSum.coerce(o)
        out = []
        for c1, p1, q1, r1 in self.terms:
            for c2, p2, q2, r2 in o.terms:
                if q1 == 0.0:
                    q, r = q2, r2
                elif q2 == 0.0:
                    q, r = q1, r1
                elif r1 == r2:
                    q, r = q1 + q2, r1
                else:
                    raise ValueError("product of exponentials with different powers of |E| "
                                     "is outside the supported coefficient family")
                out.append((c1 * c2, p1 + p2, q, r))
        return TermSum(out)

    __rmul__ = __mul__

    def __pow__(self, n):
        if isinstance(n, TermSum):
            n = n.const_value()
        n = float(n)
        if len(self.terms) == 1:
            c, p, q, r = self.terms[0]
            return TermSum([(c ** n, p * n, q * n, r)])
        if n == int(n) and n >= 0:
            out = TermSum.const(1.0)
            for _ in range(int(n)):
                out = out * self
            return out
        raise ValueError("non-integer power of a sum is outside the supported coefficient family")

    def __truediv__(self, o):
        return self * TermSum.coerce(o) ** -1.0

    def __rtruediv__(self, o):
        return TermSum.coerce(o) * self ** -1.0

    def exp(self):
        """exp of a constant or of a single monomial a*E^b."""
        if not self.terms:
            return TermSum.const(1.0)
        const = sum(c for c, p, q, _ in self.terms if p == 0.0 and q == 0.0)
        rest = [(c, p, q, r) for c, p, q, r in self.terms if not (p == 0.0 and q == 0.0)]
        if len(rest) > 1 or any(q != 0.0 for _, _, q, _ in rest):
            raise ValueError("exp() argument must be a*E_m**b (+ constant)")
        if not rest:
            return TermSum.const(math.exp(const))
        c, p, _, _ = rest[0]
        return TermSum([(math.exp(const), 0.0, c, p)])

    def __repr__(self):
        return "TermSum(" + " + ".join(
            f"{c:g}*E^{p:g}" + (f"*exp({q:g}*E^{r:g})" if q else "") for c, p, q, r in self.terms) + ")"


_BINOPS = {ast.Add: lambda a, b: a + b, ast.Sub: lambda a, b: a - b,
           ast.Mult: lambda a, b: a * b, ast.Div: lambda a, b: a / b,
           ast.Pow: lambda a, b: a ** b}


def parse(text, symbol="E_m"):
    """Parse a deck expression in ``E_m`` into a :class:`TermSum` (no eval)."""
    try:
        tree = ast.parse(text.strip(), mode="eval")
    except SyntaxError as exc:
        raise ValueError(f"cannot parse coefficient expression {text!r}") from exc

    def walk(node):
        if isinstance(node, ast.Expression):
            return walk(node.body)
        if isinstance(node, ast.Constant) and isinstance(node.value, (int, float)):
            return TermSum.const(node.value)
        if isinstance(node, ast.Name):
            if node.id == symbol:
                return TermSum.field()
            if node.id == "pi":
                return TermSum.const(math.pi)
            raise ValueError(f"unknown symbol {node.id!r} in coefficient expression {text!r}")
        if isinstance(node, ast.UnaryOp) and isinstance(node.op, (ast.USub, ast.UAdd)):
            v = walk(node.operand)
            return -v if isinstance(node.op, ast.USub) else v
        if isinstance(node, ast.BinOp) and type(node.op) in _BINOPS:
            return _BINOPS[type(node.op)](walk(node.left), walk(node.right))
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Name) and not node.keywords:
            args = [walk(a) for a in node.args]
            if node.func.id == "exp" and len(args) == 1:
                return args[0].exp()
            if node.func.id == "sqrt" and len(args) == 1:
                return args[0] ** 0.5
            if node.func.id == "pow" and len(args) == 2:
                return args[0] ** args[1]
        raise ValueError(f"unsupported construct in coefficient expression {text!r}")

    return walk(tree)
