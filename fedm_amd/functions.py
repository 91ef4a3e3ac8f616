"""FEDM's ``fedm.functions`` surface on top of the MI355X device path.

Same function names, argument order and error behaviour as the reference
(fedm/functions.py); the objects that are UFL/DOLFIN types there are light
descriptors here (see :mod:`fedm_amd.forms`) because DOLFIN does not exist on
the GPU box.  The hot path -- ``Problem.F``, ``Problem.J`` and
``nonlinear_solver.solve`` -- runs in libfedm_hip.so.
"""
import warnings
from pathlib import Path
from typing import Any, List, Optional, Tuple

import numpy as np

from .mesh import CircleSubDomain, LineSubDomain, Marking_boundaries  # noqa: F401  (fedm/functions.py:49-124)
from .physical_constants import elementary_charge, kB, kB_eV
from .utils import comma_separated, print_rank_0

DOLFIN_EPS = 3.0e-16


def _must_be_one_of(where, what, value, allowed, quote=True):
    """The reference's wording for a bad option: ``fedm.<where>: <what> '<value>' not recognised.
    Must be one of a, b, c.``"""
    if value not in allowed:
        shown = f"'{value}'" if quote else f"{value}"
        raise ValueError(f"fedm.{where}: {what} {shown} not recognised. Must be one of "
                         f"{comma_separated(allowed)}.")


# ---------------------------------------------------------------------------
# fedm/functions.py:15-45
# ---------------------------------------------------------------------------
def modify_approximation_vars(approximation_type, number_of_species, particle_species,
                              masses, charges):
    """LFA drops the first species; n_eq = n_species + 1.  Mutates the lists like the
    reference does."""
    if approximation_type not in ("LFA", "LMEA"):
        raise ValueError(f"fedm.modify_approximation_vars: The approximation type {approximation_type} "
                         f"is not recognised. Must be one of {comma_separated(['LFA', 'LMEA'])}.")
    if approximation_type == "LFA":
        number_of_species -= 1
        particle_species.pop(0)
        masses.pop(0)
        charges.pop(0)
    number_of_eq = number_of_species + 1
    return number_of_species, number_of_eq, particle_species, masses, charges


# ---------------------------------------------------------------------------
# fedm/functions.py:127-171 -- list helpers
# ---------------------------------------------------------------------------
def Mixed_element_list(number_of_equations, element):
    return [element] * number_of_equations


def Function_space_list(number_of_equations, function_space):
    return [function_space] * number_of_equations


def Function_definition(function_space, function_type, eq_number=1):
    from . import forms
    functions = {"TrialFunction": forms.TrialFunction, "TestFunction": forms.TestFunction,
                 "Function": forms.Function}
    if function_type not in functions:
        raise ValueError(f"fedm.Function_definition: Invalid function_type '{function_type}'. "
                         f"Possible values are {comma_separated(functions)}.")
    return [functions[function_type](function_space) for _ in range(eq_number)]


# ---------------------------------------------------------------------------
# fedm/functions.py:219-528 -- weak-form builders.  They return descriptors that
# Problem() compiles into the device model (fedm_amd.device.Model).
# ---------------------------------------------------------------------------
class FluxDesc:
    def __init__(self, sign, u, D, mu, E, grad_diffusion, logarithm_representation):
        self.sign, self.u, self.D, self.mu, self.E = sign, u, D, mu, E
        self.grad_diffusion, self.log = grad_diffusion, logarithm_representation


class FormPiece:
    def __add__(self, other):
        return FormSum([self]) + other

    def __radd__(self, other):
        return FormSum([self]).__radd__(other)


class FormSum:
    """F = 0.0; F += piece ... (fedm-streamer.py:252-271)"""

    def __init__(self, pieces=()):
        self.pieces = list(pieces)

    def __add__(self, other):
        if isinstance(other, FormSum):
            return FormSum(self.pieces + other.pieces)
        if isinstance(other, FormPiece):
            return FormSum(self.pieces + [other])
        if isinstance(other, (int, float)) and other == 0:
            return self
        return NotImplemented

    __radd__ = __add__


class BalanceEq(FormPiece):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class PoissonEq(FormPiece):
    def __init__(self, **kw):
        self.__dict__.update(kw)


class BoundaryTerm(FormPiece):
    def __init__(self, **kw):
        self.__dict__.update(kw)


def Flux(sign, u, D, mu, E, grad_diffusion=True, logarithm_representation=True):
    """Drift-diffusion flux, fedm/functions.py:219-237: -grad(D exp(u)) [or -D grad(exp(u))]
    + sign * mu * E * exp(u).  With P1 elements and coefficients that depend on |E| only,
    D is constant per cell and both diffusion forms coincide."""
    return FluxDesc(sign, u, D, mu, E, grad_diffusion, logarithm_representation)


def weak_form_balance_equation(equation_type, dt, dt_old, dx, u, u_old, u_old1, v, f, Gamma,
                               r=0.5 / np.pi, D=None, log_representation=False):
    """Weak form of a particle balance equation, fedm/functions.py:240-368 (same positional
    arguments, same ValueErrors)."""
    me = "weak_form_balance_equation_log_representation"       # the name the reference reports
    kinds = ["reaction", "diffusion-reaction", "drift-diffusion-reaction"]
    if equation_type not in kinds:                              # (sic: unbalanced quote)
        raise ValueError(f"fedm.{me}: The equation type {equation_type}' is not recognised. "
                         f"Must be one of {comma_separated(kinds)}.")
    if D is None and equation_type == "diffusion-reaction":
        raise ValueError(f"fedm.{me}: When 'equation_type' is diffusion-reaction, must also supply "
                         "the diffusion coefficient 'D'.")
    return BalanceEq(equation_type=equation_type, dt=dt, dt_old=dt_old, dx=dx, u=u, u_old=u_old,
                     u_old1=u_old1, v=v, f=f, Gamma=Gamma, r=r, D=D, log=bool(log_representation))


def weak_form_balance_equation_log_representation(*args, **kwargs):
    return weak_form_balance_equation(*args, **kwargs, log_representation=True)


def weak_form_Poisson_equation(dx, u, v, f, r=0.5 / np.pi):
    """2*pi*r*(inner(grad(u), grad(v)) - f*v)*dx, fedm/functions.py:379-401."""
    return PoissonEq(dx=dx, u=u, v=v, f=f, r=r)


def Boundary_flux(bc_type, equation_type, particle_type, sign, mu, E, normal, u, gamma, v,
                  ds_temp, r=0.5 / np.pi, vth=0.0, ref=1.0, Ion_flux=0.0):
    """Boundary terms, fedm/functions.py:404-528: same checks, warning and return
    convention (a form piece, or 0.0 when the condition contributes nothing)."""
    if "_" in bc_type:
        warnings.warn("fedm.BoundaryFlux: bc_type should have spaces, not underscores")
        bc_type = bc_type.replace("_", " ")
    _must_be_one_of("Boundary_flux", "boundary condition type", bc_type, ["zero flux", "flux source", "Neumann"])
    if bc_type != "zero flux":
        _must_be_one_of("Boundary_flux", "equation type", equation_type,
                        ["reaction", "diffusion-reaction", "drift-diffusion-reaction"])
    if (bc_type, equation_type) == ("flux source", "diffusion-reaction") and particle_type not in ("Heavy", "electrons"):
        raise ValueError(f"fedm.Boundary_flux: particle type '{particle_type}' not recognised.\n"
                         f"Must be one of {comma_separated(['Heavy', 'electrons'])}.")
    if bc_type == "flux source" and equation_type != "reaction":
        # (1-ref)/(1+ref) * [vth/2 (+ |sign mu E.n|)] exp(u) [- 2 gamma Ion_flux/(1+ref) for electrons],
        # functions.py:514-522: lowered onto the LMEA device model by fedm_amd.lmea.compile_lmea
        return BoundaryTerm(kind="flux source", equation_type=equation_type, particle_type=particle_type,
                            u=u, sign=sign, mu=mu, gamma=gamma, tag=ds_temp.tag, ds=ds_temp, r=r, vth=vth,
                            ref=ref, Ion_flux=Ion_flux)
    if bc_type == "Neumann" and equation_type == "drift-diffusion-reaction":
        return BoundaryTerm(kind="Neumann", u=u, sign=sign, mu=mu, tag=ds_temp.tag,
                            ds=ds_temp, r=r)
    return 0.0


def _axisymmetric(r):
    """r = Expression('x[0]') (cylindrical) vs the default 0.5/pi (fedm/functions.py:251)."""
    if isinstance(r, (int, float)):
        if abs(r - 0.5 / np.pi) > 1e-15:
            raise NotImplementedError("constant r other than the default 0.5/pi")
        return False
    return True


def _constant_nodal(x, dim=None):
    """The value of a coefficient that is the same at every node: a number, a Constant, or a Function
    interpolated from one (fedm-tof.py:111-113).  None when it varies."""
    from . import forms
    if isinstance(x, (int, float)):
        return float(x)
    if isinstance(x, forms.Constant):
        return np.asarray(x.value, dtype=float) if isinstance(x.value, tuple) else float(x.value)
    if isinstance(x, forms.Function):
        v = np.asarray(x.vector(), dtype=float)
        first = v[0]
        if np.all(v == first):
            return np.asarray(first, dtype=float) if v.ndim == 2 else float(first)
    return None


def _match_manual_flux(gamma, index):
    """``Gamma = -grad(D*exp(u)) + w*exp(u)`` written out by hand (fedm-tof.py:115): (D, (w_r, w_z))
    when D and the drift velocity w are the same at every node, else None."""
    from . import forms

    def density_factor(node):
        # coefficient c of  c * exp(u_index)  (either order)
        if isinstance(node, forms.Sym) and node.op == "mul":
            a, b = node.args
            for coef, dens in ((a, b), (b, a)):
                if isinstance(dens, forms.Density) and dens.index == index:
                    return coef
        return None
    if not (isinstance(gamma, forms.Sym) and gamma.op == "add"):
        return None
    diffusion = drift = None
    for term in gamma.args:
        if isinstance(term, forms.Sym) and term.op == "neg" and isinstance(term.args[0], forms.Sym) \
                and term.args[0].op == "grad":
            diffusion = density_factor(term.args[0].args[0])
        else:
            drift = density_factor(term)
    if diffusion is None or drift is None:
        return None
    D, w = _constant_nodal(diffusion), _constant_nodal(drift)
    if D is None or w is None or np.ndim(D) != 0 or np.shape(w) != (2,):
        return None
    return float(D), (float(w[0]), float(w[1]))


def compile_forms(F, quadrature_degree=None):
    """FormSum -> (device Model, mesh, facet tags).  The compiler of this façade: what
    FFC does for the reference, restricted to the model family of the device kernels."""
    from . import forms
    from .device import Model, Reaction
    from .physical_constants import elementary_charge as q_e, epsilon_0 as eps0
    from .termsum import TermSum
    pieces = F.pieces if isinstance(F, FormSum) else [F]
    from . import lmea
    if lmea.is_lmea(pieces):
        qd = quadrature_degree if quadrature_degree is not None else forms.parameters["form_compiler"]["quadrature_degree"]
        if qd is None or qd < 0:
            raise NotImplementedError("set parameters['form_compiler']['quadrature_degree'] (fedm-gd.py:28)")
        model, mesh, tags, binding = lmea.compile_lmea(pieces, qd)
        model.field_binding = binding
        return model, mesh, tags
    balances = sorted([p for p in pieces if isinstance(p, BalanceEq)], key=lambda p: p.u.index)
    poissons = [p for p in pieces if isinstance(p, PoissonEq)]
    bterms = [p for p in pieces if isinstance(p, BoundaryTerm)]
    if not balances:
        raise ValueError("no balance equation in the form")
    if len(poissons) > 1:
        raise ValueError("more than one Poisson equation in the form")
    ns = len(balances)
    if [p.u.index for p in balances] != list(range(ns)):
        raise ValueError("balance equations must use the leading components of the mixed space")
    space = balances[0].u.space
    mesh = space.mesh
    axis = _axisymmetric(balances[0].r)
    eq_type = [p.equation_type for p in balances]
    zero = TermSum.const(0.0)
    mu, D, Z = [zero] * ns, [zero] * ns, [0.0] * ns
    reactions = []
    # one representation per model: log_representation of the balance equations, logarithm_representation
    # of their fluxes and the way the densities enter the sources (exp(u[i]) or u[i]) must agree
    log = bool(getattr(balances[0], "log", True))
    if any(bool(getattr(p, "log", True)) != log for p in balances):
        raise ValueError("all balance equations of a model must use the same representation (logarithmic or not)")

    def check_densities(terms, where):
        for term in terms:
            wrong = set(term.powers) - term.bare if not log else term.bare
            if wrong:
                raise ValueError(f"{where}: densities must enter as " + ("exp(u[i])" if log else "u[i]")
                                 + f" in the {'logarithmic' if log else 'non-logarithmic'} representation")
    drift_w, ext_sources = [None] * ns, []
    for s, p in enumerate(balances):
        if isinstance(p.f, forms.Expression):
            # a source given as a spatial Expression of degree d (fedm-tof.py:116): interpolated at the
            # P_d lattice nodes of every cell and refreshed before each solve (Problem.before_solve)
            ext_sources.append((s, p.f))
        if p.equation_type == "drift-diffusion-reaction" and not isinstance(p.Gamma, FluxDesc):
            manual = _match_manual_flux(p.Gamma, p.u.index)
            if manual is None:
                raise ValueError("drift-diffusion-reaction needs Gamma = Flux(...) or "
                                 "-grad(D*exp(u)) + w*exp(u) with constant D and w")
            if not log:
                raise ValueError("a flux written with exp(u) belongs to the logarithmic representation")
            D[s], drift_w[s], Z[s] = TermSum.const(manual[0]), manual[1], -1.0
            continue
        if p.equation_type == "drift-diffusion-reaction":
            g = p.Gamma
            if not isinstance(g, FluxDesc):
                raise ValueError("drift-diffusion-reaction needs Gamma = Flux(...)")
            if bool(g.log) != log:
                raise ValueError("Flux(..., logarithm_representation=...) must match the balance equation's representation")
            mu[s], D[s], Z[s] = TermSum.coerce(g.mu), TermSum.coerce(g.D), float(g.sign)
        elif p.equation_type == "diffusion-reaction":
            D[s] = TermSum.coerce(p.D)
        if isinstance(p.f, forms.Expression):
            continue
        check_densities(forms.RateSum.coerce(p.f).terms, f"source term of species {s}")
        for term in forms.RateSum.coerce(p.f).terms:
            power = [int(term.powers.get(i, 0)) for i in range(ns)]
            if any(i >= ns for i in term.powers):
                raise ValueError("source terms may only contain exp(u) of the species")
            for rc in reactions:                       # same rate on several species rows
                if rc.k.terms == term.coef.terms and list(rc.power) == power and rc.net[s] == 0:
                    rc.net[s] = 1
                    break
            else:
                net = [0] * ns
                net[s] = 1
                reactions.append(Reaction(term.coef, power, net))
    if poissons:
        ps = poissons[0]
        if ps.u.index != ns:
            raise ValueError("the potential must be the last component of the mixed space")
        check_densities(forms.RateSum.coerce(ps.f).terms, "Poisson source")
        for term in forms.RateSum.coerce(ps.f).terms:
            if len(term.powers) != 1 or list(term.powers.values()) != [1] or not term.coef.is_const():
                raise ValueError("the Poisson source must be sum_i Z_i e/eps0 exp(u_i)")
            i = next(iter(term.powers))
            zi = term.coef.const_value() * eps0 / q_e
            if abs(zi - round(zi)) < 1e-12:
                zi = float(round(zi))            # charge numbers
            if Z[i] not in (0.0,) and abs(Z[i] - zi) > 1e-9 * max(1.0, abs(zi)):
                raise ValueError("charge in the Poisson source differs from the flux sign")
            Z[i] = zi
    n_tags = max([b.tag for b in bterms], default=0)
    tags_mf = next((b.ds.subdomain_data for b in bterms if b.ds.subdomain_data is not None), None)
    if tags_mf is not None:
        n_tags = max(n_tags, int(np.max(tags_mf)))
    bc_kind = [["zero flux"] * ns for _ in range(n_tags)]
    for b in bterms:
        bc_kind[b.tag - 1][b.u.index] = b.kind
    qd = quadrature_degree
    if qd is None:
        qd = forms.parameters["form_compiler"]["quadrature_degree"]
    if (qd is None or qd < 0) and ns == 1 and not poissons and ext_sources and drift_w[0] is not None:
        # the time-of-flight forms set no degree (fedm-tof.py): UFL's estimate for them, pinned on the
        # reference's golden of that case
        qd = 8
    if qd is None or qd < 0:
        raise NotImplementedError(
            "set parameters['form_compiler']['quadrature_degree'] (UFL's automatic degree "
            "estimation is reproduced only for the time-of-flight case, see DESIGN.md)")
    extra = {}
    if any(w is not None for w in drift_w):
        extra["drift_w"] = drift_w
    if ext_sources:
        degrees = [0] * ns
        for s, e in ext_sources:
            if int(e.degree) not in (1, 2):
                raise NotImplementedError("Expression sources of degree 1 or 2")
            degrees[s] = int(e.degree)
        extra["ext_source_degree"] = degrees
    model = Model(n_species=ns, poisson=bool(poissons), eq_type=eq_type, Z=Z, mu=mu, D=D,
                  reactions=reactions, bc_kind=bc_kind, quadrature_degree=int(qd),
                  axisymmetric=axis, log_representation=log, **extra)
    model.expression_sources = ext_sources
    return model, mesh, tags_mf


# ---------------------------------------------------------------------------
# fedm/functions.py:174-202 -- the drop-in seam
# ---------------------------------------------------------------------------
class Problem:
    """Nonlinear problem: ``Problem(J, F, bcs)`` as in the reference.

    ``F`` is the sum of weak-form pieces built with the functions above (or anything already
    bound to a device problem); ``J`` is accepted for signature compatibility -- the exact
    Jacobian is part of the device kernels.  ``F(b, x)`` assembles the residual and applies
    the Dirichlet rows (functions.py:188-194), ``J(A, x)`` the Jacobian (functions.py:196-202);
    ``b``/``A`` may be None to keep the result on the device (what ``solve`` does)."""

    def __init__(self, J, F, bcs, device_problem=None, device=0):
        self.bilinear_form = J
        self.linear_form = F
        self.bcs = bcs
        self.before_solve = None
        if device_problem is None and isinstance(F, (FormSum, FormPiece)):
            from . import forms
            from .device import DeviceProblem
            model, mesh, tags = compile_forms(F)
            dofs = [np.zeros(0, dtype=np.int64)]
            vals = [np.zeros(0)]
            for bc in bcs or []:
                d, v = bc.rows(mesh, model.n_eq)
                dofs.append(d)
                vals.append(v)
            device_problem = DeviceProblem(mesh.coords, mesh.cells, model, facet_tags=tags,
                                           dirichlet_dofs=np.concatenate(dofs),
                                           dirichlet_vals=np.concatenate(vals), device=device)
            # mixed Functions of the script become handles of the device-resident states
            # (u_new from action(F, u_new), u_old / u_old1 from the balance equations); what a
            # rev_assigner put into them before this point is uploaded now
            pieces = F.pieces if isinstance(F, FormSum) else [F]
            owners = {"new": getattr(F, "u_new", None)}
            for p in pieces:
                for which, comp in (("old", getattr(p, "u_old", None)), ("old1", getattr(p, "u_old1", None))):
                    if isinstance(comp, forms.FunctionComponent):
                        owners[which] = comp.function
                    elif isinstance(comp, forms.Function) and model.n_eq == 1:
                        owners[which] = comp          # one equation on a plain space (fedm-tof.py:102-104)
            initial = {}
            for which, fn in owners.items():
                if isinstance(fn, forms.Function):
                    fn.state = forms.DeviceState(device_problem, which)
                    if getattr(fn, "parts", None):
                        initial["u_" + which] = np.stack([np.asarray(f.vector(), dtype=float) for f in fn.parts], axis=1)
                    elif model.n_eq == 1 and getattr(fn, "_v", None) is not None:
                        initial["u_" + which] = np.asarray(fn._v, dtype=float)   # interpolated before Problem()
            if initial:
                device_problem.set_state(**initial)
            binding = getattr(model, "field_binding", None)
            timed = [bc for bc in (bcs or []) if isinstance(bc.value, forms.Expression)]
            sources = getattr(model, "expression_sources", [])
            source_nodes, source_programs = {}, {}
            for s_, e in sources:
                try:    # Expression strings run on the device; Python callables are evaluated here
                    ops, consts, names = forms.expression_program(e)
                    device_problem.set_ext_source_program(s_, ops, consts, len(names))
                    source_programs[s_] = names
                    continue
                except NotImplementedError:
                    pass
                lam = np.array([(i / e.degree, j / e.degree) for j in range(e.degree + 1)
                                for i in range(e.degree + 1 - j)])
                phi = np.stack([1 - lam[:, 0] - lam[:, 1], lam[:, 0], lam[:, 1]], axis=1)
                source_nodes[s_] = np.einsum("na,cad->cnd", phi, mesh.coords[mesh.cells])

            # a script that calls the nonlinear solver itself (fedm-tof.py:149) has the step sizes in the
            # dt / dt_old Expressions of its balance equations; adaptive_solver sets the same values
            steps = next(((p.dt, p.dt_old) for p in pieces
                          if hasattr(getattr(p, "dt", None), "time_step")
                          and hasattr(getattr(p, "dt_old", None), "time_step")), None)

            def refresh():
                # time-dependent Dirichlet values (functions.py:1042-1044 advances their `t`) and the
                # nodal coefficient Functions the script has just interpolated on the host
                if steps is not None and sources:
                    device_problem.set_step(steps[0].time_step, steps[1].time_step)
                if timed:
                    device_problem.set_dirichlet_values(
                        np.concatenate([bc.rows(mesh, model.n_eq)[1] for bc in bcs]))
                if binding is not None:
                    device_problem.set_gd_fields(binding.stack(mesh.num_vertices()))
                for s_, e in sources:       # the script has advanced the Expression's parameters (f.t = t)
                    if s_ in source_programs:
                        device_problem.eval_ext_source(s_, [float(getattr(e, n)) for n in source_programs[s_]])
                    else:
                        device_problem.set_ext_source(s_, np.asarray(e(source_nodes[s_]), dtype=float))
            if timed or binding is not None or sources:
                self.before_solve = refresh
        self.device = device_problem if device_problem is not None else getattr(F, "device", None)
        if self.device is None:
            raise ValueError("fedm.Problem: the form is not bound to a device problem")

    def F(self, b=None, x=None):
        if x is not None and not hasattr(x, "device"):
            self.device.set_state(u_new=x)
        Fv, _ = self.device.residual()
        if b is not None:
            b[...] = Fv.reshape(np.shape(b))
        return Fv

    def J(self, A=None, x=None):
        if x is not None and not hasattr(x, "device"):
            self.device.set_state(u_new=x)
        self.device.jacobian()
        return self.device.jacobian_csr()


class PETScSNESSolver:
    """Newton solver with the parameter names FEDM scripts set
    (fedm-streamer.py:294-299, fedm-tof.py:130-135).  ``linear_solver`` / ``preconditioner`` are
    accepted for compatibility: the device path always solves the Newton systems with restarted
    (flexible) GMRES -- preconditioned by the field split (Chebyshev sweeps on the species block
    + one multigrid V-cycle on the potential block) once ``setup_multigrid`` has installed a
    hierarchy, by point-block Jacobi otherwise -- to ``krylov_relative_tolerance``."""

    def __init__(self):
        self.parameters = {"relative_tolerance": 1e-9, "absolute_tolerance": 1e-10,
                           "solution_tolerance": 1e-16, "maximum_iterations": 50,
                           "linear_solver": "gmres", "preconditioner": "default",
                           "krylov_restart": 30, "krylov_relative_tolerance": 1e-5,
                           "krylov_maximum_iterations": 10000}

    def solve(self, problem, x=None):
        p = self.parameters
        dev = problem.device
        if getattr(problem, "before_solve", None) is not None:
            problem.before_solve()      # e.g. time-dependent Dirichlet values (functions.py:1042-1044)
        return dev.newton_solve(rtol=p["relative_tolerance"], max_it=p["maximum_iterations"],
                                atol=p["absolute_tolerance"], stol=p["solution_tolerance"],
                                ksp_restart=p["krylov_restart"],
                                ksp_rtol=p["krylov_relative_tolerance"],
                                ksp_max_it=p["krylov_maximum_iterations"])


def Poisson_solver(A, L, b, bcs, u, solver_type="mumps", preconditioner="hypre_amg"):
    """fedm/functions.py:1154-1161: assemble the right-hand side ``L`` (into ``b``), apply the Dirichlet
    values and solve ``A u = b`` -- with the host-side tensors of ``fedm_amd.forms.assemble`` (``A``
    already carries the boundary rows, as in the reference).  The solver names are accepted; the solve is
    a sparse direct one."""
    from . import forms
    b = forms.assemble(L, tensor=b)
    for bc in bcs:
        bc.apply(b)
    forms.solve(A, u.vector(), b, solver_type)
    return b


def Normal_vector(mesh):
    """Outward unit normal projected onto P1, fedm/functions.py:1133-1151: the solution of
    ``inner(u, v)*ds = inner(n, v)*ds`` over the exterior facets, interior vertices (zero rows made
    identity rows by ``ident_zeros``) zero.  Host post-processing: a boundary mass matrix of the
    boundary vertices, solved once per component.  Returns a vector Function (nodal array [vertex][2])."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from . import forms
    cell, local = mesh.exterior_facets()
    a = mesh.cells[cell, np.array([1, 0, 0])[local]].astype(np.int64)
    b = mesh.cells[cell, np.array([2, 2, 1])[local]].astype(np.int64)
    opposite = mesh.coords[mesh.cells[cell, local]]
    pa, pb = mesh.coords[a], mesh.coords[b]
    edge = pb - pa
    length = np.hypot(edge[:, 0], edge[:, 1])
    normal = np.stack([edge[:, 1], -edge[:, 0]], axis=1) / length[:, None]
    outward = np.einsum("fd,fd->f", normal, 0.5 * (pa + pb) - opposite) > 0.0
    normal[~outward] *= -1.0
    nv = mesh.num_vertices()
    rows = np.concatenate([a, a, b, b])
    cols = np.concatenate([a, b, a, b])
    vals = np.concatenate([length / 3.0, length / 6.0, length / 6.0, length / 3.0])     # P1 mass of an edge
    M = sp.coo_matrix((vals, (rows, cols)), shape=(nv, nv)).tocsr()
    on_boundary = np.zeros(nv, dtype=bool)
    on_boundary[a] = on_boundary[b] = True
    M = (M + sp.diags((~on_boundary).astype(float))).tocsc()                               # ident_zeros()
    lu = spla.splu(M)
    out = np.zeros((nv, 2))
    for d in range(2):
        rhs = np.bincount(np.concatenate([a, b]), weights=np.tile(0.5 * length * normal[:, d], 2), minlength=nv)
        out[:, d] = lu.solve(rhs)
    return forms.Function(forms.VectorFunctionSpace(mesh, "CG", 1), values=out)


def Max(a, b):
    """(a + b + |a - b|)/2, fedm/functions.py:205-209; symbolic operands: the wall flux of an ion
    species, Max(dot(Gamma, normal), 0) (fedm-gd.py:351)."""
    from . import forms, lmea
    if isinstance(a, forms.Sym) or isinstance(b, forms.Sym):
        return lmea.positive_part(a, b)
    return (a + b + abs(a - b)) / 2.0


def Min(a, b):
    from . import forms, lmea
    if isinstance(a, forms.Sym) or isinstance(b, forms.Sym):
        return lmea.positive_part(None, None)
    return (a + b - abs(a - b)) / 2.0


# ---------------------------------------------------------------------------
# fedm/functions.py:915-951 -- step-size controllers (host scalars)
# ---------------------------------------------------------------------------
def _within(dt, dt_min, dt_max):
    return max(min(dt, dt_max), dt_min)


def adaptive_timestep(dt, error, tol=1e-4, dt_min=1e-13, dt_max=1e-9):
    """PID step controller, fedm/functions.py:915-927; error = [e_n, e_{n-1}, e_{n-2}].  The
    three factors are multiplied left to right and then applied to dt: the golden error logs
    replay bit for bit only in that order."""
    e0, e1, e2 = error[0], error[1], error[2]
    proportional = (e1 / e0) ** 0.075
    integral = (tol / e0) ** 0.175
    derivative = (e1 ** 2 / (e0 * e2)) ** 0.01
    return _within(dt * (proportional * integral * derivative), dt_min, dt_max)


def adaptive_timestep_PI34(dt, error, tol=1e-4, dt_min=1e-13, dt_max=1e-9):
    """Soederlind's PI.3.4 controller, fedm/functions.py:930-937."""
    e0, e1 = error[0], error[1]
    growth = (0.8 * tol / e0) ** (0.3 / 3) * (0.8 * e1 / e0) ** (0.4 / 3)
    return _within(dt * growth, dt_min, dt_max)


def adaptive_timestep_H211b(dt, dt_old, error, tol=1e-4, dt_min=1e-13, dt_max=1e-9):
    """Soederlind's H211b digital filter, fedm/functions.py:940-951."""
    e0, e1 = error[0], error[1]
    growth = (0.8 * tol / e0) ** (1 / 12) * (0.8 * tol / e1) ** (1 / 12) * (dt / dt_old) ** (-1 / 4)
    return _within(dt * growth, dt_min, dt_max)


class ErrorGreaterThanTTOL(Exception):
    pass


# ---------------------------------------------------------------------------
# fedm/functions.py:958-1130
# ---------------------------------------------------------------------------
_ERROR_LOGS = {}


def _error_log(error_file):
    """The error log as a writable object: the open file the reference's scripts pass
    (fedm-streamer.py:276), or -- for a path -- a handle opened once in append mode and kept (opening
    the file at every time step cost 20 us of a 1.6 ms step); every row is flushed."""
    if hasattr(error_file, "write"):
        return error_file
    key = str(error_file)
    handle = _ERROR_LOGS.get(key)
    if handle is None or handle.closed:
        handle = _ERROR_LOGS[key] = open(key, "a")
    return handle


def adaptive_solver(nonlinear_solver, problem, t, dt, dt_old, u_new, u_old, var_list_new,
                    var_list_old, assigner, error, error_file, max_error, ttol, dt_min,
                    time_dependent_arguments=None, approximation="LMEA"):
    """One accepted time step with FEDM's accept/reject rule (same argument list; the reference
    recurses after a rejection, this loops).

    An attempt advances the time-dependent arguments, solves, measures the relative change of
    one field (a device reduction, functions.py:1062-1064) and appends a row to the error log.
    Rejected when that change reaches ``ttol`` (next attempt with dt * 0.5 * ttol / max_error) or
    when anything raised (dt * 0.5); below ``dt_min`` the run ends with SystemExit.
    ``dt``/``dt_old`` carry ``.time_step``; ``u_new``/``u_old`` are the device-resident states."""
    dev = problem.device
    watched = {"LMEA": 0, "LFA": dev.n_eq - 2}
    while True:
        print_rank_0(f"Attempting to solve the equation for t = {t} with dt = {dt.time_step}", flush=True)
        step = dt.time_step
        try:
            for arg in (time_dependent_arguments or ()):
                arg.t = t + step
            dev.set_step(step, dt_old.time_step)
            dev.watch_component = watched.get(approximation)   # its error norm comes with the solve
            nonlinear_solver.solve(problem, u_new.vector())
            assigner.assign(var_list_new, u_new)
            error[0] = dev.field_error(watched[approximation]) if approximation in watched else dev.state_error()
            rows = _error_log(error_file)
            rows.write(f"{error[0]:<23}  {dt_old.time_step:<23}  {step:<23}\n")
            rows.flush()
            max_error[0] = max(error)
            if error[0] >= ttol:
                raise ErrorGreaterThanTTOL
            return t + step
        except Exception as problem_found:     # the reference catches everything here as well
            t = (t + step) - step              # it advances t first and steps back: same rounding
            if isinstance(problem_found, ErrorGreaterThanTTOL):
                dt.time_step *= 0.5 * ttol / max_error[0]
                print_rank_0("Residual is greater than the prescribed tolerance. Reducing "
                             "time-step size and repeating calculation.")
            else:
                dt.time_step *= 0.5
                print_rank_0("An exception was raised while solving. Reducing time-step size "
                             "and repeating calculation.")
            if dt.time_step < dt_min:
                raise SystemExit("Minimum time-step size reached, program is terminating.")
            u_new.assign(u_old)
            assigner.assign(var_list_new, u_new)


# ---------------------------------------------------------------------------
# fedm/functions.py:531-750 -- coefficient table look-ups (nodal arrays)
# ---------------------------------------------------------------------------
def _choice(where, what, value, allowed):
    if value not in allowed:
        raise ValueError(f"fedm.{where}: {what} '{value}' not recognised. Must be one of "
                         f"{comma_separated(allowed)}.")


def _equal_lengths(where, message, reference, *others):
    if any(len(o) != len(reference) for o in others):
        raise ValueError(f"fedm.{where}: {message} must be the same length.")


# How a nodal coefficient field follows from its dependence tag.  Each rule gets the table
# (kx, ky) and a context with the nodal mean energy, reduced field, gas temperature and, for the
# Einstein relation, the mobility field; ``None`` means "leave the field as it is".
def _transport_rules(status, N0, Tgas, energy, redfield):
    table = lambda arg: (lambda kx, ky, mu: np.interp(arg(), kx, ky) / N0)
    return {
        0: None,
        "const": (lambda kx, ky, mu: ky / N0) if status == "initial" else None,
        "Umean": table(lambda: energy.vector()[:]),
        "E/N": table(lambda: redfield.vector()[:]),
        "Tgas": table(lambda: Tgas),
        "ESR": lambda kx, ky, mu: kB * Tgas * mu.vector()[:] / elementary_charge,
    }


def _rate_rules(status, energy, redfield):
    table = lambda arg: (lambda kx, ky: np.interp(arg(), kx, ky))
    return {
        0: None,
        "const": (lambda kx, ky: ky) if status == "initial" else None,
        "Umean": table(lambda: energy.vector()[:]),
        "E/N": table(lambda: redfield.vector()[:]),
        "Te": table(lambda: 2 * energy.vector()[:] / (3 * kB_eV)),
        "fun:Te,Tgas": None,          # accepted and ignored, as in the reference (functions.py:730)
        "fun:Tgas": None,
    }


def Transport_coefficient_interpolation(status, dependences, N0, Tgas, k_coeffs, kxs, kys,
                                        energy, redfield, mus=None):
    """fedm/functions.py:531-639: refresh the nodal transport coefficients (``np.interp`` with
    clamped ends, divided by N0; Einstein relation for 'ESR')."""
    me = "Transport_coefficient_interpolation"
    rules = _transport_rules(status, N0, Tgas, energy, redfield)
    _choice(me, "status", status, ["initial", "update"])
    for tag in dependences:
        _choice(me, "dependence", tag, [0, "const", "Umean", "E/N", "ESR", "Tgas"])
    if mus is None:
        if "ESR" in dependences:
            raise ValueError(f"fedm.{me}: Must provide mus (mobilities) when using ESR dependence.")
        mus = [None] * len(k_coeffs)
    _equal_lengths(me, "The lists 'dependences', 'kxs', 'kys', 'k_coeffs', and (optionally) 'mus'",
                   k_coeffs, dependences, kxs, kys, mus)
    for field, tag, kx, ky, mu in zip(k_coeffs, dependences, kxs, kys, mus):
        rule = rules[tag]
        if rule is not None:
            field.vector()[:] = rule(kx, ky, mu)


def Rate_coefficient_interpolation(status, dependences, k_coeffs, kxs, kys, energy, redfield,
                                   Te=300.0, Tgas=300.0):
    """fedm/functions.py:642-750: the same for the rate coefficients (no N0 scaling; 'Te' looks
    the table up at 2/3 of the mean energy in kelvin-equivalent electron volts)."""
    me = "Rate_coefficient_interpolation"
    rules = _rate_rules(status, energy, redfield)
    _choice(me, "status", status, ["initial", "update"])
    for tag in dependences:
        _choice(me, "dependence", tag, [0, "const", "Umean", "E/N", "Te", "fun:Te,Tgas", "fun:Tgas"])
    _equal_lengths(me, "The lists 'dependences', 'kxs', 'kys', and 'k_coeffs'", k_coeffs, dependences, kxs, kys)
    for field, tag, kx, ky in zip(k_coeffs, dependences, kxs, kys):
        rule = rules[tag]
        if rule is not None:
            field.vector()[:] = rule(kx, ky)


def semi_implicit_coefficients(dependences, mean_energy_new, mean_energy_old, coefficients,
                               coefficient_diffs):
    """fedm/functions.py:753-774: c + c' (mean_energy_new - mean_energy_old) for the entries that
    depend on the mean energy, c otherwise."""
    _equal_lengths("semi_implicit_coefficients", "The lists 'dependences', 'coefficients', and 'coefficient_diffs'",
                   dependences, coefficients, coefficient_diffs)
    from . import forms, lmea
    if any(isinstance(c, forms.Function) for c in coefficients):
        # nodal coefficients of a device model: the kernels form c + c' (eps_new - eps_old) at the
        # quadrature points themselves (csrc/gd.hip, gd_point)
        return [lmea.SemiImplicit(c, dc, tag == "Umean", mean_energy_new, mean_energy_old)
                for c, dc, tag in zip(coefficients, coefficient_diffs, dependences)]
    shift = mean_energy_new - mean_energy_old
    return [c + dc * shift if tag == "Umean" else c
            for c, dc, tag in zip(coefficients, coefficient_diffs, dependences)]


# ---------------------------------------------------------------------------
# fedm/functions.py:777-912 -- reaction source terms
# ---------------------------------------------------------------------------
def _exp(x):
    from . import forms
    return forms.exp(x)


def _nodal(k_coeffs):
    from . import forms, lmea
    return any(isinstance(k, (forms.Function, lmea.SemiImplicit)) for k in k_coeffs)


def _reaction_rates(p_matrix, densities, k_coeffs):
    """rate_j = k_j * prod_i n_i^P_ji, multiplying in species order (symbolic or numeric n_i)."""
    rates = []
    for powers, k in zip(np.asarray(p_matrix), k_coeffs):
        product = 1.0
        for n_i, p in zip(densities, powers):
            product = product * n_i ** int(p)
        rates.append(product * k)
    return rates


def Source_term(coupling, approx, p_matrix, l_matrix, g_matrix, k_coeffs, N0, u):
    """fedm/functions.py:777-843: f_i = sum_j (G - L)_ji rate_j.  The background gas (density N0)
    is species 0 of the matrices; which entries of ``u`` are particle species depends on the
    coupling (the last one is the potential when coupled) and the approximation (the first one
    is the energy in LMEA)."""
    if coupling not in ("coupled", "uncoupled"):
        raise ValueError("fedm.Source_term: coupling must be 'coupled' or 'uncoupled'.")
    if approx not in ("LFA", "LMEA"):
        raise ValueError("fedm.Source_term: approx must be 'LFA' or 'LMEA'.")
    first = 0 if (coupling, approx) == ("coupled", "LFA") else 1
    last = len(u) - (1 if coupling == "coupled" else 0)
    if _nodal(k_coeffs):      # nodal rate coefficients: the device evaluates the rates (csrc/gd.hip)
        from . import lmea
        return [lmea.LmeaSource(i, p_matrix, l_matrix, g_matrix, k_coeffs, N0)
                for i in range(np.asarray(g_matrix).shape[1])]
    rates = _reaction_rates(p_matrix, [N0] + [_exp(u[i]) for i in range(first, last)], k_coeffs)
    net = (np.asarray(g_matrix) - np.asarray(l_matrix)).astype(int)
    sources = []
    for column in net.T:
        total = 0.0
        for rate, nu in zip(rates, column):
            total = total + rate * int(nu)
        sources.append(total)
    return sources


def Energy_Source_term(coupling, p_matrix, l_matrix, g_matrix, k_coeffs, u_loss, mean_energy,
                       N0, n, Ei=0):
    """fedm/functions.py:845-912: -sum_j rate_j * loss_j, where the two sentinel loss values of
    the decks stand for (Ei - mean energy) and the mean energy itself."""
    last = len(n) - (1 if coupling == "coupled" else 0)
    if _nodal(k_coeffs):
        from . import lmea
        return lmea.LmeaEnergySource(p_matrix, k_coeffs, u_loss, N0, mean_energy=mean_energy, Ei=Ei)
    rates = _reaction_rates(p_matrix, [N0] + [_exp(n[i]) for i in range(1, last)], k_coeffs)
    total = 0.0
    for rate, loss in zip(rates, u_loss):
        if 7e77 < loss < 8e77:
            factor = Ei - mean_energy
        elif 9e99 < loss < 1e100:
            factor = mean_energy
        else:
            factor = loss
        total = total + (-rate) * factor
    return total
