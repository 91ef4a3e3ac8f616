"""FEDM's ``fedm.functions`` surface on top of the MI355X device path.

Same function names, argument order and error behaviour as the reference
(fedm/functions.py); the objects that are UFL/DOLFIN types there are light
descriptors here (see :mod:`fedm_amd.forms`) because DOLFIN does not exist on
the GPU box.  The hot path -- ``Problem.F``, ``Problem.J`` and
``nonlinear_solver.solve`` -- runs in libfedm_hip.so.
"""
import warnings
from pathlib import Path
from textwrap import dedent
from typing import Any, List, Optional, Tuple

import numpy as np

from .mesh import Marking_boundaries  # noqa: F401  (fedm/functions.py:86)
from .physical_constants import elementary_charge, kB, kB_eV
from .utils import comma_separated, print_rank_0

DOLFIN_EPS = 3.0e-16


# ---------------------------------------------------------------------------
# fedm/functions.py:15-45
# ---------------------------------------------------------------------------
def modify_approximation_vars(approximation_type, number_of_species, particle_species,
                              masses, charges):
    """LFA drops the first species; n_eq = n_species + 1.  Mutates the lists like the
    reference does."""
    approximation_types = ["LFA", "LMEA"]
    if approximation_type not in approximation_types:
        err_msg = dedent(
            f"""\
            fedm.modify_approximation_vars: The approximation type {approximation_type}
            is not recognised. Must be one of {comma_separated(approximation_types)}.
            """
        )
        raise ValueError(err_msg.rstrip().replace("\n", " "))
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
        err_msg = dedent(
            f"""\
            fedm.Function_definition: Invalid function_type '{function_type}'.
            Possible values are {comma_separated(functions)}.
            """
        )
        raise ValueError(err_msg.rstrip().replace("\n", " "))
    function = functions[function_type]
    return [function(function_space) for _ in range(eq_number)]


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
    equation_types = ["reaction", "diffusion-reaction", "drift-diffusion-reaction"]
    if equation_type not in equation_types:
        err_msg = dedent(
            f"""\
            fedm.weak_form_balance_equation_log_representation: The equation type
            {equation_type}' is not recognised. Must be one of
            {comma_separated(equation_types)}.
            """
        )
        raise ValueError(err_msg.rstrip().replace("\n", " "))
    if equation_type == "diffusion-reaction" and D is None:
        raise ValueError(
            "fedm.weak_form_balance_equation_log_representation: When 'equation_type' "
            "is diffusion-reaction, must also supply the diffusion coefficient 'D'."
        )
    if not log_representation:
        raise NotImplementedError(
            "the device path implements the logarithmic representation "
            "(weak_form_balance_equation_log_representation), the one every FEDM example uses")
    return BalanceEq(equation_type=equation_type, dt=dt, dt_old=dt_old, dx=dx, u=u, u_old=u_old,
                     u_old1=u_old1, v=v, f=f, Gamma=Gamma, r=r, D=D)


def weak_form_balance_equation_log_representation(*args, **kwargs):
    return weak_form_balance_equation(*args, **kwargs, log_representation=True)


def weak_form_Poisson_equation(dx, u, v, f, r=0.5 / np.pi):
    """2*pi*r*(inner(grad(u), grad(v)) - f*v)*dx, fedm/functions.py:379-401."""
    return PoissonEq(dx=dx, u=u, v=v, f=f, r=r)


def Boundary_flux(bc_type, equation_type, particle_type, sign, mu, E, normal, u, gamma, v,
                  ds_temp, r=0.5 / np.pi, vth=0.0, ref=1.0, Ion_flux=0.0):
    """Boundary terms, fedm/functions.py:404-528: same checks, warning and return
    convention (a form piece, or 0.0 when the condition contributes nothing)."""
    bc_types = ["zero flux", "flux source", "Neumann"]
    equation_types = ["reaction", "diffusion-reaction", "drift-diffusion-reaction"]
    particle_types = ["Heavy", "electrons"]
    if "_" in bc_type:
        warnings.warn("fedm.BoundaryFlux: bc_type should have spaces, not underscores")
        bc_type = bc_type.replace("_", " ")
    if bc_type not in bc_types:
        err_msg = dedent(
            f"""\
            fedm.Boundary_flux: boundary condition type '{bc_type}' not recognised.
            Must be one of {comma_separated(bc_types)}.
            """
        )
        raise ValueError(err_msg.rstrip().replace("\n", " "))
    if bc_type != "zero flux" and equation_type not in equation_types:
        err_msg = dedent(
            f"""\
            fedm.Boundary_flux: equation type '{equation_type}' not recognised.
            Must be one of {comma_separated(equation_types)}.
            """
        )
        raise ValueError(err_msg.rstrip().replace("\n", " "))
    if (bc_type == "flux source" and equation_type == "diffusion-reaction"
            and particle_type not in particle_types):
        err_msg = dedent(
            f"""\
            fedm.Boundary_flux: particle type '{particle_type}' not recognised.
            Must be one of {comma_separated(particle_types)}.
            """
        )
        raise ValueError(err_msg.rstrip())
    if bc_type == "flux source" and equation_type != "reaction":
        raise NotImplementedError(
            "'flux source' boundaries (glow-discharge model) are not on the device path yet")
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


def compile_forms(F, quadrature_degree=None):
    """FormSum -> (device Model, mesh, facet tags).  The compiler of this façade: what
    FFC does for the reference, restricted to the model family of the device kernels."""
    from . import forms
    from .device import Model, Reaction
    from .physical_constants import elementary_charge as q_e, epsilon_0 as eps0
    from .termsum import TermSum
    pieces = F.pieces if isinstance(F, FormSum) else [F]
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
    for s, p in enumerate(balances):
        if p.equation_type == "drift-diffusion-reaction":
            g = p.Gamma
            if not isinstance(g, FluxDesc):
                raise ValueError("drift-diffusion-reaction needs Gamma = Flux(...)")
            mu[s], D[s], Z[s] = TermSum.coerce(g.mu), TermSum.coerce(g.D), float(g.sign)
        elif p.equation_type == "diffusion-reaction":
            D[s] = TermSum.coerce(p.D)
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
    if qd is None or qd < 0:
        raise NotImplementedError(
            "set parameters['form_compiler']['quadrature_degree'] (UFL's automatic degree "
            "estimation is reproduced only for the time-of-flight case, see DESIGN.md)")
    model = Model(n_species=ns, poisson=bool(poissons), eq_type=eq_type, Z=Z, mu=mu, D=D,
                  reactions=reactions, bc_kind=bc_kind, quadrature_degree=int(qd),
                  axisymmetric=axis)
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
        if device_problem is None and isinstance(F, (FormSum, FormPiece)):
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
    (fedm-streamer.py:294-299, fedm-tof.py:130-135).  ``linear_solver`` is accepted
    for compatibility; the device path always runs GMRES + point-block Jacobi."""

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


def Max(a, b):
    return (a + b + abs(a - b)) / 2.0


def Min(a, b):
    return (a + b - abs(a - b)) / 2.0


# ---------------------------------------------------------------------------
# fedm/functions.py:915-951 -- step-size controllers (host scalars)
# ---------------------------------------------------------------------------
def adaptive_timestep(dt, error, tol=1e-4, dt_min=1e-13, dt_max=1e-9):
    dt *= (
        (error[1] / error[0]) ** 0.075
        * (tol / error[0]) ** 0.175
        * (error[1] ** 2 / (error[0] * error[2])) ** 0.01
    )
    return max(min(dt, dt_max), dt_min)


def adaptive_timestep_PI34(dt, error, tol=1e-4, dt_min=1e-13, dt_max=1e-9):
    dt *= (0.8 * tol / error[0]) ** (0.3 / 3) * (0.8 * error[1] / error[0]) ** (0.4 / 3)
    return max(min(dt, dt_max), dt_min)


def adaptive_timestep_H211b(dt, dt_old, error, tol=1e-4, dt_min=1e-13, dt_max=1e-9):
    dt *= (
        (0.8 * tol / error[0]) ** (1 / 12)
        * (0.8 * tol / error[1]) ** (1 / 12)
        * (dt / dt_old) ** (-1 / 4)
    )
    return max(min(dt, dt_max), dt_min)


class ErrorGreaterThanTTOL(Exception):
    pass


# ---------------------------------------------------------------------------
# fedm/functions.py:958-1130
# ---------------------------------------------------------------------------
def adaptive_solver(nonlinear_solver, problem, t, dt, dt_old, u_new, u_old, var_list_new,
                    var_list_old, assigner, error, error_file, max_error, ttol, dt_min,
                    time_dependent_arguments=None, approximation="LMEA"):
    """One accepted time step with FEDM's accept/reject rule.

    Same argument list as the reference.  ``dt``/``dt_old`` carry ``.time_step``;
    ``u_new``/``u_old`` are the device-resident mixed states; the error norm of
    functions.py:1062-1064 is a device reduction over one component."""
    print_rank_0(
        f"Attempting to solve the equation for t = {t} with dt = {dt.time_step}",
        flush=True,
    )
    dev = problem.device
    try:
        t += dt.time_step
        if time_dependent_arguments is not None:
            for arg in time_dependent_arguments:
                arg.t = t
        dev.set_step(dt.time_step, dt_old.time_step)
        nonlinear_solver.solve(problem, u_new.vector())
        assigner.assign(var_list_new, u_new)
        if approximation == "LMEA" or approximation == "LFA":
            idx = 0 if approximation == "LMEA" else dev.n_eq - 2
            error[0] = dev.field_error(idx)
        else:
            error[0] = dev.state_error()
        with open(error_file, "a") as f_err:
            f_err.write(f"{error[0]:<23}  {dt_old.time_step:<23}  {dt.time_step:<23}\n")
            f_err.flush()
        max_error[0] = max(error)
        if error[0] >= ttol:
            raise ErrorGreaterThanTTOL
    except Exception as exc:
        t -= dt.time_step
        if isinstance(exc, ErrorGreaterThanTTOL):
            dt.time_step *= 0.5 * ttol / max_error[0]
            print_rank_0(
                "Residual is greater than the prescribed tolerance. Reducing "
                "time-step size and repeating calculation."
            )
        else:
            dt.time_step *= 0.5
            print_rank_0(
                "An exception was raised while solving. Reducing time-step size "
                "and repeating calculation."
            )
        if dt.time_step < dt_min:
            raise SystemExit("Minimum time-step size reached, program is terminating.")
        u_new.assign(u_old)
        assigner.assign(var_list_new, u_new)
        t = adaptive_solver(nonlinear_solver, problem, t, dt, dt_old, u_new, u_old,
                            var_list_new, var_list_old, assigner, error, error_file,
                            max_error, ttol, dt_min, time_dependent_arguments, approximation)
    return t


# ---------------------------------------------------------------------------
# fedm/functions.py:531-750 -- coefficient table look-ups (nodal arrays)
# ---------------------------------------------------------------------------
def Transport_coefficient_interpolation(status, dependences, N0, Tgas, k_coeffs, kxs, kys,
                                        energy, redfield, mus=None):
    possible_statuses = ["initial", "update"]
    possible_dependences = [0, "const", "Umean", "E/N", "ESR", "Tgas"]
    if status not in possible_statuses:
        err_msg = dedent(
            f"""\
            fedm.Transport_coefficient_interpolation: status '{status}' not recognised.
            Must be one of {comma_separated(possible_statuses)}.
            """
        )
        raise ValueError(err_msg.rstrip().replace("\n", " "))
    for dependence in dependences:
        if dependence not in possible_dependences:
            err_msg = dedent(
                f"""\
                fedm.Transport_coefficient_interpolation: dependence '{dependence}' not
                recognised. Must be one of {comma_separated(possible_dependences)}.
                """
            )
            raise ValueError(err_msg.rstrip().replace("\n", " "))
    if mus is None:
        if "ESR" in dependences:
            raise ValueError(
                "fedm.Transport_coefficient_interpolation: Must provide mus "
                "(mobilities) when using ESR dependence."
            )
        mus = [None] * len(k_coeffs)
    if not all([len(x) == len(k_coeffs) for x in [dependences, kxs, kys, mus]]):
        raise ValueError(
            "fedm.Transport_coefficient_interpolation: The lists 'dependences', 'kxs', "
            "'kys', 'k_coeffs', and (optionally) 'mus' must be the same length."
        )
    for k_coeff, dependence, kx, ky, mu in zip(k_coeffs, dependences, kxs, kys, mus):
        if dependence == "const" and status == "initial":
            k_coeff.vector()[:] = ky / N0
        elif dependence == "Umean":
            k_coeff.vector()[:] = np.interp(energy.vector()[:], kx, ky) / N0
        elif dependence == "E/N":
            k_coeff.vector()[:] = np.interp(redfield.vector()[:], kx, ky) / N0
        elif dependence == "ESR":
            k_coeff.vector()[:] = kB * Tgas * mu.vector()[:] / elementary_charge
        elif dependence == "Tgas":
            k_coeff.vector()[:] = np.interp(Tgas, kx, ky) / N0


def Rate_coefficient_interpolation(status, dependences, k_coeffs, kxs, kys, energy, redfield,
                                   Te=300.0, Tgas=300.0):
    possible_statuses = ["initial", "update"]
    possible_dependences = [0, "const", "Umean", "E/N", "Te", "fun:Te,Tgas", "fun:Tgas"]
    if status not in possible_statuses:
        raise ValueError(
            f"fedm.Rate_coefficient_interpolation: status '{status}' not recognised. "
            f"Must be one of {comma_separated(possible_statuses)}."
        )
    for dependence in dependences:
        if dependence not in possible_dependences:
            raise ValueError(
                f"fedm.Rate_coefficient_interpolation: dependence '{dependence}' not "
                f"recognised. Must be one of {comma_separated(possible_dependences)}."
            )
    if not all([len(x) == len(k_coeffs) for x in [dependences, kxs, kys]]):
        raise ValueError(
            "fedm.Rate_coefficient_interpolation: The lists 'dependences', 'kxs', "
            "'kys', and 'k_coeffs' must be the same length."
        )
    for k_coeff, dependence, kx, ky in zip(k_coeffs, dependences, kxs, kys):
        if dependence == "const" and status == "initial":
            k_coeff.vector()[:] = ky
        elif dependence == "Te":
            k_coeff.vector()[:] = np.interp(2 * energy.vector()[:] / (3 * kB_eV), kx, ky)
        elif dependence == "Umean":
            k_coeff.vector()[:] = np.interp(energy.vector()[:], kx, ky)
        elif dependence == "E/N":
            k_coeff.vector()[:] = np.interp(redfield.vector()[:], kx, ky)
        # 'fun:...' is unreachable in the reference as well (functions.py:730)


def semi_implicit_coefficients(dependences, mean_energy_new, mean_energy_old, coefficients,
                               coefficient_diffs):
    if not all([len(x) == len(dependences) for x in [coefficients, coefficient_diffs]]):
        raise ValueError(
            "fedm.semi_implicit_coefficients: The lists 'dependences', 'coefficients', "
            "and 'coefficient_diffs' must be the same length."
        )
    si_coefficients = []
    for coeff, diff, dep in zip(coefficients, coefficient_diffs, dependences):
        if dep == "Umean":
            si_coefficients.append(coeff + diff * (mean_energy_new - mean_energy_old))
        else:
            si_coefficients.append(coeff)
    return si_coefficients


# ---------------------------------------------------------------------------
# fedm/functions.py:777-912 -- reaction source terms
# ---------------------------------------------------------------------------
def _exp(x):
    from . import forms
    return forms.exp(x)


def Source_term(coupling, approx, p_matrix, l_matrix, g_matrix, k_coeffs, N0, u):
    couplings = ["coupled", "uncoupled"]
    approximations = ["LFA", "LMEA"]
    if coupling not in couplings:
        raise ValueError("fedm.Source_term: coupling must be 'coupled' or 'uncoupled'.")
    if approx not in approximations:
        raise ValueError("fedm.Source_term: approx must be 'LFA' or 'LMEA'.")
    start = 0 if coupling == "coupled" and approx == "LFA" else 1
    end = len(u) - 1 if coupling == "coupled" else len(u)
    exp_u = [N0] + [_exp(u[i]) for i in range(start, end)]
    p_matrix = np.asarray(p_matrix)
    rate = []
    for j in range(p_matrix.shape[0]):
        temp = 1.0
        for i in range(p_matrix.shape[1]):
            temp = temp * exp_u[i] ** int(p_matrix[j, i])
        rate.append(temp * k_coeffs[j])
    gl = np.asarray(g_matrix) - np.asarray(l_matrix)
    f_temp = []
    for i in range(gl.shape[1]):
        acc = 0.0
        for j in range(gl.shape[0]):
            acc = acc + rate[j] * int(gl[j, i])
        f_temp.append(acc)
    return f_temp


def Energy_Source_term(coupling, p_matrix, l_matrix, g_matrix, k_coeffs, u_loss, mean_energy,
                       N0, n, Ei=0):
    neq = len(n) - 1 if coupling == "coupled" else len(n)
    exp_u = [N0] + [_exp(n[i]) for i in range(1, neq)]
    p_matrix = np.asarray(p_matrix)
    total = 0.0
    for idx, loss in enumerate(u_loss):
        temp = 1.0
        for i in range(p_matrix.shape[1]):
            temp = temp * exp_u[i] ** int(p_matrix[idx, i])
        rate = -temp * k_coeffs[idx]
        if loss > 7e77 and loss < 8e77:
            rate = rate * (Ei - mean_energy)
        elif loss > 9e99 and loss < 1e100:
            rate = rate * mean_energy
        else:
            rate = rate * loss
        total = total + rate
    return total
