#!/usr/bin/env python3
"""Abnormal glow discharge in argon at 1 Torr between plane electrodes 1 cm apart: electron energy
balance + Ar*, Ar+ and electron balances + Poisson, local mean energy approximation (LMEA), with
tabulated transport and rate coefficients and a reaction scheme read from the FEDM deck.

A driver of our own for the fedm_amd facade.  It solves the case of the reference's
examples/glow_discharge/fedm-gd.py (deck `4_particles`, 100 x 100 crossed mesh, -250 V switched on
with a 1 ns rise, reflecting metallic electrodes with secondary emission), and the run reproduces
the reference's golden error log and species snapshots (tests/test_gpu_glow_discharge.py).  That the
reference's own script lowers to the same device model is checked in tests/test_reference_scripts.py
in the build container.

The nodal coefficients are refreshed once per time step on the host with
`Transport_coefficient_interpolation` / `Rate_coefficient_interpolation` (as a FEDM script does); the
Newton solves run on the device.  `fedm_amd.cases.glow_discharge.Case` is the variant that keeps the
whole per-step pipeline on the GPU.

    python examples/glow_discharge.py gd_output
"""
import contextlib
import io
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd import forms as fem                          # noqa: E402  (what `dolfin` is to a FEDM script)
from fedm_amd import file_io, functions as fedm            # noqa: E402
from fedm_amd.physical_constants import elementary_charge, epsilon_0, kB     # noqa: E402

REPO = Path(__file__).resolve().parent.parent
DECK = "4_particles"
GAP = RADIUS = 0.01                       # electrode distance and discharge radius [m]
PRESSURE, GAS_TEMPERATURE, VOLTAGE = 1.0, 300.0, -250.0
GAS_DENSITY = PRESSURE * 3.21877e22
# the four sides in the order of the boundary tags: powered electrode (z = 0), grounded electrode
# (z = gap), axis, dielectric outer wall
SIDES = [["line", 0.0, 0.0, 0.0, RADIUS], ["line", GAP, GAP, 0.0, RADIUS],
         ["line", 0.0, GAP, 0.0, 0.0], ["line", 0.0, GAP, RADIUS, RADIUS]]
METAL, OPEN = [0.3, 0.3, 5e-4, 0.3], [1.0, 1.0, 1.0, 1.0]     # reflection coefficients per species
REFLECTION = [METAL, METAL, OPEN, OPEN]
SECONDARY_EMISSION = [0.06, 0.06, 0, 0]
SECONDARY_ENERGY = 5.0                    # mean energy of the emitted electrons [eV]
KINDS = ["reaction", "diffusion-reaction", "drift-diffusion-reaction", "drift-diffusion-reaction"]
WALL_ROLE = ["Heavy", "Heavy", "Heavy", "electrons"]
CHARGE_ROLE = ["Neutral", "Neutral", "Ion", "electrons"]
START_DENSITY = [GAS_DENSITY, 1e12, 1e12, 1e12]
OUTPUT_WINDOWS = [1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5]
OUTPUT_STRIDES = [1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-6]


def read_deck(input_dir):
    """Species, reaction scheme and coefficient tables; derivatives of the energy-dependent tables
    (needed by the semi-implicit treatment) by finite differences of the tables."""
    file_io.files.file_input = Path(input_dir) if input_dir else REPO / "decks" / "glow_discharge" / "file_input"
    deck_dir = file_io.files.file_input / DECK
    d = {}
    n, d["names"], property_files, d["file_names"] = file_io.read_speclist(deck_dir)
    d["masses"], d["charge_numbers"] = file_io.read_particle_properties(property_files, DECK)
    d["power"], d["loss"], d["gain"] = file_io.reaction_matrices(deck_dir, d["names"])
    rate_files = file_io.rate_coefficient_file_names(deck_dir)
    d["energy_loss"] = file_io.read_energy_loss(deck_dir)
    d["mu_x"], d["mu_y"], d["mu_dep"] = file_io.read_transport_coefficients(d["file_names"], "mobility", DECK)
    d["D_x"], d["D_y"], d["D_dep"] = file_io.read_transport_coefficients(d["file_names"], "Diffusion", DECK)
    d["k_dep"] = file_io.read_dependences(rate_files)
    d["k_x"], d["k_y"] = file_io.read_rate_coefficients(rate_files, d["k_dep"])
    e = n - 1                                                  # the electrons are the last species
    d["dD_e"] = np.gradient(d["D_y"][e], d["D_x"][e]) / GAS_DENSITY
    d["dmu_e"] = np.gradient(d["mu_y"][e], d["mu_x"][e]) / GAS_DENSITY
    d["dk"] = [np.gradient(ky, kx) if dep == "Umean" else 0.0 for kx, ky, dep in zip(d["k_x"], d["k_y"], d["k_dep"])]
    d["n_species"], d["n_equations"], d["names"], d["masses"], d["charge_numbers"] = fedm.modify_approximation_vars(
        "LMEA", n, d["names"], d["masses"], d["charge_numbers"])
    d["n_reactions"] = len(rate_files)
    return d


def reduced_field(potential):
    """|E|/N in Td as a nodal field: the argument of the field-dependent tables."""
    strength = fem.sqrt(fem.dot(-fem.grad(potential), -fem.grad(potential)))
    return fem.project(1e21 * strength / GAS_DENSITY, solver_type="mumps")


class Discharge:
    """Spaces, Functions and nodal coefficient fields of the model."""

    def __init__(self, mesh, deck):
        ns, neq, nr = deck["n_species"], deck["n_equations"], deck["n_reactions"]
        lagrange = fem.FiniteElement("Lagrange", mesh.ufl_cell(), 1)
        self.mixed = fem.FunctionSpace(mesh, fem.MixedElement(fedm.Mixed_element_list(neq, lagrange)))
        self.scalar = V = fem.FunctionSpace(mesh, lagrange)
        parts = fedm.Function_space_list(neq, V)
        self.split, self.join = fem.FunctionAssigner(parts, self.mixed), fem.FunctionAssigner(self.mixed, parts)
        self.trial, self.tests = fem.TrialFunction(self.mixed), fem.TestFunctions(self.mixed)
        self.now, self.before, self.before2 = (fem.Function(self.mixed) for _ in range(3))
        self.phi, self.phi_before, self.phi_before2 = (fem.Function(V) for _ in range(3))
        self.field, self.field_before = fem.Function(V), fem.Function(V)
        self.dens_now, self.dens_before, self.dens_before2 = (fedm.Function_definition(V, "Function", ns) for _ in range(3))
        self.energy, self.energy_before, self.energy_before2 = (fem.Function(V) for _ in range(3))
        self.D, self.dD, self.mu, self.dmu = (fedm.Function_definition(V, "Function", ns) for _ in range(4))
        self.k, self.dk = (fedm.Function_definition(V, "Function", nr) for _ in range(2))
        self.scratch = fem.Function(V)

    def start(self, deck, writers, t):
        """Uniform start: 3 eV mean energy, 1e12 m^-3 of every charged or excited species."""
        V, e = self.scalar, deck["n_species"] - 1
        three_ev = fem.interpolate(fem.Expression("3.0", degree=1), V)
        self.energy.assign(three_ev)
        self.energy_before.assign(three_ev)
        self.energy_before2.assign(fem.Constant(0.0))
        for k, n0 in enumerate(START_DENSITY):
            ln_n = fem.Expression("std::log(ic)", ic=n0, degree=1)
            self.dens_now[k].assign(ln_n)
            self.dens_before[k].assign(ln_n)
            self.dens_before2[k].assign(fem.Constant(0.0))
        # the energy equation is solved for ln(n_e * mean energy)
        ln_energy_density = fem.Expression("std::log(a) + b", a=self.energy, b=self.dens_before[e], degree=1)
        self.w_now, self.w_before = fem.interpolate(ln_energy_density, V), fem.interpolate(ln_energy_density, V)
        self.w_before2 = fem.interpolate(fem.Constant(0.0), V)
        for k, name in enumerate(deck["file_names"]):
            self.scratch.assign(self.dens_before[k])
            self.scratch.rename(name, str(k))
            writers["density"][k].write_checkpoint(self.scratch, name, t * 1e6, fem.XDMFFile.Encoding.HDF5, False)

    def refresh_tables(self, deck, status):
        """Nodal transport and rate coefficients from the tables, at the mean energy / reduced field of
        the last accepted step; on updates also the table slopes the semi-implicit terms multiply."""
        energy = self.energy if status == "initial" else self.energy_before
        first = (self.mu,) if status == "initial" else ()
        fedm.Transport_coefficient_interpolation(status, deck["mu_dep"], GAS_DENSITY, GAS_TEMPERATURE, self.mu,
                                                 deck["mu_x"], deck["mu_y"], energy, self.field, *first)
        fedm.Transport_coefficient_interpolation(status, deck["D_dep"], GAS_DENSITY, GAS_TEMPERATURE, self.D,
                                                 deck["D_x"], deck["D_y"], energy, self.field, self.mu)
        fedm.Rate_coefficient_interpolation(status, deck["k_dep"], self.k, deck["k_x"], deck["k_y"], energy,
                                            self.field, Te=0, Tgas=0)
        if status == "initial":
            return
        e, at = deck["n_species"] - 1, self.energy_before.vector()[:]
        for j, dep in enumerate(deck["k_dep"]):
            if dep == "Umean":
                self.dk[j].vector()[:] = np.interp(at, deck["k_x"][j], deck["dk"][j])
        self.dmu[e].vector()[:] = np.interp(at, deck["mu_x"][e], deck["dmu_e"])
        self.dD[e].vector()[:] = np.interp(at, deck["D_x"][e], deck["dD_e"])


def electrode_conditions(space, powered_value):
    def powered(x, on_boundary):
        return on_boundary and fem.near(x[1], 0, fem.DOLFIN_EPS)

    def grounded(x, on_boundary):
        return on_boundary and fem.near(x[1], GAP, fem.DOLFIN_EPS)
    return [fem.DirichletBC(space, powered_value, powered), fem.DirichletBC(space, fem.Constant(0.0), grounded)]


def starting_potential(dis, deck, r, voltage, writers, t):
    """Poisson's equation for the uniform start (zero net charge) with the electrode values at t = 0."""
    charge = sum(elementary_charge * z * fem.exp(n) for z, n in zip(deck["charge_numbers"], dis.dens_before))
    form = fedm.weak_form_Poisson_equation(fem.dx, fem.TrialFunction(dis.scalar), fem.TestFunction(dis.scalar),
                                           charge / (fem.Constant(1.0) * epsilon_0), r)
    conditions = electrode_conditions(dis.scalar, voltage)
    matrix, load = fem.assemble(fem.lhs(form), tensor=None), fem.assemble(fem.rhs(form), tensor=None)
    for condition in conditions:
        condition.apply(matrix)
        condition.apply(load)
    fem.solve(matrix, dis.phi.vector(), load, "mumps")
    dis.phi_before2.assign(dis.phi_before)
    dis.phi_before.assign(dis.phi)
    dis.scratch.assign(dis.phi)
    dis.scratch.rename("Phi", str(0))
    writers["potential"] << (dis.scratch, t)


def coupled_form(dis, deck, mesh, r, dt, dt_before, wall_tags):
    """Energy balance (row 0), particle balances (rows 1..3), Poisson (row 4), in logarithmic variables,
    with the coefficients linearised around the last step's mean energy (semi-implicit)."""
    ns, neq = deck["n_species"], deck["n_equations"]
    e, i_phi = ns - 1, neq - 1
    u, v, Z = dis.trial, dis.tests, deck["charge_numbers"]
    ds = fem.Measure("ds", domain=mesh, subdomain_data=wall_tags)
    outward = fem.FacetNormal(mesh)
    E = -fem.grad(u[i_phi])
    # mean energy of the new step from the unknowns, linearised: eps_old + (w - n_e eps_old) / n_e,old
    energy_new = dis.energy_before + (fem.exp(u[0]) - fem.exp(u[e]) * dis.energy_before) / fem.exp(dis.dens_before[e])
    k = fedm.semi_implicit_coefficients(deck["k_dep"], energy_new, dis.energy_before, dis.k, dis.dk)
    mu = fedm.semi_implicit_coefficients(deck["mu_dep"], energy_new, dis.energy_before, dis.mu, dis.dmu)
    D = fedm.semi_implicit_coefficients(deck["D_dep"], energy_new, dis.energy_before, dis.D, dis.dD)
    gradient_of_nD = [role == "electrons" for role in CHARGE_ROLE]            # electrons: flux = -grad(D n)

    def drift_diffusion(j, unknown, scale=1.0):
        d_j, mu_j = (D[j], mu[j]) if scale == 1.0 else (scale * D[j] / 3.0, scale * mu[j] / 3.0)
        return fedm.Flux(Z[j], unknown, d_j, mu_j, E, grad_diffusion=gradient_of_nD[j], logarithm_representation=True)

    fluxes = [0] + [drift_diffusion(j, u[j]) for j in range(1, ns)]
    ions_to_wall = sum(fedm.Max(fem.dot(fluxes[j], outward), 0) for j in range(1, ns) if CHARGE_ROLE[j] == "Ion")
    energy_flux = drift_diffusion(e, u[0], scale=5.0)
    thermal_speed = [0] + [np.sqrt(8.0 * kB * GAS_TEMPERATURE / (fem.pi * deck["masses"][j])) for j in range(1, e)]
    thermal_speed.append(fem.sqrt(16.0 * elementary_charge * dis.energy / (3.0 * fem.pi * deck["masses"][e])))

    sources = fedm.Source_term("coupled", "LMEA", deck["power"], deck["loss"], deck["gain"], k, GAS_DENSITY, u)
    heating = fedm.Energy_Source_term("coupled", deck["power"], deck["loss"], deck["gain"], k, deck["energy_loss"],
                                      u[0] / u[e], GAS_DENSITY, u)
    heating += -fem.dot(drift_diffusion(e, u[e]), E)                          # Joule heating

    form = 0
    for j in range(1, ns):
        form += fedm.weak_form_balance_equation_log_representation(
            KINDS[j], dt, dt_before, fem.dx, u[j], dis.before[j], dis.before2[j], v[j], sources[j], fluxes[j], r, D[j])
    for tag, (reflect, emit) in enumerate(zip(REFLECTION, SECONDARY_EMISSION), start=1):
        for j in range(1, ns):
            form += fedm.Boundary_flux("flux source", KINDS[j], WALL_ROLE[j], Z[j], mu[j], E, outward, u[j], emit,
                                       v[j], ds(tag), r, thermal_speed[j], reflect[j], ions_to_wall)
    energy_form = fedm.weak_form_balance_equation_log_representation(
        KINDS[e], dt, dt_before, fem.dx, u[0], dis.before[0], dis.before2[0], v[0], heating, energy_flux, r)
    emitted_energy = fem.Expression("u_p", u_p=SECONDARY_ENERGY, degree=1)
    for tag, (reflect, emit) in enumerate(zip(REFLECTION, SECONDARY_EMISSION), start=1):
        energy_form += fedm.Boundary_flux("flux source", KINDS[e], WALL_ROLE[e], Z[e], 5.0 * mu[e] / 3.0, E, outward,
                                          u[0], emit * emitted_energy, v[0], ds(tag), r, 1.3333 * thermal_speed[e],
                                          reflect[e], ions_to_wall)
    form += energy_form
    space_charge = sum(elementary_charge * z * fem.exp(u[j]) for j, z in enumerate(Z))
    form += fedm.weak_form_Poisson_equation(fem.dx, u[i_phi], v[i_phi], space_charge / (fem.Constant(1.0) * epsilon_0), r)
    return form


def main(nx=100, ny=100, T_final=1e-11, input_dir=None, output_dir="gd_output", quiet=True, ttol=2e-3,
         stop_before_device=None):
    fem.parameters["form_compiler"]["quadrature_degree"] = 4
    deck = read_deck(input_dir)
    file_io.files.output_folder_path = Path(output_dir)
    ns, neq = deck["n_species"], deck["n_equations"]
    writers = {"density": file_io.output_files("xdmf", "number density", deck["file_names"]),
               "potential": file_io.output_files("pvd", "potential", ["Phi"])[0]}
    t = 0.0
    dt = fem.Expression("time_step", time_step=1e-13, degree=0)
    dt_before = fem.Expression("time_step", time_step=1e30, degree=0)
    dt_limits = dict(dt_min=1e-15, dt_max=1e-8)
    r = fem.Expression("x[0]", degree=1)
    file_io.log("conditions", file_io.files.model_log, dt.time_step, VOLTAGE, PRESSURE, GAP, GAS_DENSITY, GAS_TEMPERATURE)
    file_io.log("properties", file_io.files.model_log, "Ar", DECK, deck["file_names"], deck["masses"],
                [z * elementary_charge for z in deck["charge_numbers"]])

    mesh = fem.RectangleMesh((0, 0), (RADIUS, GAP), nx, ny, "crossed")
    wall_tags = fedm.Marking_boundaries(mesh, SIDES)
    file_io.log("matrices", file_io.files.model_log, deck["gain"], deck["loss"], deck["power"])
    file_io.log("initial time", file_io.files.model_log, t)
    dis = Discharge(mesh, deck)
    dis.start(deck, writers, t)
    voltage = fem.Expression("U0*(1-exp(-t/1e-9))", U0=VOLTAGE, t=t, pi=fem.pi, degree=0)    # switched on with a 1 ns rise
    starting_potential(dis, deck, r, voltage, writers, t)
    dis.field.assign(reduced_field(dis.phi))
    dis.field_before.assign(dis.field)
    dis.refresh_tables(deck, "initial")

    form = coupled_form(dis, deck, mesh, r, dt, dt_before, wall_tags)
    parts_now = [dis.w_now] + dis.dens_now[1:] + [dis.phi]
    parts_before = [dis.w_before] + dis.dens_before[1:] + [dis.phi_before]
    parts_before2 = [dis.w_before2] + dis.dens_before2[1:] + [dis.phi_before2]
    for mixed, parts in ((dis.now, parts_now), (dis.before, parts_before), (dis.before2, parts_before2)):
        dis.join.assign(mixed, parts)
    residual = fem.action(form, dis.now)
    jacobian = fem.derivative(residual, dis.now, dis.trial)
    problem = (stop_before_device or fedm.Problem)(jacobian, residual, electrode_conditions(dis.mixed.sub(neq - 1), voltage))

    # the device's linear solver (PETSc's defaults in a FEDM script): multigrid on the potential block,
    # Chebyshev(8) sweeps on the species block (tools/gd_cycle.py)
    from fedm_amd.device import chebyshev_weights
    problem.device.setup_multigrid(nu=1)
    problem.device.set_fieldsplit(chebyshev_weights(8, 0.3, 2.2))
    newton = fedm.PETScSNESSolver()
    newton.parameters.update(relative_tolerance=1e-4, linear_solver="mumps")     # iteration limit: the solver's default

    out_files = [writers["potential"]] + writers["density"][1:]
    out_names = ["Phi"] + deck["file_names"][1:]
    out_now, out_before = [dis.phi] + dis.dens_now[1:], [dis.phi_before] + dis.dens_before[1:]
    window, stride = OUTPUT_WINDOWS[0], OUTPUT_STRIDES[0]
    step_errors, recent_errors, steps = [0.0] * (ns + 1), [1] * 3, 0
    while t < T_final:
        t_before = t
        dis.before2.assign(dis.before)
        dis.before.assign(dis.now)
        dis.split.assign(parts_before, dis.before)
        dis.field_before.assign(dis.field)
        dis.energy_before2.assign(dis.energy_before)
        dis.energy_before.assign(dis.energy)
        dis.field.assign(reduced_field(dis.phi))
        dis.refresh_tables(deck, "update")
        with contextlib.redirect_stdout(io.StringIO() if quiet else sys.stdout):
            t = fedm.adaptive_solver(newton, problem, t, dt, dt_before, dis.now, dis.before, parts_now, parts_before,
                                     dis.split, step_errors, file_io.files.error_file, recent_errors, ttol,
                                     dt_limits["dt_min"], time_dependent_arguments=[voltage], approximation="LMEA")
        file_io.log("time", file_io.files.model_log, t)
        dis.energy.vector()[:] = np.exp(dis.w_now.vector()[:] - dis.dens_now[ns - 1].vector()[:])
        window, stride = file_io.file_output(t, t_before, window, stride, OUTPUT_WINDOWS, OUTPUT_STRIDES,
                                             ["pvd"] + ["xdmf"] * (ns - 1), out_files, out_names, out_now, out_before,
                                             unit="us")
        dt_before.time_step = dt.time_step
        dt.time_step = fedm.adaptive_timestep(dt.time_step, recent_errors, ttol, **dt_limits)
        recent_errors[1:] = recent_errors[:2]
        steps += 1
    return dict(t=t, steps=steps, output=str(file_io.files.output_folder_path), species=deck["file_names"],
                error_file=str(file_io.files.error_file), problem=problem)


if __name__ == "__main__":
    result = main(output_dir=sys.argv[1] if len(sys.argv) > 1 else "gd_output", quiet=False)
    print({k: v for k, v in result.items() if k != "problem"})
