#!/usr/bin/env python3
"""Positive streamer in atmospheric air between two planar electrodes (the benchmark of Bagheri et
al., Plasma Sources Sci. Technol. 27 (2018) 095002, case 1) on one MI355X through the fedm_amd facade.

A driver of our own for the functions a FEDM user knows (`fedm.functions`, `fedm.file_io`): the deck
is read with the deck readers, the weak form is put together from `weak_form_balance_equation_log_
representation`, `weak_form_Poisson_equation`, `Flux` and `Boundary_flux`, the time loop is
`adaptive_solver` + `adaptive_timestep`, results go through `file_output`.  It solves the case of
the reference's examples/streamer_discharge/fedm-streamer.py (same deck, boundary table, initial
condition, tolerances and end time), so the two can be compared; that the reference's script itself
lowers to the same device model is checked in tests/test_reference_scripts.py, in the build
container, against the script where it lies.

The mesh is loaded from a DOLFIN XML file like the reference's; as that file is not distributed
(.MISSING_LARGE_BLOBS), `--mesh-spacing` generates a locally refined unstructured stand-in first
(`fedm_amd.cases.streamer.refined_mesh`).  `--cells N` runs on an N x N graded tensor-product mesh
instead (the headline bench's mesh).

    python examples/streamer_discharge.py --mesh-spacing 8e-6 --end 1.4e-8 --out streamer_output
"""
import argparse
import contextlib
import io
import sys
from dataclasses import dataclass, field
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd import forms as fem                          # noqa: E402  (what `dolfin` is to a FEDM script)
from fedm_amd import file_io, functions as fedm            # noqa: E402
from fedm_amd.physical_constants import elementary_charge, epsilon_0     # noqa: E402
from fedm_amd.termsum import parse as deck_expression      # noqa: E402

REPO = Path(__file__).resolve().parent.parent
SEED = "std::log(1e13+5e18*exp(-(pow(x[0], 2)+pow(x[1]-1e-2, 2))/pow(0.4e-3, 2)))"    # ions: background + seed
BACKGROUND = "std::log(1e13)"                                                           # electrons


@dataclass
class Case:
    """Conditions of the run (Bagheri et al. 2018, case 1; values as in fedm-streamer.py:26-41,67-76,96-108)."""
    deck: str = "benchmark_model"
    pressure_torr: float = 760.0
    gas_temperature: float = 300.0
    anode_voltage: float = 18750.0
    side: float = 0.0125                        # the domain is side x side in (r, z)
    end_time: float = 1.4e-8
    first_step: float = 5e-12
    longest_step: float = 5e-12
    shortest_step: float = 1e-15
    step_tolerance: float = 1e-3
    newton_rtol: float = 1e-4
    newton_max_it: int = 20
    kinds: tuple = ("reaction", "drift-diffusion-reaction")        # ions do not move, electrons drift and diffuse
    roles: tuple = ("Ions", "electrons")
    output_windows: tuple = (1e-11, 1e-10, 1e-9)
    # per boundary (cathode z=0, anode z=side, axis r=0, outer wall r=side): condition per species
    walls: dict = field(default_factory=lambda: {
        "cathode": ("zero flux", "Neumann"), "anode": ("zero flux", "Neumann"),
        "axis": ("zero flux", "zero flux"), "outer": ("zero flux", "zero flux")})

    @property
    def gas_density(self):
        return self.pressure_torr * 3.21877e22

    def wall_lines(self):
        s = self.side
        return {"cathode": ["line", 0.0, 0.0, 0.0, s], "anode": ["line", s, s, 0.0, s],
                "axis": ["line", 0.0, s, 0.0, 0.0], "outer": ["line", 0.0, s, s, s]}


def read_deck(case, input_dir):
    """Species, charges and transport data through the FEDM deck readers."""
    file_io.files.file_input = Path(input_dir) if input_dir else REPO / "decks" / "streamer_discharge" / "file_input"
    deck_dir = file_io.files.file_input / case.deck
    n_species, names, property_files, file_names = file_io.read_speclist(deck_dir)
    masses, charge_numbers = file_io.read_particle_properties(property_files, case.deck)
    n_species, n_equations, names, masses, charge_numbers = fedm.modify_approximation_vars(
        "LFA", n_species, names, masses, charge_numbers)
    _, diffusion, _ = file_io.read_transport_coefficients(names, "Diffusion", case.deck)
    _, mobility, _ = file_io.read_transport_coefficients(names, "mobility", case.deck)
    return dict(n_species=n_species, n_equations=n_equations, names=names, file_names=file_names, masses=masses,
                charge_numbers=charge_numbers, diffusion=diffusion, mobility=mobility)


def load_mesh(case, cells, mesh_spacing, mesh_file, workdir):
    if cells:
        from fedm_amd.mesh import geometric_lines
        return fem.RectangleMesh((0.0, 0.0), (case.side, case.side), cells, cells,
                                 x_lines=geometric_lines(case.side, cells, 4.0))
    if mesh_file is None:
        from fedm_amd.cases import streamer
        from fedm_amd import mesh_io
        mesh_file = Path(workdir) / "mesh.xml"
        mesh_file.parent.mkdir(parents=True, exist_ok=True)
        mesh_io.write_dolfin_xml(streamer.refined_mesh(mesh_spacing), mesh_file)
    return fem.Mesh(str(mesh_file))


class Fields:
    """The mixed space of the coupled system, the scalar space of its parts, and the Functions."""

    def __init__(self, mesh, n_equations):
        lagrange = fem.FiniteElement("Lagrange", mesh.ufl_cell(), 1)
        self.mixed = fem.FunctionSpace(mesh, fem.MixedElement(fedm.Mixed_element_list(n_equations, lagrange)))
        self.scalar = fem.FunctionSpace(mesh, lagrange)
        parts = fedm.Function_space_list(n_equations, self.scalar)
        self.split = fem.FunctionAssigner(parts, self.mixed)          # mixed -> parts
        self.join = fem.FunctionAssigner(self.mixed, parts)           # parts -> mixed
        self.trial, self.tests = fem.TrialFunction(self.mixed), fem.TestFunctions(self.mixed)
        self.now, self.before, self.before2 = (fem.Function(self.mixed) for _ in range(3))
        self.parts_now = fedm.Function_definition(self.scalar, "Function", n_equations)
        self.parts_before = fedm.Function_definition(self.scalar, "Function", n_equations)
        self.scratch = fem.Function(self.scalar)


def electrode_conditions(case, space):
    """Dirichlet values of the potential on the two electrodes for `space` (scalar or a sub-space)."""
    def on_cathode(x, on_boundary):
        return on_boundary and fem.near(x[1], 0)

    def on_anode(x, on_boundary):
        return on_boundary and fem.near(x[1], case.side)
    return [fem.DirichletBC(space, fem.Constant(0.0), on_cathode),
            fem.DirichletBC(space, fem.Constant(case.anode_voltage), on_anode)]


def initial_state(case, deck, fl, dx, radius, writers, t):
    """Seed + background densities and the potential they produce (one linear Poisson solve)."""
    for k, text in enumerate((SEED, BACKGROUND)):
        for target in (fl.parts_before, fl.parts_now):
            target[k] = fem.interpolate(fem.Expression(text, degree=1), fl.scalar)
    for k, writer in enumerate(writers["density"]):
        fl.scratch.assign(fl.parts_before[k])
        fl.scratch.rename(deck["file_names"][k + 1], str(k + 1))
        writer << (fl.scratch, t)
    space_charge = (fem.exp(fl.parts_before[0]) - fem.exp(fl.parts_before[1])) * elementary_charge / epsilon_0
    poisson = fedm.weak_form_Poisson_equation(dx, fem.TrialFunction(fl.scalar), fem.TestFunction(fl.scalar),
                                              space_charge, radius)
    conditions = electrode_conditions(case, fl.scalar)
    matrix, load = fem.assemble(fem.lhs(poisson)), fem.assemble(fem.rhs(poisson))
    for condition in conditions:
        condition.apply(matrix)
        condition.apply(load)
    potential = fem.Function(fl.scalar)
    fem.solve(matrix, potential.vector(), load)
    fl.scratch.assign(potential)
    writers["potential"] << (fl.scratch, t)
    for target in (fl.parts_before, fl.parts_now):
        target[-1].assign(potential)


def coupled_form(case, deck, fl, mesh, dx, radius, dt, dt_before):
    """Ion and electron balance in logarithmic variables + Poisson, local field approximation."""
    wall_tags = fedm.Marking_boundaries(mesh, list(case.wall_lines().values()))
    ds = fem.Measure("ds", domain=mesh, subdomain_data=wall_tags)
    outward = fem.FacetNormal(mesh)
    u, v = fl.trial, fl.tests
    n_sp, i_phi = deck["n_species"], deck["n_equations"] - 1
    field_vector = -fem.grad(u[i_phi])
    field_strength = fem.sqrt(fem.inner(-fem.grad(u[i_phi]), -fem.grad(u[i_phi])))
    # the deck gives the electron coefficients as expressions of the field strength E_m (parsed, not
    # eval'd); the ions of this model do not move
    mobility = [deck["mobility"][0], deck_expression(deck["mobility"][1])]
    diffusion = [deck["diffusion"][0], deck_expression(deck["diffusion"][1])]
    ionisation = (1.1944e6 + 4.3666e26 * field_strength ** (-3)) * fem.exp(-2.73e7 / field_strength) - 340.75
    production = ionisation * mobility[1] * field_strength * fem.exp(u[1])            # alpha |mu E| n_e
    sources = [production, production]
    fluxes = [0.0, fedm.Flux(deck["charge_numbers"][1], u[1], diffusion[1], mobility[1], field_vector,
                             grad_diffusion=False)]
    charge_density = sum(z * fem.exp(u[k]) * elementary_charge / epsilon_0
                         for k, z in enumerate(deck["charge_numbers"][:n_sp]))
    form = 0.0
    for k in range(n_sp):
        form += fedm.weak_form_balance_equation_log_representation(
            case.kinds[k], dt, dt_before, dx, u[k], fl.before[k], fl.before2[k], v[k], sources[k], fluxes[k],
            radius, diffusion[k])
    form += fedm.weak_form_Poisson_equation(dx, u[i_phi], v[i_phi], charge_density, radius)
    for tag, wall in enumerate(case.walls, start=1):
        for k in range(n_sp):
            form += fedm.Boundary_flux(case.walls[wall][k], case.kinds[k], case.roles[k], deck["charge_numbers"][k],
                                       mobility[k], field_vector, outward, u[k], 0.0, v[k], ds(tag), radius)
    return form, wall_tags


def quiet_stdout(quiet):
    return contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()


def main(cells=None, mesh_spacing=2.5e-5, mesh_file=None, end_time=None, input_dir=None, output_dir=None,
         quiet=False, stop_before_device=None):
    """Runs the case; returns (state as (n_vertices, 3) array of ln n_i, ln n_e, Phi; error-log path)."""
    case = Case() if end_time is None else Case(end_time=end_time)
    fem.parameters["form_compiler"]["quadrature_degree"] = 2
    deck = read_deck(case, input_dir)
    if output_dir is not None:
        file_io.files.output_folder_path = Path(output_dir)
    out = file_io.files.output_folder_path
    writers = {"density": file_io.output_files("pvd", "number density", list(case.roles)),
               "potential": file_io.output_files("pvd", "potential", ["Phi"])[0]}
    dt = fem.Expression("time_step", time_step=case.first_step, degree=0)
    dt_before = fem.Expression("time_step", time_step=1e30, degree=0)     # "no previous step": BDF2 starts as BDF1
    charges = [z * elementary_charge for z in deck["charge_numbers"]]
    file_io.log("conditions", file_io.files.model_log, dt.time_step, case.anode_voltage, case.pressure_torr, case.side,
                case.gas_density, case.gas_temperature)
    file_io.log("properties", file_io.files.model_log, "Air", case.deck, deck["file_names"], deck["masses"], charges)

    mesh = load_mesh(case, cells, mesh_spacing, mesh_file, out / "mesh")
    with quiet_stdout(quiet):
        file_io.mesh_statistics(mesh)
    radius = fem.Expression("x[0]", degree=1)
    dx = fem.Measure("dx", domain=mesh)
    fl = Fields(mesh, deck["n_equations"])
    t = 0.0
    file_io.log("initial time", file_io.files.model_log, t)
    initial_state(case, deck, fl, dx, radius, writers, t)
    form, wall_tags = coupled_form(case, deck, fl, mesh, dx, radius, dt, dt_before)
    fem.File(str(out / "mesh" / "boundary_mesh_function.pvd")) << wall_tags
    fl.join.assign(fl.before, list(fl.parts_before))
    fl.join.assign(fl.now, list(fl.parts_now))

    residual = fem.action(form, fl.now)
    jacobian = fem.derivative(residual, fl.now, fl.trial)
    build_problem = stop_before_device or fedm.Problem
    problem = build_problem(jacobian, residual, electrode_conditions(case, fl.mixed.sub(deck["n_equations"] - 1)))
    problem.device.setup_multigrid(nu=1)                 # preconditioner of the device's GMRES (no script counterpart)
    newton = fedm.PETScSNESSolver()
    newton.parameters["relative_tolerance"] = case.newton_rtol
    newton.parameters["maximum_iterations"] = case.newton_max_it
    newton.parameters["linear_solver"] = "gmres"

    # what file_output interpolates between: potential first, then the species
    order = [deck["n_equations"] - 1] + list(range(deck["n_species"]))
    out_files = [writers["potential"]] + list(writers["density"])
    out_names = ["Phi"] + list(case.roles)
    out_now, out_before = [fl.parts_now[k] for k in order], [fl.parts_before[k] for k in order]
    window, stride = case.output_windows[0], case.output_windows[0]
    step_errors, recent_errors = [0.0] * deck["n_species"], [1] * 3
    while abs(t - case.end_time) / case.end_time > 1e-6:
        t_before = t
        fl.before2.assign(fl.before)
        fl.before.assign(fl.now)
        fl.split.assign(list(fl.parts_before), fl.before)
        with quiet_stdout(quiet):
            t = fedm.adaptive_solver(newton, problem, t, dt, dt_before, fl.now, fl.before, list(fl.parts_now),
                                     list(fl.parts_before), fl.split, step_errors, file_io.files.error_file,
                                     recent_errors, case.step_tolerance, case.shortest_step,
                                     time_dependent_arguments=[], approximation="LFA")
        file_io.log("time", file_io.files.model_log, t)
        dt_before.time_step = dt.time_step
        dt.time_step = fedm.adaptive_timestep(dt.time_step, recent_errors, case.step_tolerance, case.shortest_step,
                                              case.longest_step)
        recent_errors[1:] = recent_errors[:2]
        window, stride = file_io.file_output(t, t_before, window, stride, list(case.output_windows),
                                             list(case.output_windows), ["pvd"] * len(order), out_files, out_names,
                                             out_now, out_before)
    return problem.device.get_state(), file_io.files.error_file


if __name__ == "__main__":
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--cells", type=int, help="N x N graded tensor-product mesh instead of the unstructured one")
    ap.add_argument("--mesh-spacing", type=float, default=2.5e-5, help="finest spacing of the generated mesh [m]")
    ap.add_argument("--mesh", help="DOLFIN XML mesh to load instead of generating one")
    ap.add_argument("--end", type=float, default=1e-10, help="end time [s] (the reference script: 1.4e-8)")
    ap.add_argument("--out", default="streamer_output")
    a = ap.parse_args()
    state, log_path = main(cells=a.cells, mesh_spacing=a.mesh_spacing, mesh_file=a.mesh, end_time=a.end,
                           output_dir=a.out)
    print(open(log_path).read())
