#!/usr/bin/env python3
"""Positive streamer in air (Bagheri et al. 2018) -- the reference's
examples/streamer_discharge/fedm-streamer.py with the same sequence of calls, on the
MI355X device path.  Differences forced by the platform: no DOLFIN (so `from fedm_amd.forms
import *` stands for `from dolfin import *`), the mesh is generated (the reference's mesh.xml is not
distributed), results are returned instead of written as PVD files.  The initial conditions are the
reference's C++ Expression strings, the potential of the initial time step is assembled and solved as
the script does it (lhs / rhs / assemble / bc.apply / solve, on the host).
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd.forms import *                      # noqa: F401,F403  (stands for `from dolfin import *`)
from fedm_amd.physical_constants import *         # noqa: F401,F403
from fedm_amd.file_io import *                    # noqa: F401,F403
from fedm_amd.functions import *                  # noqa: F401,F403
from fedm_amd.mesh import RectangleMesh, geometric_lines
from fedm_amd.termsum import parse as parse_coefficient


def main(n=64, T_final=1e-10, input_dir=None, output_dir=None, quiet=False):
    parameters["form_compiler"]["quadrature_degree"] = 2

    linear_solver = "gmres"
    maximum_iterations = 20
    relative_tolerance = 1e-4

    model = 'benchmark_model'
    U_w = 18750.0
    approximation = 'LFA'
    files.file_input = Path(input_dir) if input_dir else \
        Path(__file__).resolve().parent.parent / "decks" / "streamer_discharge" / "file_input"
    if output_dir is not None:
        files.output_folder_path = Path(output_dir)
    path = files.file_input / model

    number_of_species, particle_species, particle_prop, particle_species_file_names = read_speclist(path)
    M, sign = read_particle_properties(particle_prop, model)
    equation_type = ['reaction', 'drift-diffusion-reaction']
    particle_species_type = ['Ions', 'electrons']
    number_of_species, number_of_equations, particle_species, M, sign = modify_approximation_vars(
        approximation, number_of_species, particle_species, M, sign)

    t = 0.0
    dt_min, dt_max, dt_init, dt_old_init = 1e-15, 5e-12, 5e-12, 1e30
    dt = Expression("time_step", time_step=dt_init, degree=0)
    dt_old = Expression("time_step", time_step=dt_old_init, degree=0)
    ttol = 1e-3
    error = [0.0] * number_of_species
    max_error = [1] * 3

    r = Expression('x[0]', degree=1)
    box_width = box_height = 0.0125
    boundaries = [['line', 0.0, 0.0, 0.0, box_width],
                  ['line', box_height, box_height, 0.0, box_width],
                  ['line', 0.0, box_height, 0.0, 0.0],
                  ['line', 0.0, box_height, box_width, box_width]]
    number_of_boundaries = len(boundaries)
    bc_type = [['zero flux', 'Neumann'], ['zero flux', 'Neumann'],
               ['zero flux', 'zero flux'], ['zero flux', 'zero flux']]
    gamma = [0.0, 0.0]

    mesh = RectangleMesh((0.0, 0.0), (box_width, box_height), n, n,
                         x_lines=geometric_lines(box_width, n, 4.0))
    boundary_mesh_function = Marking_boundaries(mesh, boundaries)
    normal = FacetNormal(mesh)
    dx_ = Measure('dx', domain=mesh)
    ds_ = Measure('ds', domain=mesh, subdomain_data=boundary_mesh_function)

    P1 = FiniteElement("Lagrange", None, 1)
    Element_list = Mixed_element_list(number_of_equations, P1)
    ME = FunctionSpace(mesh, MixedElement(Element_list))
    V = FunctionSpace(mesh, P1)
    assigner = FunctionAssigner(Function_space_list(number_of_equations, V), ME)

    u = TrialFunction(ME)
    v = TestFunctions(ME)
    mu = Function_definition(V, 'Function', number_of_equations)
    D = Function_definition(V, 'Function', number_of_equations)
    Gamma = Function_definition(V, 'Function', number_of_equations)
    f = Function_definition(V, 'Function', number_of_equations)

    # variables of the initial Poisson problem and of the post-processing (fedm-streamer.py:150-163)
    PhiV = TrialFunction(V)
    vp = TestFunction(V)
    Phi = Function(V)
    u_newV = Function_definition(V, 'Function', number_of_equations)
    u_oldV = Function_definition(V, 'Function', number_of_equations)

    # initial conditions (:168-171): the reference's C++ strings
    u_oldV[0] = interpolate(Expression('std::log(1e13+5e18*exp(-(pow(x[0], 2)+pow(x[1]-1e-2, 2))/pow(0.4e-3, 2)))', degree=1), V)
    u_oldV[1] = interpolate(Expression('std::log(1e13)', degree=1), V)
    u_newV[0] = interpolate(Expression('std::log(1e13+5e18*exp(-(pow(x[0], 2)+pow(x[1]-1e-2, 2))/pow(0.4e-3, 2)))', degree=1), V)
    u_newV[1] = interpolate(Expression('std::log(1e13)', degree=1), V)

    Phi_cathode, Phi_anode = Constant(0.0), Constant(U_w)

    def Cathode(x, on_boundary):
        return near(x[1], 0) and on_boundary

    def Anode(x, on_boundary):
        return near(x[1], box_height) and on_boundary

    # the potential of the initial time step (:196-215): assembled and solved on the host
    potential_Cathode_bc = DirichletBC(V, Phi_cathode, Cathode)
    potential_Anode_bc = DirichletBC(V, Phi_anode, Anode)
    bcs_potential = [potential_Cathode_bc, potential_Anode_bc]
    potential_f = (exp(u_oldV[0]) - exp(u_oldV[1])) * elementary_charge / epsilon_0
    Fp = weak_form_Poisson_equation(dx_, PhiV, vp, potential_f, r)
    a, L = lhs(Fp), rhs(Fp)
    potential_A = assemble(a)
    [bc_.apply(potential_A) for bc_ in bcs_potential]
    potential_b = assemble(L)
    [bc_.apply(potential_b) for bc_ in bcs_potential]
    solve(potential_A, Phi.vector(), potential_b)
    u_oldV[2].assign(Phi)                                                                  # :224-225
    u_newV[2].assign(Phi)

    E = -grad(u[2])
    E_m = sqrt(inner(-grad(u[2]), -grad(u[2])))

    D_x, D_y, Diffusion_dependence = read_transport_coefficients(particle_species, 'Diffusion', model)
    mu_x, mu_y, mu_dependence = read_transport_coefficients(particle_species, 'mobility', model)

    bc = [DirichletBC(ME.sub(2), Phi_cathode, Cathode), DirichletBC(ME.sub(2), Phi_anode, Anode)]

    mu[0] = mu_y[0]
    D[0] = D_y[0]
    mu[1] = parse_coefficient(mu_y[1])      # the reference eval()s these deck strings (:237-238)
    D[1] = parse_coefficient(D_y[1])
    alpha = (1.1944e6 + 4.3666e26 * E_m**(-3)) * exp(-2.73e7 / E_m) - 340.75

    Gamma[0] = 0.0
    Gamma[1] = Flux(sign[1], u[1], D[1], mu[1], E, grad_diffusion=False)

    f[0] = alpha * mu[1] * E_m * exp(u[1])
    f[1] = alpha * mu[1] * E_m * exp(u[1])
    i = 0
    while i < number_of_species:
        f[2] += sign[i] * exp(u[i]) * elementary_charge / epsilon_0
        i += 1

    F = 0.0
    i = 0
    while i < number_of_species:
        F += weak_form_balance_equation_log_representation(
            equation_type[i], dt, dt_old, dx_, u[i], None, None, v[i], f[i], Gamma[i], r, D[i])
        i += 1
    F += weak_form_Poisson_equation(dx_, u[number_of_equations - 1], v[number_of_equations - 1],
                                    f[number_of_equations - 1], r)
    i = 0
    while i < number_of_boundaries:
        j = 0
        while j < number_of_species:
            F += Boundary_flux(bc_type[i][j], equation_type[j], particle_species_type[j], sign[j],
                               mu[j], E, normal, u[j], gamma[j], v[j], ds_(i + 1), r)
            j += 1
        i += 1

    F = action(F, None)
    J = derivative(F, None, u)
    problem = Problem(J, F, bc)

    # the device states start from the script's per-field Functions (what the reference's assigner does
    # with u_oldV / u_newV, :287-290)
    dev = problem.device
    U0 = np.stack([np.asarray(fn.vector(), dtype=float) for fn in u_oldV], axis=1)
    dev.set_state(U0, U0, U0)
    dev.setup_multigrid(nu=1)
    u_new, u_old, u_old1 = DeviceState(dev, "new"), DeviceState(dev, "old"), DeviceState(dev, "old1")

    nonlinear_solver = PETScSNESSolver()
    nonlinear_solver.parameters['relative_tolerance'] = relative_tolerance
    nonlinear_solver.parameters["linear_solver"] = linear_solver
    nonlinear_solver.parameters['maximum_iterations'] = maximum_iterations

    import contextlib, io
    while abs(t - T_final) / T_final > 1e-6:
        u_old1.assign(u_old)
        u_old.assign(u_new)
        with contextlib.redirect_stdout(io.StringIO() if quiet else sys.stdout):
            t = adaptive_solver(nonlinear_solver, problem, t, dt, dt_old, u_new, u_old, None, None,
                                assigner, error, files.error_file, max_error, ttol, dt_min,
                                time_dependent_arguments=[], approximation=approximation)
        dt_old.time_step = dt.time_step
        dt.time_step = adaptive_timestep(dt.time_step, max_error, ttol, dt_min, dt_max)
        max_error[2] = max_error[1]
        max_error[1] = max_error[0]
    return dev.get_state(), files.error_file


if __name__ == "__main__":
    state, log = main(n=int(sys.argv[1]) if len(sys.argv) > 1 else 64)
    print(open(log).read())
