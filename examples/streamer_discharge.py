#!/usr/bin/env python3
"""Positive streamer in air (Bagheri et al. 2018) -- the reference's
examples/streamer_discharge/fedm-streamer.py with the same sequence of calls (its lines are cited on
the right), on the MI355X device path.

Differences forced by the platform: no DOLFIN (`from fedm_amd.forms import *` stands for
`from dolfin import *`); the mesh is generated (the reference's mesh.xml is not distributed: a graded
tensor-product mesh of `n` x `n` cells); the deck's coefficient strings are parsed
(`parse_coefficient`) where the reference eval()s them.  The initial conditions are the reference's
C++ Expression strings, the potential of the initial time step is the script's own
lhs / rhs / assemble / bc.apply / solve (on the host), the time loop is `adaptive_solver` with the
Newton solves on the device.
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd.forms import *                      # noqa: F401,F403,E402  (stands for `from dolfin import *`)
from fedm_amd.physical_constants import *         # noqa: F401,F403,E402
from fedm_amd.file_io import *                    # noqa: F401,F403,E402
from fedm_amd.functions import *                  # noqa: F401,F403,E402
from fedm_amd.mesh import geometric_lines                  # noqa: E402  (grading of the generated mesh)
from fedm_amd.termsum import parse as parse_coefficient    # noqa: E402


def main(n=64, T_final=1e-10, input_dir=None, output_dir=None, quiet=False):
    parameters["form_compiler"]["optimize"] = True                                         # :19-23
    parameters["form_compiler"]["cpp_optimize"] = True
    parameters["std_out_all_processes"] = False
    parameters['krylov_solver']['nonzero_initial_guess'] = True
    parameters["form_compiler"]["quadrature_degree"] = 2

    linear_solver = "gmres"          # (the reference's default is "mumps"; the device solves with GMRES)   :26-28
    maximum_iterations = 20
    relative_tolerance = 1e-4

    model = 'benchmark_model'                                                              # :33-41
    coordinates = 'cylindrical'
    gas = 'Air'
    Tgas = 300.0
    p0 = 760.0
    N0 = p0 * 3.21877e22
    U_w = 18750.0
    approximation = 'LFA'
    files.file_input = Path(input_dir) if input_dir else \
        Path(__file__).resolve().parent.parent / "decks" / "streamer_discharge" / "file_input"
    if output_dir is not None:
        files.output_folder_path = Path(output_dir)
    path = files.file_input / model

    number_of_species, particle_species, particle_prop, particle_species_file_names = read_speclist(path)   # :47-60
    M, sign = read_particle_properties(particle_prop, model)
    equation_type = ['reaction', 'drift-diffusion-reaction']
    particle_species_type = ['Ions', 'electrons']
    number_of_species, number_of_equations, particle_species, M, sign = modify_approximation_vars(
        approximation, number_of_species, particle_species, M, sign)
    charge = [i * elementary_charge for i in sign]
    vtkfile_u = output_files('pvd', 'number density', particle_species_type)
    vtkfile_Phi = output_files('pvd', 'potential', ['Phi'])
    output_file_list = [vtkfile_Phi[0], vtkfile_u[0], vtkfile_u[1]]
    file_type = ['pvd', 'pvd', 'pvd']

    t_old = None                                                                           # :65-89
    t0 = 0.0
    t = t0
    dt_min = 1e-15
    dt_max = 5e-12
    dt_init = 5e-12
    dt_old_init = 1e30
    dt = Expression("time_step", time_step=dt_init, degree=0)
    dt_old = Expression("time_step", time_step=dt_old_init, degree=0)
    ttol = 1e-3
    t_output_list = [1e-11, 1e-10, 1e-9]
    t_output_step_list = [1e-11, 1e-10, 1e-9]
    t_output_step = t_output_list[0]
    t_output = t_output_step_list[0]
    error = [0.0] * number_of_species
    max_error = [1] * 3

    if coordinates == 'cylindrical':                                                       # :94-112
        r = Expression('x[0]', degree=1)
        z = Expression('x[1]', degree=1)                                                   # noqa: F841
    box_width = 0.0125
    box_height = 0.0125
    boundaries = [['line', 0.0, 0.0, 0.0, box_width],
                  ['line', box_height, box_height, 0.0, box_width],
                  ['line', 0.0, box_height, 0.0, 0.0],
                  ['line', 0.0, box_height, box_width, box_width]]
    number_of_boundaries = len(boundaries)
    bc_type_grounded = ['zero flux', 'Neumann']
    bc_type_powered = ['zero flux', 'Neumann']
    bc_type_axis = ['zero flux', 'zero flux']
    bc_type_wall = ['zero flux', 'zero flux']
    bc_type = [bc_type_grounded, bc_type_powered, bc_type_axis, bc_type_wall]
    gamma = [0.0, 0.0]
    log('conditions', files.model_log, dt.time_step, U_w, p0, box_height, N0, Tgas)
    log('properties', files.model_log, gas, model, particle_species_file_names, M, charge)

    mesh = RectangleMesh((0.0, 0.0), (box_width, box_height), n, n,                        # :117-127 (Mesh('mesh.xml'))
                         x_lines=geometric_lines(box_width, n, 4.0))
    with open_quietly(quiet):
        mesh_statistics(mesh)
    boundary_mesh_function = Marking_boundaries(mesh, boundaries)
    normal = FacetNormal(mesh)
    File(str(files.output_folder_path / 'mesh' / 'boundary_mesh_function.pvd')) << boundary_mesh_function
    dx = Measure('dx', domain=mesh)
    ds = Measure('ds', domain=mesh, subdomain_data=boundary_mesh_function)
    log('initial time', files.model_log, t)

    P1 = FiniteElement("Lagrange", mesh.ufl_cell(), 1)                                     # :132-163
    Element_list = Mixed_element_list(number_of_equations, P1)
    ME = FunctionSpace(mesh, MixedElement(Element_list))
    V = FunctionSpace(mesh, P1)
    W = VectorFunctionSpace(mesh, 'P', 1)                                                  # noqa: F841
    assigner = FunctionAssigner(Function_space_list(number_of_equations, V), ME)
    rev_assigner = FunctionAssigner(ME, Function_space_list(number_of_equations, V))
    temp_output_variable = Function(V)
    u = TrialFunction(ME)
    v = TestFunctions(ME)
    u_new = Function(ME)
    u_old = Function(ME)
    u_old1 = Function(ME)
    PhiV = TrialFunction(V)
    vp = TestFunction(V)
    Phi = Function(V)
    Phi_old = Function(V)                                                                  # noqa: F841
    u_newV = Function_definition(V, 'Function', number_of_equations)
    u_oldV = Function_definition(V, 'Function', number_of_equations)
    u_old1V = Function_definition(V, 'Function', number_of_equations)                      # noqa: F841
    mu = Function_definition(V, 'Function', number_of_equations)
    D = Function_definition(V, 'Function', number_of_equations)
    Gamma = Function_definition(V, 'Function', number_of_equations)
    f = Function_definition(V, 'Function', number_of_equations)

    u_oldV[0] = interpolate(Expression('std::log(1e13+5e18*exp(-(pow(x[0], 2)+pow(x[1]-1e-2, 2))/pow(0.4e-3, 2)))', degree=1), V)   # :168-171
    u_oldV[1] = interpolate(Expression('std::log(1e13)', degree=1), V)
    u_newV[0] = interpolate(Expression('std::log(1e13+5e18*exp(-(pow(x[0], 2)+pow(x[1]-1e-2, 2))/pow(0.4e-3, 2)))', degree=1), V)
    u_newV[1] = interpolate(Expression('std::log(1e13)', degree=1), V)

    i = 0                                                                                  # :174-179
    while i < number_of_species:
        temp_output_variable.assign(u_oldV[i])
        temp_output_variable.rename(particle_species_file_names[i + 1], str(i + 1))
        vtkfile_u[i] << (temp_output_variable, t)
        i += 1

    Phi_cathode = Constant(0.0)                                                            # :185-199
    Phi_anode = Constant(U_w)

    def Cathode(x, on_boundary):
        return near(x[1], 0) and on_boundary

    def Anode(x, on_boundary):
        return near(x[1], box_height) and on_boundary

    potential_Cathode_bc = DirichletBC(V, Phi_cathode, Cathode)
    potential_Anode_bc = DirichletBC(V, Phi_anode, Anode)
    bcs_potential = [potential_Cathode_bc, potential_Anode_bc]

    potential_f = (exp(u_oldV[0]) - exp(u_oldV[1])) * elementary_charge / epsilon_0        # :201-215
    Fp = weak_form_Poisson_equation(dx, PhiV, vp, potential_f, r)
    a, L = lhs(Fp), rhs(Fp)
    potential_A = assemble(a)
    [bc_.apply(potential_A) for bc_ in bcs_potential]
    potential_b = assemble(L)
    [bc_.apply(potential_b) for bc_ in bcs_potential]
    solve(potential_A, Phi.vector(), potential_b)

    temp_output_variable.assign(Phi)                                                       # :217-225
    vtkfile_Phi[0] << (temp_output_variable, t)
    E = -grad(u[2])
    E_m = sqrt(inner(-grad(u[2]), -grad(u[2])))
    u_oldV[2].assign(Phi)
    u_newV[2].assign(Phi)

    D_x, D_y, Diffusion_dependence = read_transport_coefficients(particle_species, 'Diffusion', model)     # :227-228
    mu_x, mu_y, mu_dependence = read_transport_coefficients(particle_species, 'mobility', model)

    Cathode_bc = DirichletBC(ME.sub(2), Phi_cathode, Cathode)                              # :233-235
    Anode_bc = DirichletBC(ME.sub(2), Phi_anode, Anode)
    bc = [Cathode_bc, Anode_bc]

    D[0] = D_y[0]                                                                          # :236-249
    mu[1] = parse_coefficient(mu_y[1])          # the reference eval()s these deck strings
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

    F = 0.0                                                                                # :252-271
    i = 0
    while i < number_of_species:
        F += weak_form_balance_equation_log_representation(equation_type[i], dt, dt_old, dx, u[i], u_old[i],
                                                           u_old1[i], v[i], f[i], Gamma[i], r, D[i])
        i += 1
    F += weak_form_Poisson_equation(dx, u[number_of_equations - 1], v[number_of_equations - 1],
                                    f[number_of_equations - 1], r)
    i = 0
    while i < number_of_boundaries:
        j = 0
        while j < number_of_species:
            F += Boundary_flux(bc_type[i][j], equation_type[j], particle_species_type[j], sign[j], mu[j], E,
                               normal, u[j], gamma[j], v[j], ds(i + 1), r)
            j += 1
        i += 1

    variable_list_new = [u_newV[0], u_newV[1], u_newV[2]]                                  # :276-283
    variable_list_old = [u_oldV[0], u_oldV[1], u_oldV[2]]
    output_old_variable_list = [u_oldV[2], u_oldV[0], u_oldV[1]]
    output_new_variable_list = [u_newV[2], u_newV[0], u_newV[1]]
    output_files_variabe_names = ['Phi', particle_species_type[0], particle_species_type[1]]
    rev_assigner.assign(u_old, variable_list_old)
    rev_assigner.assign(u_new, variable_list_new)

    F = action(F, u_new)                                                                   # :288-299
    J = derivative(F, u_new, u)
    problem = Problem(J, F, bc)
    problem.device.setup_multigrid(nu=1)        # the device's preconditioner (the reference: "hypre_amg" with gmres)
    nonlinear_solver = PETScSNESSolver()
    nonlinear_solver.parameters['relative_tolerance'] = relative_tolerance
    nonlinear_solver.parameters["linear_solver"] = linear_solver
    nonlinear_solver.parameters['maximum_iterations'] = maximum_iterations

    while abs(t - T_final) / T_final > 1e-6:                                               # :304-345
        t_old = t
        u_old1.assign(u_old)
        u_old.assign(u_new)
        assigner.assign(variable_list_old, u_old)
        with open_quietly(quiet):
            t = adaptive_solver(nonlinear_solver, problem, t, dt, dt_old, u_new, u_old, variable_list_new,
                                variable_list_old, assigner, error, files.error_file, max_error, ttol, dt_min,
                                time_dependent_arguments=[], approximation=approximation)
        log('time', files.model_log, t)
        dt_old.time_step = dt.time_step
        dt.time_step = adaptive_timestep(dt.time_step, max_error, ttol, dt_min, dt_max)
        max_error[2] = max_error[1]
        max_error[1] = max_error[0]
        t_output, t_output_step = file_output(t, t_old, t_output, t_output_step, t_output_list, t_output_step_list,
                                              file_type, output_file_list, output_files_variabe_names,
                                              output_new_variable_list, output_old_variable_list)
    return problem.device.get_state(), files.error_file


def open_quietly(quiet):
    """stdout of the library's progress lines, or nothing (tests)."""
    import contextlib
    import io
    return contextlib.redirect_stdout(io.StringIO()) if quiet else contextlib.nullcontext()


if __name__ == "__main__":
    state, log_path = main(n=int(sys.argv[1]) if len(sys.argv) > 1 else 64)
    print(open(log_path).read())
