#!/usr/bin/env python3
"""Low-pressure argon glow discharge (LMEA: electron energy + 4 particles + Poisson) -- the
reference's examples/glow_discharge/fedm-gd.py with the same sequence of calls (its lines are
cited on the right), on the MI355X device path.

Differences forced by the platform, as in examples/streamer_discharge.py: no DOLFIN
(`from fedm_amd.forms import *` stands for `from dolfin import *`); the C++ Expression strings are the
reference's (arithmetic subset, evaluated without a JIT).  The initial Poisson solve
(fedm-gd.py:283-300) is the script's own lhs / rhs / assemble / bc.apply / solve, on the host.

What the script does per time step is what the reference does: it refreshes the nodal transport
and rate coefficients ON THE HOST with `Transport_coefficient_interpolation` & co. (numpy), and
`adaptive_solver` runs the Newton solves on the device.  The device-resident variant of the same
pipeline (no state leaves the GPU between output times) is `fedm_amd.cases.glow_discharge.Case`.
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd.forms import *                      # noqa: F401,F403  (stands for `from dolfin import *`)
from fedm_amd.physical_constants import *         # noqa: F401,F403
from fedm_amd.file_io import *                    # noqa: F401,F403
from fedm_amd.functions import *                  # noqa: F401,F403


def main(nx=100, ny=100, T_final=1e-11, input_dir=None, output_dir="gd_output", quiet=True, ttol=2e-3):
    parameters["form_compiler"]["quadrature_degree"] = 4                                   # :28

    linear_solver = "mumps"                                                                # :32-34
    maximum_iterations = 20
    relative_tolerance = 1e-4

    model = '4_particles'                                                                  # :39-50
    semi_implicit = True
    gas = 'Ar'
    Tgas = 300.0
    p0 = 1.0
    N0 = p0 * 3.21877e22
    U_w = -250.0
    approximation = 'LMEA'
    files.file_input = Path(input_dir) if input_dir else \
        Path(__file__).resolve().parent.parent / "decks" / "glow_discharge" / "file_input"
    files.output_folder_path = Path(output_dir)
    path = files.file_input / model

    number_of_species, particle_species, particle_prop, particle_species_file_names = read_speclist(path)   # :56
    M, sign = read_particle_properties(particle_prop, model)
    charge = [i * elementary_charge for i in sign]
    equation_type = ['reaction', 'diffusion-reaction', 'drift-diffusion-reaction', 'drift-diffusion-reaction']
    particle_type = ['Heavy', 'Heavy', 'Heavy', 'electrons']
    particle_species_type = ['Neutral', 'Neutral', 'Ion', 'electrons']
    n_ic = [N0, 1e12, 1e12, 1e12]
    grad_diff = [pst == 'electrons' for pst in particle_species_type]

    power_matrix, loss_matrix, gain_matrix = reaction_matrices(path, particle_species)    # :69-90
    k_file_names = rate_coefficient_file_names(path)
    energy_loss = read_energy_loss(path)
    number_of_reactions = len(k_file_names)
    mu_x, mu_y, mobility_dependence = read_transport_coefficients(particle_species_file_names, 'mobility', model)
    D_x, D_y, Diffusion_dependence = read_transport_coefficients(particle_species_file_names, 'Diffusion', model)
    k_dependence = read_dependences(k_file_names)
    k_x, k_y = read_rate_coefficients(k_file_names, k_dependence)
    De_diff = np.gradient(D_y[number_of_species - 1], D_x[number_of_species - 1]) / N0
    mue_diff = np.gradient(mu_y[number_of_species - 1], mu_x[number_of_species - 1]) / N0
    k_diff = []
    i = 0
    while i < len(k_y):
        if k_dependence[i] == "Umean":
            k_diff.append(np.gradient(k_y[i], k_x[i]))
        else:
            k_diff.append(0.0)
        i += 1

    number_of_species, number_of_equations, particle_species, M, sign = modify_approximation_vars(      # :93
        approximation, number_of_species, particle_species, M, sign)

    xdmf_file_u = output_files('xdmf', 'number density', particle_species_file_names)      # :95-98
    vtkfile_Phi = output_files('pvd', 'potential', ['Phi'])
    output_file_list = [vtkfile_Phi[0], xdmf_file_u[1], xdmf_file_u[2], xdmf_file_u[3]]
    file_type = ['pvd', 'xdmf', 'xdmf', 'xdmf']

    t_old = None                                                                           # :103-128
    t = 0.0
    dt_min, dt_max, dt_init, dt_old_init = 1e-15, 1e-8, 1e-13, 1e30
    dt = Expression("time_step", time_step=dt_init, degree=0)
    dt_old = Expression("time_step", time_step=dt_old_init, degree=0)
    dt_old1 = Expression("time_step", time_step=dt_old_init, degree=0)
    t_output_list = [1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5]
    t_output_step_list = [1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-6]
    t_output_step = t_output_list[0]
    t_output = t_output_step_list[0]
    error = [0.0] * (number_of_species + 1)
    max_error = [1] * 3

    r = Expression('x[0]', degree=1)                                                       # :133-155
    gap_length = 0.01
    wall = 0.01
    boundaries = [['line', 0.0, 0.0, 0.0, wall], ['line', gap_length, gap_length, 0.0, wall],
                  ['line', 0.0, gap_length, 0.0, 0.0], ['line', 0.0, gap_length, wall, wall]]
    number_of_boundaries = len(boundaries)
    ref_metallic = [0.3, 0.3, 5e-4, 0.3]
    ref_zero = [1.0, 1.0, 1.0, 1.0]
    ref_coeff = [ref_metallic, ref_metallic, ref_zero, ref_zero]
    gamma_metallic = 0.06
    gamma = [gamma_metallic, gamma_metallic, 0, 0]
    we_metalic = 5.0

    log('conditions', files.model_log, dt.time_step, U_w, p0, gap_length, N0, Tgas)        # :157-158
    log('properties', files.model_log, gas, model, particle_species_file_names, M, charge)

    mesh_plasma = RectangleMesh((0, 0), (wall, gap_length), nx, ny, "crossed")             # :163-173
    boundary_mesh_function = Marking_boundaries(mesh_plasma, boundaries)
    ds_plasma = Measure('ds', domain=mesh_plasma, subdomain_data=boundary_mesh_function)
    normal_plasma = FacetNormal(mesh_plasma)
    log('matrices', files.model_log, gain_matrix, loss_matrix, power_matrix)
    log('initial time', files.model_log, t)

    P1 = FiniteElement("Lagrange", mesh_plasma.ufl_cell(), 1)                              # :178-201
    elements_list = Mixed_element_list(number_of_equations, P1)
    Element = MixedElement(elements_list)
    ME = FunctionSpace(mesh_plasma, Element)
    V = FunctionSpace(mesh_plasma, P1)
    function_space_list = Function_space_list(number_of_equations, V)
    assigner = FunctionAssigner(function_space_list, ME)
    rev_assigner = FunctionAssigner(ME, function_space_list)
    temp_output_variable = Function(V)

    u = TrialFunction(ME)
    v = TestFunctions(ME)
    u_new = Function(ME)
    u_old = Function(ME)
    u_old1 = Function(ME)
    F = 0

    Phi = Function(V)
    Phi_old = Function(V)
    Phi_old1 = Function(V)
    u_phi = TrialFunction(V)                                                               # :196-202
    v_phi = TestFunction(V)
    rho_poisson = 0
    rho_poisson_C = 0
    redE = Function(V)
    redE_old = Function(V)
    E = -grad(u[number_of_equations - 1])                                                  # :205-206
    E_magnitude = sqrt(inner(E, E))

    u_oldV = Function_definition(V, 'Function', number_of_species)                         # :209-216
    u_old1V = Function_definition(V, 'Function', number_of_species)
    u_newV = Function_definition(V, 'Function', number_of_species)
    mean_energy = Function(V)
    mean_energy_old = Function(V)
    mean_energy_old1 = Function(V)
    mean_energy_e = mean_energy_old + (exp(u[0]) - exp(u[number_of_species - 1]) * mean_energy_old) \
        / exp(u_oldV[number_of_species - 1])
    Gamma = [0]

    vth = [0] * number_of_species                                                          # :218-232
    i = 1
    while i < number_of_species - 1:
        vth[i] = np.sqrt(8.0 * kB * Tgas / (pi * M[i]))
        i += 1
    vth[number_of_species - 1] = sqrt(16.0 * elementary_charge * mean_energy / (3.0 * pi * M[number_of_species - 1]))
    D = Function_definition(V, 'Function', number_of_species)
    D_diff = Function_definition(V, 'Function', number_of_species)
    mu = Function_definition(V, 'Function', number_of_species)
    mu_diff = Function_definition(V, 'Function', number_of_species)
    rate_coefficient = Function_definition(V, 'Function', number_of_reactions)
    rate_coefficient_diff = Function_definition(V, 'Function', number_of_reactions)
    epsilon = Constant(1.0) * epsilon_0

    n_init = [0] * number_of_species                                                       # :237-262
    i = 0
    while i < number_of_species:
        n_init[i] = Expression('std::log(ic)', ic=n_ic[i], degree=1)
        i += 1
    mean_energy_init = interpolate(Expression('3.0', degree=1), V)
    mean_energy.assign(mean_energy_init)
    mean_energy_old.assign(mean_energy_init)
    mean_energy_old1.assign(Constant(0.0))
    i = 0
    while i < number_of_species:
        u_newV[i].assign(n_init[i])
        u_oldV[i].assign(n_init[i])
        u_old1V[i].assign(Constant(0.0))
        rho_poisson += elementary_charge * sign[i] * exp(u_oldV[i])
        rho_poisson_C += elementary_charge * sign[i] * exp(u[i])
        i += 1
    log_energy_density = Expression('std::log(a) + b', a=mean_energy, b=u_oldV[number_of_species - 1], degree=1)
    we_newV = interpolate(log_energy_density, V)
    we_oldV = interpolate(log_energy_density, V)
    we_old1V = interpolate(Constant(0.0), V)

    i = 0                                                                                  # :265-270
    while i < number_of_species:
        temp_output_variable.assign(u_oldV[i])
        temp_output_variable.rename(particle_species_file_names[i], str(i))
        xdmf_file_u[i].write_checkpoint(temp_output_variable, particle_species_file_names[i], t * 1e6,
                                        XDMFFile.Encoding.HDF5, False)
        i += 1

    Phi_grounded = Constant(0.0)                                                           # :276-290
    Phi_powered = Expression('U0*(1-exp(-t/1e-9))', U0=U_w, t=t, pi=pi, degree=0)

    def Powered_electrode(x, on_boundary):
        return near(x[1], 0, DOLFIN_EPS) and on_boundary

    def Grounded_electrode(x, on_boundary):
        return near(x[1], gap_length, DOLFIN_EPS) and on_boundary

    Powered_Electrode_bc = DirichletBC(V, Phi_powered, Powered_electrode)                  # :280-300
    Grounded_bc = DirichletBC(V, Phi_grounded, Grounded_electrode)
    Voltage_bcs = [Powered_Electrode_bc, Grounded_bc]
    f_potential = rho_poisson / epsilon
    F_potential = weak_form_Poisson_equation(dx, u_phi, v_phi, f_potential, r)
    a_potential, L_potential = lhs(F_potential), rhs(F_potential)
    A_potential = None
    A_potential = assemble(a_potential, tensor=A_potential)
    [bc_.apply(A_potential) for bc_ in Voltage_bcs]
    b_potential = None
    b_potential = assemble(L_potential, tensor=b_potential)
    [bc_.apply(b_potential) for bc_ in Voltage_bcs]
    solve(A_potential, Phi.vector(), b_potential, 'mumps')

    Phi_old1.assign(Phi_old)                                                               # :302-315
    Phi_old.assign(Phi)
    temp_output_variable.assign(Phi)
    temp_output_variable.rename('Phi', str(0))
    vtkfile_Phi[0] << (temp_output_variable, t)
    redE.assign(project(1e21 * sqrt(dot(-grad(Phi), -grad(Phi))) / N0, solver_type='mumps'))
    redE_old.assign(redE)
    Transport_coefficient_interpolation('initial', mobility_dependence, N0, Tgas, mu, mu_x, mu_y, mean_energy, redE, mu)
    Transport_coefficient_interpolation('initial', Diffusion_dependence, N0, Tgas, D, D_x, D_y, mean_energy, redE, mu)
    Rate_coefficient_interpolation('initial', k_dependence, rate_coefficient, k_x, k_y, mean_energy, redE, Te=0, Tgas=0)

    if semi_implicit:                                                                      # :317-333
        rate_coefficient_si = semi_implicit_coefficients(k_dependence, mean_energy_e, mean_energy_old,
                                                         rate_coefficient, rate_coefficient_diff)
        mu_si = semi_implicit_coefficients(mobility_dependence, mean_energy_e, mean_energy_old, mu, mu_diff)
        D_si = semi_implicit_coefficients(Diffusion_dependence, mean_energy_e, mean_energy_old, D, D_diff)
    else:
        rate_coefficient_si, mu_si, D_si = rate_coefficient, mu, D

    Powered_Electrode_bc_C = DirichletBC(ME.sub(number_of_equations - 1), Phi_powered, Powered_electrode)    # :338-344
    Grounded_bc_C = DirichletBC(ME.sub(number_of_equations - 1), Phi_grounded, Grounded_electrode)
    Voltage_bcs_C = [Powered_Electrode_bc_C, Grounded_bc_C]
    f_potential_C = rho_poisson_C / epsilon
    F_potential_C = weak_form_Poisson_equation(dx, u[number_of_equations - 1], v[number_of_equations - 1],
                                               f_potential_C, r)

    Ion_flux = 0                                                                           # :346-355
    i = 1
    while i < number_of_species:
        Gamma.append(Flux(sign[i], u[i], D_si[i], mu_si[i], E, grad_diffusion=grad_diff[i],
                          logarithm_representation=True))
        if particle_species_type[i] == 'Ion':
            Ion_flux += Max(dot(Gamma[i], normal_plasma), 0)
        i += 1
    Gamma_en = Flux(sign[number_of_species - 1], u[0], 5.0 * D_si[number_of_species - 1] / 3.0,
                    5.0 * mu_si[number_of_species - 1] / 3.0, E, grad_diffusion=grad_diff[number_of_species - 1],
                    logarithm_representation=True)
    u_see_met = Expression('u_p', u_p=we_metalic, degree=1)

    f = Source_term('coupled', approximation, power_matrix, loss_matrix, gain_matrix, rate_coefficient_si, N0, u)   # :357-359
    f_en = Energy_Source_term('coupled', power_matrix, loss_matrix, gain_matrix, rate_coefficient_si, energy_loss,
                              u[0] / u[number_of_species - 1], N0, u)
    f_en += -dot(Flux(sign[number_of_species - 1], u[number_of_species - 1], D_si[number_of_species - 1],
                      mu_si[number_of_species - 1], E, grad_diffusion=grad_diff[number_of_species - 1],
                      logarithm_representation=True), E)

    i = 1                                                                                  # :361-364
    while i < number_of_species:
        F += weak_form_balance_equation_log_representation(equation_type[i], dt, dt_old, dx, u[i], u_old[i],
                                                           u_old1[i], v[i], f[i], Gamma[i], r, D_si[i])
        i += 1

    i = 0                                                                                  # :367-374
    while i < number_of_boundaries:
        j = 1
        while j < number_of_species:
            Fb = Boundary_flux('flux source', equation_type[j], particle_type[j], sign[j], mu_si[j], E,
                               normal_plasma, u[j], gamma[i], v[j], ds_plasma(i + 1), r, vth[j], ref_coeff[i][j],
                               Ion_flux)
            F += Fb
            j += 1
        i += 1

    F_en = weak_form_balance_equation_log_representation(equation_type[number_of_species - 1], dt, dt_old, dx,   # :377-383
                                                         u[0], u_old[0], u_old1[0], v[0], f_en, Gamma_en, r)
    i = 0
    while i < number_of_boundaries:
        F_en += Boundary_flux('flux source', equation_type[number_of_species - 1],
                              particle_type[number_of_species - 1], sign[number_of_species - 1],
                              5.0 * mu_si[number_of_species - 1] / 3.0, E, normal_plasma, u[0],
                              gamma[i] * u_see_met, v[0], ds_plasma(i + 1), r,
                              1.3333 * vth[number_of_species - 1], ref_coeff[i][number_of_species - 1], Ion_flux)
        i += 1

    F += F_en                                                                              # :385-386
    F += F_potential_C

    variable_list_new = [we_newV, u_newV[1], u_newV[2], u_newV[3], Phi]                    # :391-400
    variable_list_old = [we_oldV, u_oldV[1], u_oldV[2], u_oldV[3], Phi_old]
    variable_list_old1 = [we_old1V, u_old1V[1], u_old1V[2], u_old1V[3], Phi_old1]
    output_old_variable_list = [Phi_old, u_oldV[1], u_oldV[2], u_oldV[3]]
    output_new_variable_list = [Phi, u_newV[1], u_newV[2], u_newV[3]]
    output_files_variable_names = ['Phi', particle_species_file_names[1], particle_species_file_names[2],
                                   particle_species_file_names[3]]
    rev_assigner.assign(u_new, variable_list_new)
    rev_assigner.assign(u_old, variable_list_old)
    rev_assigner.assign(u_old1, variable_list_old1)

    F = action(F, u_new)                                                                   # :402-403
    J = derivative(F, u_new, u)

    problem = Problem(J, F, Voltage_bcs_C)                                                 # :408

    # the device's linear solver (no counterpart in the script: PETSc's defaults there)
    problem.device.setup_multigrid(nu=1)
    from fedm_amd.device import chebyshev_weights
    problem.device.set_fieldsplit(chebyshev_weights(8, 0.3, 2.2))   # tools/gd_cycle.py

    nonlinear_solver = PETScSNESSolver()                                                   # :411-414
    nonlinear_solver.parameters['relative_tolerance'] = relative_tolerance
    nonlinear_solver.parameters["linear_solver"] = linear_solver
    nonlinear_solver.parameters['maximum_iterations'] = maximum_iterations

    import contextlib, io
    steps = 0
    while t < T_final:                                                                     # :421-471
        t_old = t
        u_old1.assign(u_old)
        u_old.assign(u_new)
        assigner.assign(variable_list_old, u_old)
        redE_old.assign(redE)
        mean_energy_old1.assign(mean_energy_old)
        mean_energy_old.assign(mean_energy)

        redE.assign(project(1e21 * sqrt(dot(-grad(Phi), -grad(Phi))) / N0, solver_type='mumps'))
        Transport_coefficient_interpolation('update', mobility_dependence, N0, Tgas, mu, mu_x, mu_y, mean_energy_old, redE)
        Transport_coefficient_interpolation('update', Diffusion_dependence, N0, Tgas, D, D_x, D_y, mean_energy_old, redE, mu)
        Rate_coefficient_interpolation('update', k_dependence, rate_coefficient, k_x, k_y, mean_energy_old, redE, Te=0, Tgas=0)
        i = 0
        while i < len(k_y):
            if k_dependence[i] == "Umean":
                rate_coefficient_diff[i].vector()[:] = np.interp(mean_energy_old.vector()[:], k_x[i], k_diff[i])
            i += 1
        mu_diff[number_of_species - 1].vector()[:] = np.interp(mean_energy_old.vector()[:], mu_x[number_of_species - 1], mue_diff)
        D_diff[number_of_species - 1].vector()[:] = np.interp(mean_energy_old.vector()[:], D_x[number_of_species - 1], De_diff)

        with contextlib.redirect_stdout(io.StringIO() if quiet else sys.stdout):
            t = adaptive_solver(nonlinear_solver, problem, t, dt, dt_old, u_new, u_old, variable_list_new,
                                variable_list_old, assigner, error, files.error_file, max_error, ttol, dt_min,
                                time_dependent_arguments=[Phi_powered], approximation=approximation)
        log('time', files.model_log, t)
        mean_energy.vector()[:] = np.exp(we_newV.vector()[:] - u_newV[number_of_species - 1].vector()[:])

        t_output, t_output_step = file_output(t, t_old, t_output, t_output_step, t_output_list, t_output_step_list,
                                              file_type, output_file_list, output_files_variable_names,
                                              output_new_variable_list, output_old_variable_list, unit='us')

        dt_old1.time_step = dt_old.time_step
        dt_old.time_step = dt.time_step
        dt.time_step = adaptive_timestep(dt.time_step, max_error, ttol, dt_min, dt_max)
        max_error[2] = max_error[1]
        max_error[1] = max_error[0]
        steps += 1
    return dict(t=t, steps=steps, output=str(files.output_folder_path), species=particle_species_file_names,
                error_file=str(files.error_file), problem=problem)


if __name__ == "__main__":
    res = main(output_dir=sys.argv[1] if len(sys.argv) > 1 else "gd_output", quiet=False)
    print({k: v for k, v in res.items() if k != "problem"})
