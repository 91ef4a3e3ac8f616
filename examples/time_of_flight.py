#!/usr/bin/env python3
"""Time-of-flight verification case -- the reference's examples/time_of_flight/fedm-tof.py with the
same sequence of calls, on the MI355X device path: one electron balance equation in the logarithmic
variable, the flux written out by hand, the analytic source and solution as C++ Expression strings.
`from fedm_amd.forms import *` stands for `from dolfin import *`; the Expression strings are the
reference's, evaluated by the arithmetic subset of fedm_amd.forms.compile_cpp_expression (no JIT).
`main` returns what the reference's test harness (tests/integrated_tests/time_of_flight/fedm_tof.py)
returns: the projected numerical and exact densities and their relative L2 difference.
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd.forms import *                      # noqa: F401,F403,E402  (stands for `from dolfin import *`)
from fedm_amd.physical_constants import *         # noqa: F401,F403,E402
from fedm_amd.file_io import *                    # noqa: F401,F403,E402
from fedm_amd.functions import *                  # noqa: F401,F403,E402


def main(nx=160, ny=320, box_width=5e-4, box_height=1e-3, t0=2.5e-9, T_final=3e-9, t_output=3e-9,
         dt_init=1e-12, output_dir=None, quiet=False):
    linear_solver = "mumps"
    maximum_iterations = 50
    relative_tolerance = 1e-10

    model = 'Time_of_flight'
    gas = 'Air'
    Tgas = 300.0
    p0 = 760.0
    N0 = p0 * 3.21877e22
    if output_dir is not None:
        files.output_folder_path = Path(output_dir)

    particle_species_type = ['electrons', 'analytical solution']
    M = me
    charge = -elementary_charge
    equation_type = ['drift-diffusion-reaction']
    wez = 1.7e5
    De = 0.12
    alpha_e = 5009.51

    log('properties', files.model_log, gas, model, particle_species_type, M, charge)
    vtkfile_u = output_files('pvd', 'number density', particle_species_type)

    t = t0
    dt = Expression("time_step", time_step=dt_init, degree=0)
    dt_old = Expression("time_step", time_step=1e30, degree=0)       # BDF1 for the first step
    t_output_step = 1e-9

    r = Expression('x[0]', degree=1)
    z = Expression('x[1]', degree=1)

    mesh = RectangleMesh(Point(0, 0), Point(box_width, box_height), nx, ny)
    mesh_statistics(mesh)
    h = MPI.max(MPI.comm_world, mesh.hmax())

    log('conditions', files.model_log, dt.time_step, 'None', p0, box_height, N0, Tgas)
    log('initial time', files.model_log, t)

    V = FunctionSpace(mesh, 'P', 1)
    W = VectorFunctionSpace(mesh, 'P', 1)

    u = TrialFunction(V)
    v = TestFunction(V)
    u_old = Function(V)
    u_old1 = Function(V)
    u_new = Function(V)

    u_analytical = Expression(
        'std::log(exp(-(pow(x[1]-w*t, 2)+pow(x[0], 2))/(4.0*D*t)+alpha*w*t)/pow(4*D*t*pi,1.5))',
        D=De, w=wez, alpha=alpha_e, t=t, pi=pi, degree=3)
    u_old.assign(interpolate(u_analytical, V))
    u_old1.assign(interpolate(u_analytical, V))

    w = interpolate(Constant(('0', wez)), W)
    D = interpolate(Constant(De), V)
    alpha_eff = interpolate(Constant(alpha_e), V)                       # noqa: F841 (as in the reference)

    Gamma = -grad(D * exp(u)) + w * exp(u)
    f = Expression(
        'exp(-(pow(x[1]-w*t, 2)+pow(x[0], 2))/(4.0*D*t)+alpha*w*t)*(w*alpha)/(8*pow(pi,1.5)*pow(D*t, 1.5))',
        D=De, w=wez, alpha=alpha_e, t=t, pi=pi, degree=2)

    F = weak_form_balance_equation_log_representation(equation_type[0], dt, dt_old, dx, u, u_old, u_old1,
                                                      v, f, Gamma, r)

    u_new.assign(interpolate(Expression(
        'std::log(exp(-(pow(x[1]-w*t, 2)+pow(x[0], 2))/(4.0*D*t)+alpha*w*t)/pow(4.0*D*t*pi,1.5) + DOLFIN_EPS)',
        D=De, w=wez, alpha=alpha_e, t=t, pi=pi, degree=2), V))

    F = action(F, u_new)
    J = derivative(F, u_new, u)
    problem = Problem(J, F, [])

    nonlinear_solver = PETScSNESSolver()
    nonlinear_solver.parameters['relative_tolerance'] = relative_tolerance
    nonlinear_solver.parameters["linear_solver"] = linear_solver
    nonlinear_solver.parameters['maximum_iterations'] = maximum_iterations

    n_exact = Function(V)
    n_num = Function(V)
    relative_error = None

    while abs(t - T_final) / T_final > 1e-6:
        u_old1.assign(u_old)
        u_old.assign(u_new)
        t += dt.time_step

        log('time', files.model_log, t)
        if not quiet:
            print_time(t)

        f.t = t
        u_analytical.t = t

        nonlinear_solver.solve(problem, u_new.vector())

        if abs(t - t_output) / t_output <= 1e-6:
            n_exact.assign(project(exp(u_analytical), V, solver_type='mumps'))
            n_num.assign(project(exp(u_new), V, solver_type='mumps'))
            relative_error = errornorm(n_num, n_exact, 'l2') / norm(n_exact, 'l2')
            with open(files.error_file, "a") as f_err:
                f_err.write('h_max = ' + str(h) + '\t dt = ' + str(dt.time_step)
                            + '\t relative_error = ' + str(relative_error) + '\n')
            if MPI.rank(MPI.comm_world) == 0 and not quiet:
                print(relative_error)
            vtkfile_u[0] << (n_num, t)
            vtkfile_u[1] << (n_exact, t)
            t_output += t_output_step

        if t > (t0 + dt_init):
            dt_old.time_step = dt.time_step          # BDF2 after the first step
    return n_num.vector().copy(), n_exact.vector().copy(), relative_error


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=160)
    ap.add_argument("--ny", type=int, default=320)
    ap.add_argument("--t-final", type=float, default=3e-9)
    ap.add_argument("--output-dir", default=None)
    a = ap.parse_args()
    print("relative_error =", main(a.nx, a.ny, T_final=a.t_final, t_output=a.t_final,
                                   output_dir=a.output_dir)[2])
