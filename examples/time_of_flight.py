#!/usr/bin/env python3
"""Electron swarm drifting in a uniform field (time-of-flight experiment): one drift-diffusion-reaction
balance equation in the logarithmic density, checked against the closed-form solution

    n(r, z, t) = exp(-((z - w t)^2 + r^2) / (4 D t) + alpha w t) / (4 pi D t)^(3/2).

A driver of our own for the fedm_amd facade; the conditions are those of the reference's
examples/time_of_flight/fedm-tof.py (w = 1.7e5 m/s, D = 0.12 m^2/s, alpha = 5009.51 1/m, pulse released
at t = 2.5 ns, fixed 1 ps steps, BDF1 for the first two steps), so that a run with the sizes of the
reference's test harness reproduces its golden numbers (tests/test_gpu_parity.py).  `main` returns the
projected numerical and exact densities and their relative L2 difference.

    python examples/time_of_flight.py --nx 160 --ny 320 --t-final 3e-9
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd import forms as fem                          # noqa: E402  (what `dolfin` is to a FEDM script)
from fedm_amd import file_io, functions as fedm            # noqa: E402
from fedm_amd.physical_constants import elementary_charge, me      # noqa: E402

DRIFT, DIFFUSION, IONISATION = 1.7e5, 0.12, 5009.51       # w [m/s], D [m^2/s], alpha [1/m]
PULSE = "exp(-(pow(x[1]-w*t, 2)+pow(x[0], 2))/(4.0*D*t)+alpha*w*t)"     # numerator of the swarm density


def swarm_expression(template, t, degree):
    """An Expression in (x, t) with the swarm parameters bound; `t` is advanced by the time loop."""
    return fem.Expression(template.replace("PULSE", PULSE), D=DIFFUSION, w=DRIFT, alpha=IONISATION, t=t,
                          pi=fem.pi, degree=degree)


def main(nx=160, ny=320, box_width=5e-4, box_height=1e-3, t0=2.5e-9, T_final=3e-9, t_output=3e-9,
         dt_init=1e-12, output_dir=None, quiet=False):
    if output_dir is not None:
        file_io.files.output_folder_path = Path(output_dir)
    labels = ["electrons", "analytical solution"]
    gas_density = 760.0 * 3.21877e22
    file_io.log("properties", file_io.files.model_log, "Air", "Time_of_flight", labels, me, -elementary_charge)
    writers = file_io.output_files("pvd", "number density", labels)

    mesh = fem.RectangleMesh(fem.Point(0, 0), fem.Point(box_width, box_height), nx, ny)
    file_io.mesh_statistics(mesh)
    h_max = fem.MPI.max(fem.MPI.comm_world, mesh.hmax())
    step = fem.Expression("time_step", time_step=dt_init, degree=0)
    step_before = fem.Expression("time_step", time_step=1e30, degree=0)       # 1e30: the BDF2 formula reduces to BDF1
    file_io.log("conditions", file_io.files.model_log, step.time_step, "None", 760.0, box_height, gas_density, 300.0)
    file_io.log("initial time", file_io.files.model_log, t0)

    V = fem.FunctionSpace(mesh, "P", 1)
    trial, test = fem.TrialFunction(V), fem.TestFunction(V)
    now, before, before2 = fem.Function(V), fem.Function(V), fem.Function(V)
    exact_log = swarm_expression("std::log(PULSE/pow(4*D*t*pi,1.5))", t0, degree=3)
    for past in (before, before2):
        past.assign(fem.interpolate(exact_log, V))
    now.assign(fem.interpolate(swarm_expression("std::log(PULSE/pow(4.0*D*t*pi,1.5) + DOLFIN_EPS)", t0, degree=2), V))

    # constant drift along z and constant diffusion, as nodal fields; the flux of n = exp(u) by hand
    velocity = fem.interpolate(fem.Constant(("0", DRIFT)), fem.VectorFunctionSpace(mesh, "P", 1))
    diffusion = fem.interpolate(fem.Constant(DIFFUSION), V)
    flux = -fem.grad(diffusion * fem.exp(trial)) + velocity * fem.exp(trial)
    source = swarm_expression("PULSE*(w*alpha)/(8*pow(pi,1.5)*pow(D*t, 1.5))", t0, degree=2)     # alpha w n
    radius = fem.Expression("x[0]", degree=1)
    form = fedm.weak_form_balance_equation_log_representation("drift-diffusion-reaction", step, step_before, fem.dx,
                                                              trial, before, before2, test, source, flux, radius)
    residual = fem.action(form, now)
    problem = fedm.Problem(fem.derivative(residual, now, trial), residual, [])
    newton = fedm.PETScSNESSolver()
    newton.parameters.update(relative_tolerance=1e-10, maximum_iterations=50, linear_solver="mumps")

    density, density_exact, relative_error = fem.Function(V), fem.Function(V), None
    t, next_output = t0, t_output
    while abs(t - T_final) / T_final > 1e-6:
        before2.assign(before)
        before.assign(now)
        t += step.time_step
        file_io.log("time", file_io.files.model_log, t)
        if not quiet:
            file_io.print_time(t)
        source.t = exact_log.t = t
        newton.solve(problem, now.vector())
        if abs(t - next_output) / next_output <= 1e-6:
            density_exact.assign(fem.project(fem.exp(exact_log), V, solver_type="mumps"))
            density.assign(fem.project(fem.exp(now), V, solver_type="mumps"))
            relative_error = fem.errornorm(density, density_exact, "l2") / fem.norm(density_exact, "l2")
            with open(file_io.files.error_file, "a") as log_file:
                log_file.write(f"h_max = {h_max}\t dt = {step.time_step}\t relative_error = {relative_error}\n")
            if not quiet:
                print(relative_error)
            writers[0] << (density, t)
            writers[1] << (density_exact, t)
            next_output += 1e-9
        if t > t0 + dt_init:
            step_before.time_step = step.time_step          # second-order BDF from the third step on
    return density.vector().copy(), density_exact.vector().copy(), relative_error


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--nx", type=int, default=160)
    ap.add_argument("--ny", type=int, default=320)
    ap.add_argument("--t-final", type=float, default=3e-9)
    ap.add_argument("--output-dir", default=None)
    a = ap.parse_args()
    print("relative_error =", main(a.nx, a.ny, T_final=a.t_final, t_output=a.t_final, output_dir=a.output_dir)[2])
