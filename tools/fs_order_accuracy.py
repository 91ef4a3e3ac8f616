"""Does the order of the field split change the computed solution?  Runs the bench case for 220
steps with the species-first (lower) and the potential-first (upper) order at the default
ksp_rtol = 1e-5 and compares both with a run at ksp_rtol = 1e-9 (either order solves J d = -F; the
unscaled residual norm PETSc and this library test is dominated by the species rows, 1e18 against
1e-8 for the Poisson row, so this is the check that the potential is still solved).
Result (round 2, MI355X): lower 1e-5: 6e-6 (electrons, max norm relative to max |u|), 34 Krylov steps
per time step late; upper 1e-5: 3.8e-3, 12 steps; switching from upper to lower for the last Newton
iterations of a time step (tried, removed) does not recover the accuracy (2.3e-3 .. 3.5e-3).
python tools/fs_order_accuracy.py [n=576] [steps=220]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 220
msh = streamer.mesh(n, 4.0)
runs = {}
for name, order, rtol in (("reference: upper, ksp_rtol 1e-9", "upper", 1e-9), ("upper, 1e-5", "upper", 1e-5),
                          ("lower, 1e-5", "lower", 1e-5), ("lower, 1e-9", "lower", 1e-9), ("upper, 1e-7", "upper", 1e-7)):
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob)
    st.solver.parameters["krylov_relative_tolerance"] = rtol
    st.initialise()
    prob.set_fieldsplit_order(order)
    import time
    while st.steps < steps - 20:
        st.step()
    n0, l0, t0 = st.newton_iterations, st.linear_iterations, time.perf_counter()
    while st.steps < steps:
        st.step()
    late = (f"last 20 steps: newton/step {(st.newton_iterations - n0) / 20:.2f} gmres/step {(st.linear_iterations - l0) / 20:.2f} "
            f"ms/step {50 * (time.perf_counter() - t0):.3f}")
    U = prob.get_state()
    runs[name] = U
    ref = runs["reference: upper, ksp_rtol 1e-9"]
    diff = np.abs(U - ref).max(axis=0) / np.abs(ref).max(axis=0)
    print(f"{name:34s} t = {st.t:.6e}  newton {st.newton_iterations} gmres {st.linear_iterations}  "
          f"max |u - u_ref| / max |u_ref| per field (ions, electrons, potential): "
          + " ".join(f"{d:.2e}" for d in diff) + "  " + late, flush=True)
    prob.close()
