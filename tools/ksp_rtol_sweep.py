"""Newton-Krylov forcing term: time per step and iteration counts of the bench case against the
relative tolerance of the linear solves (PETSc's default, 1e-5, is what the library uses), early in
the run and from step 200 on.  The Newton stopping test (rtol 1e-4 on |F|) is the same for all."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer

msh = streamer.mesh(576, 4.0)
ref = None
for rtol in (1e-5, 1e-4, 1e-3, 1e-2):
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob)
    st.solver.parameters["krylov_relative_tolerance"] = rtol
    st.initialise()
    out = []
    for name, upto in (("early", 25), ("late", 220)):
        while st.steps < upto - 20:
            st.step()
        n0, l0, t0 = st.newton_iterations, st.linear_iterations, time.perf_counter()
        for _ in range(20):
            st.step()
        dt = time.perf_counter() - t0
        out.append((name, round(1e3 * dt / 20, 3), (st.newton_iterations - n0) / 20, (st.linear_iterations - l0) / 20))
    U = prob.get_state()
    if ref is None:
        ref = U
    diff = np.abs(U - ref).max(axis=0) / np.abs(ref).max(axis=0)
    print(f"ksp_rtol {rtol:g}: {out}  t = {st.t:.4e}  rel. diff to 1e-5 run after 220 steps {diff}", flush=True)
    prob.close()
