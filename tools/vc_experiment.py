"""V-cycle variants on the bench case: iterations and wall time per step."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
msh = streamer.mesh(576, 4.0)
for name, mg in [("V(1,1)", dict(nu=1)), ("V(0,1)", dict(nu=-1)), ("V(0,2)", dict(nu=-2)), ("V(2,2)", dict(nu=2))]:
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob); st.initialise()      # initial Poisson solve: CG needs the symmetric V(1,1)
    prob.setup_multigrid(**mg)
    try:
        st.step()
        n0 = st.linear_iterations; t0 = time.time()
        for _ in range(10): st.step()
        print(name, "gmres/step", (st.linear_iterations - n0) / 10, "ms/step", round((time.time() - t0) * 100, 3), flush=True)
    except Exception as e:
        print(name, "FAILED", e, flush=True)
    prob.close()
