"""Species-block polynomial of the field split against GMRES iterations and time per step on the
bench case, early (steps 5-25) and later in the run (steps 250-270); weights are set after
initialise(), which installs the default."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
from fedm_amd.device import chebyshev_weights
msh = streamer.mesh(576, 4.0)
degrees = [int(a) for a in sys.argv[1:]] or [4, 6, 7, 8, 10, 12]
for deg in degrees:
    w = chebyshev_weights(deg)
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob)
    st.initialise()
    prob.set_fieldsplit(w)
    out = []
    for start in (5, 250):
        while st.steps < start:
            st.step()
        prob.get_state()
        n0, m0 = st.linear_iterations, st.newton_iterations
        t0 = time.time()
        for _ in range(20):
            st.step()
        prob.get_state()
        out.append(f"steps {start}-{start + 20}: newton {(st.newton_iterations - m0) / 20:.2f} gmres "
                   f"{(st.linear_iterations - n0) / 20:.2f} per step, {(time.time() - t0) * 50:.3f} ms/step")
    print(f"cheb{deg}", " | ".join(out), flush=True)
    prob.close()
