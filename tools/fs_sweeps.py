import sys, time
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
from fedm_amd.device import chebyshev_weights
msh = streamer.mesh(576, 4.0)
cases = [("jacobi1", [1.0]), ("damped3 0.7", [0.7] * 3), ("cheb2", chebyshev_weights(2)), ("cheb3", chebyshev_weights(3)),
         ("cheb3 [0.4,2.2]", chebyshev_weights(3, 0.4, 2.2)), ("cheb4", chebyshev_weights(4)), ("cheb3 rev", chebyshev_weights(3)[::-1])]
for name, w in cases:
    prob = streamer.device_problem(msh.coords, msh.cells)
    prob.set_fieldsplit(w)
    st = streamer.Stepper(prob); st.initialise(); st.step()
    n0 = st.linear_iterations; t0 = time.time()
    for _ in range(10): st.step()
    print(name, np.round(w, 3), "gmres/step", (st.linear_iterations - n0) / 10, "ms/step", round((time.time() - t0) * 100, 3), flush=True)
    prob.close()
