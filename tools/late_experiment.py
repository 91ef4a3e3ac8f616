"""Which part of the preconditioner degrades late in the streamer run?"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
from fedm_amd.device import chebyshev_weights
msh = streamer.mesh(576, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
st = streamer.Stepper(prob); st.initialise()
for _ in range(250): st.step()
U = [prob.get_state(), prob.get_state_old()]
def measure(name):
    n0, l0 = st.newton_iterations, st.linear_iterations
    t0 = time.time()
    for _ in range(5): st.step()
    print(name, "newton/step", (st.newton_iterations - n0) / 5, "gmres/step", (st.linear_iterations - l0) / 5,
          "ms/step", round((time.time() - t0) * 200, 2), flush=True)
measure("default cheb4 V(1,1)")
prob.set_fieldsplit(chebyshev_weights(8)); measure("cheb8")
prob.set_fieldsplit(chebyshev_weights(4, 0.3, 2.5)); measure("cheb4 [0.3,2.5]")
prob.set_fieldsplit(chebyshev_weights(6, 0.3, 2.5)); measure("cheb6 [0.3,2.5]")
prob.set_fieldsplit(chebyshev_weights(4)); prob.setup_multigrid(nu=2); measure("cheb4 V(2,2)")
prob.set_fieldsplit([1.0]); prob.setup_multigrid(nu=1); measure("jacobi1 V(1,1)")
