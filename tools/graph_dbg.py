import sys
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
msh = streamer.mesh(n, 4.0)
for mode in ("graphs", "plain"):
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob); st.initialise()
    if mode == "plain":
        prob.profile(2)
    out = []
    for _ in range(4):
        l0, n0 = st.linear_iterations, st.newton_iterations
        st.step()
        out.append((st.newton_iterations - n0, st.linear_iterations - l0))
    print(mode, out, float(np.abs(prob.get_state()).sum()))
    prob.close()
