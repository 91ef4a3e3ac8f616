import sys, time, json
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
n = int(sys.argv[1]); nsteps = int(sys.argv[2])
msh = streamer.mesh(n, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
st = streamer.Stepper(prob)
st.initialise()
t0 = time.time(); hist = []
for k in range(nsteps):
    n0, l0 = st.newton_iterations, st.linear_iterations
    st.step()
    if k % 50 == 0 or k == nsteps - 1:
        U = prob.get_state()
        rows = st.log_rows()
        hist.append(dict(step=k, t=st.t, dt=st.dt.time_step, newton=st.newton_iterations - n0, gmres=st.linear_iterations - l0,
                         attempts=len(rows), ne_max=float(np.exp(U[:,1].max())), ni_max=float(np.exp(U[:,0].max())), wall=time.time()-t0))
        print(hist[-1], flush=True)
json.dump(hist, open('gpurun_out/longrun_%d.json' % n, 'w'))
