"""Preconditioner settings for the developed streamer on the refined unstructured mesh: the run is carried
to step S with the defaults, its state is kept, and every configuration continues from there for K steps
(ms per step, Newton and GMRES iterations per step).

usage: python tools/late_sweep.py [H_FINE=4e-6] [S=900] [K=40] [tensor]      (tensor: the 576x576 bench mesh instead)"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from fedm_amd.cases import streamer
from fedm_amd.device import chebyshev_weights

h = float(sys.argv[1]) if len(sys.argv) > 1 else 4e-6
S = int(sys.argv[2]) if len(sys.argv) > 2 else 900
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tensor = len(sys.argv) > 4 and sys.argv[4] == "tensor"
msh = streamer.mesh(576, 4.0) if tensor else streamer.refined_mesh(h, growth=0.1, channel=(0.0, 100.0 * h) + streamer.CHANNEL[2:])


def fresh(multigrid, main, hard, switch=(5.0, 3.5)):
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob)
    U = np.zeros((prob.nv, 3))
    U[:, 0], U[:, 1] = streamer.initial_log_densities(prob.coords)
    prob.set_state(U, U, U)
    prob.setup_multigrid(**multigrid)
    prob.set_fieldsplit(main, hard_weights=hard, switch_above=switch[0], back_below=switch[1])
    return prob, st


default_mg = dict(streamer.MULTIGRID)
prob, st = fresh(default_mg, chebyshev_weights(6), chebyshev_weights(4))
prob.poisson_solve(rtol=1e-12)
U = prob.get_state()
prob.set_state(U, U, U)
t0 = time.time()
for _ in range(S):
    st.step()
print(f"reference run: {S} steps to t = {st.t:.3e} in {time.time() - t0:.1f} s", flush=True)
keep = dict(u=prob.get_state(), uo=prob.get_state_old(), t=st.t, dt=st.dt.time_step, dt_old=st.dt_old.time_step,
            max_error=list(st.max_error), error=list(st.error))
# u_old1 is not downloadable: continue every configuration (the default too) from (u, u_old, u_old)
prob.close()

no_hard = dict(default_mg, hard_poly_degree=None)
configs = {
    "default: Chebyshev(6) / hard: Chebyshev(4) + polynomial cycle (2)": (default_mg, chebyshev_weights(6), chebyshev_weights(4)),
    "hard: Chebyshev(6) + polynomial cycle (2)": (default_mg, chebyshev_weights(6), chebyshev_weights(6)),
    "no hard mode: Chebyshev(6) + V(1,1) throughout": (no_hard, chebyshev_weights(6), None),
    "no hard mode: Chebyshev(6) + V(1,1) omega 0.95": (dict(no_hard, omega=0.95), chebyshev_weights(6), None),
    "no hard mode: Chebyshev(6) on [0.6, 2.0] + V(1,1)": (no_hard, chebyshev_weights(6, 0.6, 2.0), None),
    "no hard mode: Chebyshev(8) on [0.5, 2.5] + V(1,1)": (no_hard, chebyshev_weights(8, 0.5, 2.5), None),
    "polynomial cycle (2) as the main cycle, Chebyshev(6)": (dict(no_hard, poly_degree=2), chebyshev_weights(6), None),
}
# (round 3, refined mesh at 4.5 ns, ms per step / GMRES per step: 6.8 / 30.4 with the measured choice [8.9 / 35.9 with
#  FEDM_FS_POLICY=counts], 7.4 / 28.2, 6.7 / 30.4, 6.5 / 29.2, 6.5 / 29.0, 7.1 / 29.0, 7.4 / 28.0; tensor-product mesh at
#  1 ns: 5.9 / 28.0, 6.2 / 26.8, 6.9 / 36.0, 8.2 / 41.7, -, -, 6.1 / 26.4.  The interval of the species polynomial hardly
#  matters around [0.5, 2]; below 0.4 or with degree 8 on a short interval the sweeps amplify and GMRES stalls.)
for name, (mg, main, hard) in configs.items():
    prob, st = fresh(mg, main, hard)
    prob.set_state(keep["u"], keep["uo"], keep["uo"])
    st.t, st.dt.time_step, st.dt_old.time_step = keep["t"], keep["dt"], keep["dt_old"]
    st.max_error[:], st.error[:] = keep["max_error"], keep["error"]
    for _ in range(5):
        st.step()                     # settle (hard-mode switch, graphs)
    n0, l0 = st.newton_iterations, st.linear_iterations
    t0 = time.perf_counter()
    for _ in range(K):
        st.step()
    el = time.perf_counter() - t0
    print(f"{name:66s} {1e3 * el / K:6.2f} ms/step   Newton {(st.newton_iterations - n0) / K:4.2f}  GMRES "
          f"{(st.linear_iterations - l0) / K:5.1f} per step   t = {st.t:.3e}", flush=True)
    prob.close()
