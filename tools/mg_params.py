"""Multigrid parameter sweep on the bench case (smoother damping, strength threshold, prolongator damping)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
from fedm_amd import amg
msh = streamer.mesh(576, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
st = streamer.Stepper(prob); st.initialise()
U = prob.get_state()
def run(name, **mg):
    try:
        _run(name, **mg)
    except Exception as e:
        print(f"{name:28s} FAILED {e}", flush=True)


def _run(name, **mg):
    prob.set_state(U, U, U)
    st.t = 0.0; st.dt.time_step = 5e-12; st.dt_old.time_step = 1e30; st.max_error[:] = [1, 1, 1]
    levels = prob.setup_multigrid(**mg)
    st.step()
    n0 = st.linear_iterations; t0 = time.time()
    for _ in range(10): st.step()
    print(f"{name:28s} levels {levels} gmres/step {(st.linear_iterations - n0) / 10:6.2f} ms/step {(time.time() - t0) * 100:6.3f}", flush=True)
run("default nu=1 w=.67 th=.08", nu=1)
for w in (0.55, 0.75, 0.85, 0.95):
    run(f"omega={w}", nu=1, omega=w)
for th in (0.04, 0.15):
    run(f"theta={th}", nu=1, theta=th)
run("max_coarse=600", nu=1, max_coarse=600)
