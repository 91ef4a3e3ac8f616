"""Left against right (flexible) field-split preconditioning on the bench case: time per step,
iterations, and the distance between the two trajectories.  python tools/precond_side.py [n] [steps]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from fedm_amd.cases import streamer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
msh = streamer.mesh(n, 4.0)
states = {}
for side in ("left", "right", "left", "right"):
    prob = streamer.device_problem(msh.coords, msh.cells)
    prob.set_preconditioner_side(side)
    run = streamer.Stepper(prob)
    run.initialise()
    for _ in range(5):
        run.step()
    prob.get_state()
    n0 = (run.newton_iterations, run.linear_iterations)
    t0 = time.perf_counter()
    for _ in range(steps):
        run.step()
    prob.get_state()
    el = time.perf_counter() - t0
    print(f"{side:6s} {1e3 * el / steps:7.3f} ms/step  newton {(run.newton_iterations - n0[0]) / steps:.2f}"
          f"  gmres {(run.linear_iterations - n0[1]) / steps:.2f}", flush=True)
    states[side] = prob.get_state()
    del run, prob
d = np.abs(states["left"] - states["right"]).max(axis=0) / np.abs(states["left"]).max(axis=0)
print("relative difference of the states after", steps + 5, "steps:", d)
