import sys, time
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import glow_discharge as gdc
n = int(sys.argv[1]); nsteps = int(sys.argv[2])
t0 = time.time()
case = gdc.Case(nx=n, ny=n, T_final=1.0)
print("setup", time.time() - t0, "dofs", case.prob.n, "mg levels", case.prob.multigrid_levels, flush=True)
t0 = time.time()
for k in range(nsteps):
    n0, l0 = case.newton_iterations, case.linear_iterations
    t1 = time.time()
    case.step()
    print(k, "t", case.t, "dt", case.dt_old.time_step, "newton", case.newton_iterations - n0, "gmres", case.linear_iterations - l0,
          "err", case.error[0], "step s", round(time.time() - t1, 3), flush=True)
print("total", time.time() - t0)
