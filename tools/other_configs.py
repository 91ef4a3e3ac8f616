"""Wall time of the non-headline BASELINE configurations on one GPU (configs[1], configs[2])."""
import sys, time, io, contextlib
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import time_of_flight as tof, glow_discharge as gdc
# configs[1]: time of flight 2-D, the example's mesh (160 x 320, 51 681 DOFs), 100 steps of 1 ps
t0 = time.time()
out = tof.run_harness(nx=160, ny=320, box_width=5e-4, box_height=1e-3, t0=2.5e-9, T_final=2.6e-9, t_output=2.6e-9)
el = time.time() - t0
el = out["loop_seconds"]          # the time loop alone (set-up, library load and the final projections apart)
print(f"ToF 2-D 160x320: {out['steps']} steps in {el:.2f} s = {out['steps'] / el:.1f} steps/s, "
      f"Newton {out['newton_iterations'] / out['steps']:.2f}/step, GMRES {out['linear_iterations'] / out['steps']:.1f}/step, "
      f"relative error {out['relative_error']:.4e}", flush=True)
# configs[2]: glow discharge at ~200k and ~400k DOFs
for n in (141, 200):
    with contextlib.redirect_stdout(io.StringIO()):
        case = gdc.Case(nx=n, ny=n, T_final=1.0)
    case.step()
    n0, l0, t0 = case.newton_iterations, case.linear_iterations, time.time()
    for _ in range(10):
        case.step()
    el = time.time() - t0
    print(f"GD {n}x{n} crossed ({case.prob.n} DOFs): {10 / el:.1f} steps/s, Newton {(case.newton_iterations - n0) / 10:.1f}/step, "
          f"GMRES {(case.linear_iterations - l0) / 10:.1f}/step, t = {case.t:.3e} s", flush=True)
