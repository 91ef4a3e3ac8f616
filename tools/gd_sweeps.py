import sys, time, io, contextlib
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import glow_discharge as gdc
from fedm_amd.device import chebyshev_weights
n = 141
for name, w, side in [("jacobi left", None, "left"), ("cheb3 left", chebyshev_weights(3), "left"), ("cheb4 left", chebyshev_weights(4), "left"),
                      ("cheb6 left", chebyshev_weights(6), "left"), ("cheb4 [0.3,2.5] left", chebyshev_weights(4, 0.3, 2.5), "left"),
                      ("cheb4 right", chebyshev_weights(4), "right"), ("damped 0.7x3 left", [0.7]*3, "left")]:
    with contextlib.redirect_stdout(io.StringIO()):
        case = gdc.Case(nx=n, ny=n, T_final=1.0)
    if w is not None:
        case.prob.set_fieldsplit(w)
    case.prob.set_preconditioner_side(side)
    try:
        case.step()
        n0, l0, t0 = case.newton_iterations, case.linear_iterations, time.time()
        for _ in range(10):
            case.step()
        el = time.time() - t0
        print(f"{name:24s} {10 / el:6.1f} steps/s, Newton {(case.newton_iterations - n0) / 10:.1f}/step, GMRES {(case.linear_iterations - l0) / 10:.1f}/step", flush=True)
    except Exception as e:
        print(name, "FAILED", str(e)[:100], flush=True)
    case.prob.close()
