"""Glow discharge (402k DOFs): V(1,1) against the polynomial-smoother cycle and species polynomial degrees.
python tools/gd_cycle.py"""
import sys, io, contextlib, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd.cases import glow_discharge as gdc
from fedm_amd.device import chebyshev_weights
for name, mg, deg, side in (("V(1,1), Chebyshev(4), left  [round-1 default]", dict(nu=1), 4, "left"),
                            ("V(1,1), Chebyshev(6), left", dict(nu=1), 6, "left"),
                            ("V(1,1), Chebyshev(8), left", dict(nu=1), 8, "left"),
                            ("V(1,1), Chebyshev(10), left", dict(nu=1), 10, "left"),
                            ("V(1,1), Chebyshev(6) on [0.3, 2.2], left", dict(nu=1), (6, 0.3, 2.2), "left"),
                            ("V(1,1), Chebyshev(8) on [0.3, 2.2], left", dict(nu=1), (8, 0.3, 2.2), "left"),
                            ("Chebyshev(2,2) cycle, Chebyshev(6), left", dict(nu=1, poly_degree=2), 6, "left")):
    with contextlib.redirect_stdout(io.StringIO()):
        case = gdc.Case(nx=200, ny=200, T_final=1.0)
    case.prob.setup_multigrid(**mg)
    case.prob.set_fieldsplit(chebyshev_weights(*deg) if isinstance(deg, tuple) else chebyshev_weights(deg))
    case.prob.set_preconditioner_side(side)
    for _ in range(3):
        case.step()
    n0, l0, t0 = case.newton_iterations, case.linear_iterations, time.perf_counter()
    for _ in range(20):
        case.step()
    dt = time.perf_counter() - t0
    print(f"{name:46s} {20 / dt:6.1f} steps/s  newton/step {(case.newton_iterations - n0) / 20:.2f}  gmres/step "
          f"{(case.linear_iterations - l0) / 20:.1f}", flush=True)
    case.prob.close()
