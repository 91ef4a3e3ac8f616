"""Time of one multigrid cycle (replayed graph, back to back) for the installed variants; run under
rocprofv3 --kernel-trace --stats for the per-kernel split.  python tools/cycle_time.py [n=576]"""
import sys
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
msh = streamer.mesh(n, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
U0 = np.zeros((prob.nv, 3))
U0[:, 0], U0[:, 1] = streamer.initial_log_densities(prob.coords)
prob.set_state(U0, U0, U0)
for name, kw in (("V(1,1) Jacobi 0.85", dict(nu=1, omega=0.85)), ("Chebyshev degree 2", dict(poly_degree=2)),
                 ("Chebyshev degree 3", dict(poly_degree=3, poly_fraction=10.0))):
    prob.setup_multigrid(**kw)
    prob.poisson_solve(rtol=1e-6)
    t = min(prob.time_kernel(3, 200) for _ in range(3))
    print(f"{name:24s} {1e3 * t:7.2f} us per cycle", flush=True)
