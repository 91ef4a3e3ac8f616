"""Durations of the tiled species-sweep kernel for several tile sizes and layer counts (576x576 bench mesh):
run under `rocprofv3 --kernel-trace` and read with tools/fs_tiles_probe_read.py, or plainly for wall times.

usage: python tools/fs_tiles_probe.py [N=576]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from fedm_amd.cases import streamer
from fedm_amd.device import chebyshev_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
msh = streamer.mesh(n, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
U = np.zeros((prob.nv, 3))
U[:, 0], U[:, 1] = streamer.initial_log_densities(prob.coords)
prob.set_state(U, U, U)
prob.set_step(5e-12, 1e30)
prob.setup_multigrid(**streamer.MULTIGRID)
prob.set_fieldsplit(chebyshev_weights(6))
prob.jacobian()
t = np.random.default_rng(0).standard_normal(3 * prob.nv)
for slices, layers, threads in ((0, 0, 0), (8, 5, 512), (8, 5, 256), (8, 4, 512), (8, 3, 512), (8, 3, 256), (8, 2, 512), (8, 2, 256),
                                (8, 1, 256), (8, 1, 512), (4, 5, 256), (4, 3, 128), (4, 3, 256), (4, 2, 128), (4, 1, 256)):
    if slices == 0:
        prob.configure_fieldsplit_tiles(False)
    else:
        prob.configure_fieldsplit_tiles(True, slices, layers, threads)
    info = prob.fieldsplit_tiles()
    prob.fieldsplit_apply(t)
    t0 = time.perf_counter()
    prob.fieldsplit_apply(t)
    print(f"CONFIG slices {slices} layers {layers} threads {threads}: {info}  wall {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
prob.close()
