import sys
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
msh = streamer.mesh(int(sys.argv[1]) if len(sys.argv) > 1 else 96, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
try:
    U, its = streamer.initialise(prob)
    print("poisson its", its)
except Exception as e:
    print("FAILED", e)
