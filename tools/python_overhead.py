"""How much of a time step is spent in Python between the library's calls (the GPU idles through most of it):
wall time per step against the time inside the ctypes calls.  usage: python tools/python_overhead.py [N=576]"""
import sys
import time

sys.path.insert(0, ".")
from fedm_amd.cases import streamer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
msh = streamer.mesh(n, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
st = streamer.Stepper(prob)
st.initialise()
inside = [0.0]


def timed(fn):
    def call(*a):
        t0 = time.perf_counter()
        r = fn(*a)
        inside[0] += time.perf_counter() - t0
        return r
    return call


class Lib:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        f = timed(getattr(self._lib, name))
        setattr(self, name, f)
        return f


for _ in range(5):
    st.step()
prob.lib = Lib(prob.lib)
prob.get_state()
inside[0] = 0.0
t0 = time.perf_counter()
K = 40
for _ in range(K):
    st.step()
wall = time.perf_counter() - t0
print(f"{1e3 * wall / K:.3f} ms per step, {1e3 * inside[0] / K:.3f} ms inside the library's calls, "
      f"{1e6 * (wall - inside[0]) / K:.1f} us of Python per step")
