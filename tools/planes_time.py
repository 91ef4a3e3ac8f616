"""Time of the preconditioner's set-up behind a Jacobian assembly (`species_planes_kernel`: D_uu^-1, the half-precision
species planes, the single-precision coupling plane), alone and behind the assembly as in a Newton iteration.
python tools/planes_time.py [mesh=-4]     (-k: refined unstructured mesh with k um in the channel; n: n x n tensor mesh;
FEDM_HIP_LIB selects an experiment build)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedm_amd.cases import streamer
n = int(sys.argv[1]) if len(sys.argv) > 1 else -4
if n < 0:
    h = -n * 1e-6
    msh = streamer.refined_mesh(h, growth=0.1, channel=(0.0, 100.0 * h) + streamer.CHANNEL[2:])
else:
    msh = streamer.mesh(n, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
st = streamer.Stepper(prob)
st.initialise()
for _ in range(3):
    st.step()
out = {}
for name, kind in (("FJ", 0), ("planes", 4), ("FJ_then_planes", 5)):
    prob.time_kernel(kind, 5)
    out[name + "_us"] = round(1e3 * min(prob.time_kernel(kind, 40) for _ in range(3)), 2)
print(os.path.basename(os.environ.get("FEDM_HIP_LIB", "default")), out, flush=True)
