"""Phase breakdown of the one-pass patch assembly kernel (assemble3.hip, one workgroup per patch): needs a
-DFEDM_PHASE_TIMING build of assemble3.hip selected with FEDM_HIP_LIB and FEDM_LEAN3_PERSISTENT=0.
usage: phase_time3.py [mesh | -k]   (as tools/kernel_ab.py)"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ["FEDM_LEAN3_PERSISTENT"] = "0"
from fedm_amd.cases import streamer
from fedm_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
if n < 0:
    h = -n * 1e-6
    msh = streamer.refined_mesh(h, growth=0.1, channel=(0.0, 100.0 * h) + streamer.CHANNEL[2:])
else:
    msh = streamer.mesh(n, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
streamer.initialise(prob, multigrid=False)
prob.set_step(5e-12, 5e-12)
lib = C.CDLL(os.environ["FEDM_HIP_LIB"])
out = (C.c_ulonglong * 8)()
prob.jacobian()
lib.fedm_debug_phase3(out, 1)
reps = 10
for _ in range(reps):
    prob.jacobian()
lib.fedm_debug_phase3(out, 1)
npatch = prob.sizes()["n_slices"] if "n_slices" in prob.sizes() else (prob.nv + 63) // 64
names = ["header, cell record request, zeroing", "staging: halo id -> vertex data -> exp -> LDS", "barrier 1",
         "the cell (all rows, LDS atomics)", "barrier 2", "stream-out (issue)"]
tot = sum(out)
for k, v in zip(names, out):
    print(f"{k:50s} {100.0 * v / tot:5.1f} %   {v / reps / npatch / 100.0:7.2f} us per patch (100 MHz clock)")
print(f"workgroup lifetime (wave 0): {tot / reps / npatch / 100.0:.2f} us;  kernel: {1e3 * prob.time_kernel(0, 20):.1f} us "
      f"for {npatch} workgroups")
