"""Phase breakdown of the patch assembly kernel (needs a -DFEDM_PHASE_TIMING build: FEDM_HIP_LIB)."""
import ctypes as C, sys
sys.path.insert(0, '.')
from fedm_amd.cases import streamer
from fedm_amd import _lib
msh = streamer.mesh(576, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
streamer.initialise(prob, multigrid=False)
prob.set_step(5e-12, 5e-12)
lib = _lib.load()
out = (C.c_ulonglong * 8)()
lib.fedm_debug_phase(out, 1)
n = 10
for _ in range(n): prob.jacobian()
lib.fedm_debug_phase(out, 1)
# the row-at-a-time kernel (assemble_lean_kernel); the unrolled one (FEDM_ASSEMBLY_LEAN=0) has the
# phases zero / stage / barrier / cell record / setup / rows / barrier / stream out
import os
if os.environ.get("FEDM_ASSEMBLY_LEAN", "1") == "0":
    names = ["zero LDS", "stage vertices (global loads)", "barrier 1", "cell record + LDS reads", "setup",
             "rows: moments + emission + atomics", "barrier 2", "stream out"]
else:
    names = ["zero + stage vertices (issue) [+ vertex exponentials, lean2]", "barrier: staged loads arrive",
             "prologue: cell record, exp(u) [lean2: field, rate coefficient]",
             "rows (3x): set-up, moments, emission", "barriers after the rows", "stream-out + zeroing (3x, issue)",
             "LDS-only barriers", "F stream-out"]
tot = sum(out)
for k, v in zip(names, out):
    print(f"{k:62s} {100.0 * v / tot:5.1f} %   {v / n / 5203 / 100.0:7.2f} us per patch (100 MHz clock)")
print(f"workgroup lifetime (wave 0): {tot / n / 5203 / 100.0:.2f} us;  kernel: {1e3 * prob.time_kernel(0, 20):.1f} us "
      f"for 5203 workgroups")
