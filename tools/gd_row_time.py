"""Time the waves of the LMEA element kernel (gd_jacobian_rows_kernel) spend on each equation row: needs a
-DFEDM_GD_ROW_TIMING build selected with FEDM_HIP_LIB.   python tools/gd_row_time.py [N=141]"""
import ctypes as C, io, contextlib, os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd.cases import glow_discharge as gdc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 141
with contextlib.redirect_stdout(io.StringIO()):
    case = gdc.Case(nx=n, ny=n, T_final=1.0)
for _ in range(2):
    case.step()
prob = case.prob
lib = C.CDLL(os.environ["FEDM_HIP_LIB"])
out = (C.c_ulonglong * 8)()
prob.time_kernel(0, 2)
lib.fedm_debug_gd_rows(out, 1)
reps = 10
t = prob.time_kernel(0, reps)
lib.fedm_debug_gd_rows(out, 1)
n_wg = (case.mesh.cells.shape[0] + 63) // 64
names = ["energy row (wave 0)", "Ar* row (wave 1)", "Ar+ row (wave 2)", "electron row (wave 3)", "Poisson row (wave 1, second pass)",
         "set-up in front of the rows (every wave)"]
for k, name in enumerate(names):
    per = out[k] / (reps + 0) / n_wg / 100.0 / (4 if k == 5 else 1)
    print(f"{name:45s} {per:7.2f} us per workgroup")
print(f"F + J assembly {1e3 * t:.1f} us, {n_wg} workgroups of 4 waves, one per CU at a time")
