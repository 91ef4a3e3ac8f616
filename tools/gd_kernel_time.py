"""Time of one LMEA F + J assembly (glow discharge, NxN crossed mesh: 200 -> 402k DOFs, 141 -> 200k), back to back.
python tools/gd_kernel_time.py [N]     (FEDM_GD_HAND=0..5 selects the variant, FEDM_HIP_LIB an experiment build)"""
import sys, io, contextlib, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd.cases import glow_discharge as gdc
with contextlib.redirect_stdout(io.StringIO()):
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    case = gdc.Case(nx=n, ny=n, T_final=1.0)
for _ in range(2):
    case.step()
prob = case.prob
t = min(prob.time_kernel(0, 10) for _ in range(3))
tr = min(prob.time_kernel(2, 10) for _ in range(3))
print(os.environ.get("FEDM_GD_HAND", "default"), os.path.basename(os.environ.get("FEDM_HIP_LIB", "default")),
      f"{n}x{n}: F+J assembly {1e3 * t:.1f} us, residual only {1e3 * tr:.1f} us", flush=True)
