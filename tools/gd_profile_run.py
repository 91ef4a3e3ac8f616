"""Glow discharge, 200x200 crossed mesh (402k DOFs): 20 steps for a rocprofv3 kernel summary."""
import sys, io, contextlib
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd.cases import glow_discharge as gdc
with contextlib.redirect_stdout(io.StringIO()):
    case = gdc.Case(nx=200, ny=200, T_final=1.0)
for _ in range(20):
    case.step()
print("steps", 20, "newton", case.newton_iterations, "gmres", case.linear_iterations)
