"""A few time steps of the glow-discharge case of bench.py (BASELINE configs[2], 141 x 141 crossed mesh, device
pipeline) -- the target of the glow-discharge PMC passes of tools/collect_profiles.sh.  usage: gd_steps.py [steps]"""
import contextlib, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fedm_amd.cases import glow_discharge as gdc
with contextlib.redirect_stdout(io.StringIO()):
    case = gdc.Case(nx=141, ny=141, T_final=1.0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    case.step()
print("steps", case.newton_iterations, case.linear_iterations)
case.prob.close()
