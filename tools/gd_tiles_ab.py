"""Glow-discharge steps with the species sweeps on tiles (four unknowns a vertex) and one launch each (FEDM_FS_TILES=0):
python tools/gd_tiles_ab.py [N=141] [steps=60]"""
import contextlib, io, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if "--one" in sys.argv:
    sys.path.insert(0, str(ROOT))
    from fedm_amd.cases import glow_discharge as gdc
    n, steps = int(sys.argv[2]), int(sys.argv[3])
    with contextlib.redirect_stdout(io.StringIO()):
        case = gdc.Case(nx=n, ny=n, T_final=1.0)
    for _ in range(10):
        case.step()
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    l0 = case.linear_iterations
    for _ in range(steps):
        case.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"FEDM_FS_TILES={os.environ.get('FEDM_FS_TILES', 'default')}: {steps / dt:.1f} steps/s, {(case.linear_iterations - l0) / steps:.1f} GMRES/step, "
          f"tiles {case.prob.fieldsplit_tiles()}", flush=True)
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "141"
    steps = sys.argv[2] if len(sys.argv) > 2 else "60"
    for rep in range(2):
        for tiles in ("1", "0"):
            subprocess.run([sys.executable, __file__, "--one", n, steps], env=dict(os.environ, FEDM_FS_TILES=tiles))
