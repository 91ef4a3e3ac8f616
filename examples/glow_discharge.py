#!/usr/bin/env python3
"""Low-pressure argon glow discharge (LMEA, 4 particles + electron energy + Poisson) -- the
reference's examples/glow_discharge/fedm-gd.py on the MI355X device path, with its outputs:
``relative error.log`` and one XDMF/HDF5 checkpoint file per species in DOLFIN's layout
(fedm-gd.py:300-309, 420-460; fedm/file_io.py:538-616).

The deck is read with the reference's own readers (`fedm_amd.file_io`, same file formats);
the per-step coefficient pipeline (reduced field projection, table look-ups, Einstein relation,
mean energy) and the Newton solves run on the device (`fedm_amd.cases.glow_discharge.Case`).
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd import mesh_io
from fedm_amd.cases import glow_discharge as gdc
from fedm_amd.mesh import Mesh


def main(nx=100, ny=100, T_final=1e-11, output_dir=".", t_output_step=1e-11, quiet=True):
    out = Path(output_dir)
    out.mkdir(parents=True, exist_ok=True)
    case = gdc.Case(nx=nx, ny=ny, T_final=T_final, error_file=out / "relative error.log", quiet=quiet)
    mesh = Mesh(case.prob.coords, case.prob.cells)
    # one checkpoint file per particle species, components 1.. of the mixed state (0 = energy)
    names = ["Ar_1p0", "Ar_star", "Ar_plus", "electrons"][-(case.ns - 1):]
    files = [mesh_io.XDMFFile(out / f"{n}.xdmf", mesh) for n in names]
    comps = list(range(1, case.ns))
    U = case.prob.get_state()
    for f, n, cidx in zip(files, names, comps):                    # initial condition, snapshot _0
        f.write_checkpoint(U[:, cidx], n, 0.0, None, False)
    t_out, step = t_output_step, t_output_step
    while case.t < T_final * (1.0 - 1e-12):
        t_old = case.t
        case.step()
        if t_out <= case.t:
            U_new, U_old = case.prob.get_state(), case.prob.get_state_old()
            t_out, step = mesh_io.file_output(case.t, t_old, t_out, step, [T_final, 2 * T_final], [step, step],
                                              ["xdmf"] * len(files), files, names,
                                              [U_new[:, c] for c in comps], [U_old[:, c] for c in comps])
    return dict(t=case.t, steps=len(open(out / "relative error.log").readlines()), output=str(out),
                species=names)


if __name__ == "__main__":
    res = main(output_dir=sys.argv[1] if len(sys.argv) > 1 else "gd_output", quiet=False)
    print(res)
