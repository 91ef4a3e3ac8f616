"""Regenerate the committed golden fixtures from the reference checkout.

Run in the development container only (``/root/reference`` does not exist on
the GPU box):  ``python tests/golden/make_fixtures.py``

Fixtures are DATA: numbers parsed from the reference's own golden result files
and numbers produced by calling the reference's dolfin-free Python functions.
``import fedm`` needs ``dolfin`` (fedm/functions.py:8), which is not installed;
a throw-away stand-in module is created in a temporary directory (never in the
repo) only so that the import succeeds -- none of the functions exercised here
touch it except for ``dolfin.exp`` (mapped to ``math.exp``) and ``DOLFIN_EPS``.
"""
import json
import sys
import tempfile
import textwrap
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
IT = REF / "tests" / "integrated_tests"


def tof_golden():
    root = ET.parse(IT / "time_of_flight/20220707_results/electrons000000.vtu").getroot()
    arrs = {a.attrib.get("Name", "points"): a for a in root.findall(".//DataArray")}
    pts = np.array(arrs["points"].text.split(), dtype=np.float64).reshape(-1, 3)[:, :2]
    conn = np.array(arrs["connectivity"].text.split(), dtype=np.int32).reshape(-1, 3)
    field = np.array(arrs["f_3199"].text.split(), dtype=np.float64)
    line = (IT / "time_of_flight/20220707_results/relative error.log").read_text().splitlines()[0]
    parts = dict(p.strip().split(" = ") for p in line.split("\t"))
    np.savez_compressed(OUT / "tof_golden.npz", coords=pts, cells=conn, n_e=field,
                        h_max=float(parts["h_max"]), dt=float(parts["dt"]),
                        relative_error=float(parts["relative_error"]))


def error_logs():
    logs = {}
    for case in ("streamer_discharge", "glow_discharge"):
        rows = np.loadtxt(IT / case / "20220707_results" / "relative error.log")
        logs[case] = rows.tolist()
    (OUT / "error_logs.json").write_text(json.dumps(logs, indent=1))


def gd_golden():
    """Glow-discharge checkpoint goldens (XDMF/HDF5) -> per-vertex ln n, vertex order."""
    sys.path.insert(0, str(OUT))
    from h5read import H5
    h5 = H5()
    res = IT / "glow_discharge/20220707_results"
    out = {}
    for key in ("electrons", "Ar_plus", "Ar_star"):
        f = res / f"{key}.h5"
        for snap in (0, 1):
            g = f"/{key}/{key}_{snap}"
            vec = h5.read(f, g + "/vector")[:, 0]
            topo = h5.read(f, g + "/mesh/topology", np.int64)
            cd = h5.read(f, g + "/cell_dofs", np.int64)[:, 0].reshape(-1, 3)
            dof_of_vertex = np.empty(vec.size, dtype=np.int64)
            dof_of_vertex[topo.ravel()] = cd.ravel()
            out[f"{key}_{snap}"] = vec[dof_of_vertex]
        out["coords"] = h5.read(f, f"/{key}/{key}_1/mesh/geometry")
        out["cells"] = topo.astype(np.int32)
    np.savez_compressed(OUT / "gd_golden.npz", **out)


def import_reference():
    stub = tempfile.mkdtemp(prefix="dolfin_stub_")
    Path(stub, "dolfin.py").write_text(textwrap.dedent("""
        import math
        pi = math.pi
        DOLFIN_EPS = 3.0e-16
        def exp(x):
            return x.exp() if hasattr(x, "exp") else math.exp(x)
        def Constant(x):
            return x
        def set_log_level(level):
            pass
        def between(x, rng):
            return rng[0] <= x <= rng[1]
        class _Comm:
            comm_world = 0
            @staticmethod
            def rank(comm):
                return 0
        MPI = _Comm
        class SubDomain:
            pass
        class NonlinearProblem:
            pass
        def __getattr__(name):
            return type(name, (), {})
    """))
    sys.path.insert(0, stub)
    sys.path.insert(0, str(REF))
    import fedm.functions as ff
    import fedm.file_io as fio
    return ff, fio


def controller_and_sources():
    ff, fio = import_reference()
    out = {}
    # step controllers, fedm/functions.py:915-951
    args = (1e-12, [5e-4, 6e-4, 7e-4], 1e-3, 1e-15, 5e-12)
    out["adaptive_timestep"] = dict(args=args, value=ff.adaptive_timestep(*args))
    out["adaptive_timestep_PI34"] = dict(args=args, value=ff.adaptive_timestep_PI34(*args))
    h_args = (1e-12, 2e-12, [5e-4, 6e-4, 7e-4], 1e-3, 1e-15, 5e-12)
    out["adaptive_timestep_H211b"] = dict(args=h_args, value=ff.adaptive_timestep_H211b(*h_args))
    # modify_approximation_vars, fedm/functions.py:15-45
    out["modify_LFA"] = ff.modify_approximation_vars("LFA", 3, ["a", "b", "c"], [1., 2., 3.], [0., 1., -1.])
    out["modify_LMEA"] = ff.modify_approximation_vars("LMEA", 3, ["a", "b", "c"], [1., 2., 3.], [0., 1., -1.])
    # glow-discharge deck through the reference readers, fedm/file_io.py:250-521
    fio.files.file_input = IT / "glow_discharge" / "file_input"
    model = "4_particles"
    path = fio.files.file_input / model
    nsp, names, props, tc_names = fio.read_speclist(path)
    M, Z = fio.read_particle_properties(props, model)
    P, L, G = fio.reaction_matrices(path, names)
    kfiles = fio.rate_coefficient_file_names(path)
    loss = fio.read_energy_loss(path)
    kdep = fio.read_dependences(kfiles)
    kx, ky = fio.read_rate_coefficients(kfiles, kdep)
    Dx, Dy, Ddep = fio.read_transport_coefficients(tc_names, "Diffusion", model)
    mx, my, mdep = fio.read_transport_coefficients(tc_names, "mobility", model)
    summ = lambda v: [float(np.sum(a)) if np.ndim(a) else float(a) for a in v]
    out["gd_deck"] = dict(
        n=nsp, names=names, props=props, tc_names=tc_names, M=M, Z=Z,
        power=P.tolist(), loss_m=L.tolist(), gain_m=G.tolist(),
        kfiles=[f.name for f in kfiles], energy_loss=loss, kdep=kdep,
        k_len=[int(np.size(a)) for a in kx], kx_sum=summ(kx), ky_sum=summ(ky),
        Ddep=Ddep, D_len=[int(np.size(a)) for a in Dx], Dy_sum=summ(Dy),
        mdep=mdep, m_len=[int(np.size(a)) for a in mx], my_sum=summ(my))
    # streamer deck
    fio.files.file_input = IT / "streamer_discharge" / "file_input"
    model = "benchmark_model"
    path = fio.files.file_input / model
    nsp, names, props, tc_names = fio.read_speclist(path)
    M, Z = fio.read_particle_properties(props, model)
    Dx, Dy, Ddep = fio.read_transport_coefficients(names, "Diffusion", model)
    mx, my, mdep = fio.read_transport_coefficients(names, "mobility", model)
    out["streamer_deck"] = dict(n=nsp, names=names, props=props, tc_names=tc_names, M=M, Z=Z,
                                Dy=Dy, Ddep=Ddep, my=my, mdep=mdep)
    # Source_term / Energy_Source_term spot values, fedm/functions.py:777-912
    u = [float(np.log(3e12)), float(np.log(1e12)), float(np.log(1e12)), float(np.log(1e12)), 0.0]
    k = np.array([1e-16, 2e-16, 3e-16, 4e-16, 0.81e-15, 1e6, 5e-14])
    N0 = 3.21877e22
    f = ff.Source_term("coupled", "LMEA", P, L, G, k, N0, u)
    fen = ff.Energy_Source_term("coupled", P, L, G, k, loss, 5.0, N0, u)
    out["source_term"] = dict(u=u, k=k.tolist(), N0=N0, f=[float(v) for v in f],
                              f_energy=float(fen), mean_energy=5.0)
    # ... and with the decks' two sentinel losses (fedm/functions.py:906-909): reaction 1 loses Ei - mean energy,
    # reaction 6 the mean energy itself
    loss_s = list(loss)
    loss_s[1], loss_s[6] = 7.77e77, 9.99e99
    fen_s = ff.Energy_Source_term("coupled", P, L, G, k, loss_s, 5.0, N0, u, Ei=15.76)
    out["source_term"].update(energy_loss_sentinels=loss_s, Ei=15.76, f_energy_sentinels=float(fen_s))
    (OUT / "reference_values.json").write_text(json.dumps(out, indent=1, default=str))


def h5_layout():
    """Datasets of one snapshot group of DOLFIN's XDMF checkpoint (rank, i = integer / f = float)
    as found in the glow-discharge golden -- what fedm_amd.mesh_io.XDMFFile has to write."""
    sys.path.insert(0, str(OUT.parent.parent))
    from fedm_amd import h5
    f = IT / "glow_discharge/20220707_results/electrons.h5"
    layout = {}
    with h5.File(f) as hf:
        assert hf.keys("/electrons")[:2] == ["electrons_0", "electrons_1"]
        for ds, kind in (("vector", "f"), ("cell_dofs", "i"), ("x_cell_dofs", "i"), ("cells", "i"),
                         ("mesh/geometry", "f"), ("mesh/topology", "i")):
            arr = hf.read("/electrons/electrons_0/" + ds, np.int64 if kind == "i" else np.float64)
            layout[ds] = [arr.ndim, kind]
        assert sorted(hf.keys("/electrons/electrons_0")) == ["cell_dofs", "cells", "mesh", "vector", "x_cell_dofs"]
    path = OUT / "reference_values.json"
    out = json.loads(path.read_text())
    out["h5_checkpoint_layout"] = layout
    path.write_text(json.dumps(out, indent=1, default=str))


if __name__ == "__main__":
    if "--h5-layout-only" in sys.argv:
        h5_layout()
        sys.exit(0)
    tof_golden()
    gd_golden()
    error_logs()
    controller_and_sources()
    h5_layout()
    print("fixtures written to", OUT)
