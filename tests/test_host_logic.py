"""CPU tests of the host layer: step controllers (bit-exact replay of the reference's golden
error logs), deck readers and source terms against fixtures generated from the reference's own
Python (tests/golden/make_fixtures.py), coefficient parser, façade error behaviour, and that the
C-ABI library loads and exports every symbol include/fedm_hip.h declares."""
import json
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def ref(golden_dir):
    return json.loads((golden_dir / "reference_values.json").read_text())


@pytest.fixture(scope="module")
def logs(golden_dir):
    return json.loads((golden_dir / "error_logs.json").read_text())


# ---- step controllers ---------------------------------------------------------------------
@pytest.mark.parametrize("impl", ["product", "oracle"])
@pytest.mark.parametrize("case,ttol,dt_max", [("streamer_discharge", 1e-3, 5e-12),
                                              ("glow_discharge", 2e-3, 1e-11)])
def test_adaptive_timestep_replays_golden_logs(logs, impl, case, ttol, dt_max):
    """Row k of `relative error.log` predicts dt of row k+1 (fedm-streamer.py:335-340)."""
    if impl == "product":
        from fedm_amd.functions import adaptive_timestep
    else:
        from oracle.controller import adaptive_timestep
    rows = logs[case]
    max_error = [1, 1, 1]
    dt_min = 1e-15
    for k, (err, dt_old, dt) in enumerate(rows[:-1]):
        max_error[0] = err
        nxt = adaptive_timestep(dt, max_error, ttol, dt_min, dt_max)
        max_error[2], max_error[1] = max_error[1], max_error[0]
        assert nxt == rows[k + 1][2], (k, nxt, rows[k + 1][2])      # bit-exact
        assert rows[k + 1][1] == dt


def test_controller_known_answers(ref):
    from fedm_amd import functions as ff
    for name in ("adaptive_timestep", "adaptive_timestep_PI34", "adaptive_timestep_H211b"):
        assert getattr(ff, name)(*ref[name]["args"]) == ref[name]["value"]


# ---- fedm.functions façade: the reference's own unit tests ---------------------------------
def test_modify_approximation_vars(ref):
    """tests/unit_tests/functions/test_modify_approximation_vars.py:6-54"""
    from fedm_amd.functions import modify_approximation_vars
    out = modify_approximation_vars("LFA", 3, ["a", "b", "c"], [1., 2., 3.], [0., 1., -1.])
    assert list(out) == ref["modify_LFA"]
    out = modify_approximation_vars("LMEA", 3, ["a", "b", "c"], [1., 2., 3.], [0., 1., -1.])
    assert list(out) == ref["modify_LMEA"]
    with pytest.raises(ValueError) as exc:
        modify_approximation_vars("bad_type", 3, ["a"], [1.], [0.])
    assert "bad_type" in str(exc.value)


def test_facade_error_messages():
    from fedm_amd import functions as ff
    from fedm_amd.mesh import Marking_boundaries, RectangleMesh
    with pytest.raises(ValueError, match="status 'later' not recognised"):
        ff.Transport_coefficient_interpolation("later", [], 1.0, 300.0, [], [], [], None, None)
    with pytest.raises(ValueError, match="dependence 'fun' not"):
        ff.Rate_coefficient_interpolation("initial", ["fun"], [None], [0], [0], None, None)
    with pytest.raises(ValueError, match="coupling must be"):
        ff.Source_term("x", "LFA", [], [], [], [], 1.0, [])
    with pytest.raises(ValueError, match="Invalid boundary_type 'arc'"):
        Marking_boundaries(RectangleMesh((0, 0), (1, 1), 2, 2), [["arc", 0, 0, 0, 1]])
    with pytest.raises(ValueError, match="Invalid function_type"):
        ff.Function_definition(None, "Nope")


# ---- deck readers ----------------------------------------------------------------------------
def test_glow_discharge_deck(ref):
    from fedm_amd import file_io as fio
    g = ref["gd_deck"]
    fio.files.file_input = ROOT / "decks" / "glow_discharge" / "file_input"
    model = "4_particles"
    path = fio.files.file_input / model
    n, names, props, tc = fio.read_speclist(path)
    assert (n, names, props, tc) == (g["n"], g["names"], g["props"], g["tc_names"])
    M, Z = fio.read_particle_properties(props, model)
    assert M == g["M"] and Z == g["Z"]
    P, L, G = fio.reaction_matrices(path, names)
    assert P.tolist() == g["power"] and L.tolist() == g["loss_m"] and G.tolist() == g["gain_m"]
    kfiles = fio.rate_coefficient_file_names(path)
    assert [f.name for f in kfiles] == g["kfiles"]
    assert fio.read_energy_loss(path) == g["energy_loss"]
    kdep = fio.read_dependences(kfiles)
    assert kdep == g["kdep"]
    kx, ky = fio.read_rate_coefficients(kfiles, kdep)
    assert [int(np.size(a)) for a in kx] == g["k_len"]
    summ = lambda v: [float(np.sum(a)) if np.ndim(a) else float(a) for a in v]
    assert np.allclose(summ(kx), g["kx_sum"], rtol=1e-14) and np.allclose(summ(ky), g["ky_sum"], rtol=1e-14)
    Dx, Dy, Ddep = fio.read_transport_coefficients(tc, "Diffusion", model)
    mx, my, mdep = fio.read_transport_coefficients(tc, "mobility", model)
    norm = lambda d: [str(x) for x in d]
    assert norm(Ddep) == norm(g["Ddep"]) and norm(mdep) == norm(g["mdep"])
    assert [int(np.size(a)) for a in Dx] == g["D_len"] and [int(np.size(a)) for a in mx] == g["m_len"]
    assert np.allclose(summ(Dy), g["Dy_sum"], rtol=1e-14) and np.allclose(summ(my), g["my_sum"], rtol=1e-14)


def test_streamer_deck_and_model(ref):
    from fedm_amd import file_io as fio
    from fedm_amd.cases import streamer
    s = ref["streamer_deck"]
    fio.files.file_input = ROOT / "decks" / "streamer_discharge" / "file_input"
    n, names, props, tc = fio.read_speclist(fio.files.file_input / "benchmark_model")
    assert (n, names, props, tc) == (s["n"], s["names"], s["props"], s["tc_names"])
    M, Z = fio.read_particle_properties(props, "benchmark_model")
    assert M == s["M"] and Z == s["Z"]
    _, Dy, Ddep = fio.read_transport_coefficients(names, "Diffusion", "benchmark_model")
    _, my, mdep = fio.read_transport_coefficients(names, "mobility", "benchmark_model")
    assert Dy == s["Dy"] and Ddep == s["Ddep"] and my == s["my"] and mdep == s["mdep"]
    a, b = streamer.model_from_deck(), streamer.model()
    assert a.n_species == b.n_species == 2 and list(a.Z) == list(b.Z)
    for E in (1e5, 1.5e6, 2e7):
        for x, y in zip(list(a.mu) + list(a.D) + [a.reactions[0].k],
                        list(b.mu) + list(b.D) + [b.reactions[0].k]):
            assert x(E) == pytest.approx(y(E), rel=1e-15)
    with pytest.raises(RuntimeError, match="is not a directory"):
        fio.files.file_input = ROOT / "no_such_dir"


def test_missing_files_raise_like_the_reference(tmp_path):
    from fedm_amd import file_io as fio
    with pytest.raises(FileNotFoundError, match="fedm.read_dependence"):
        fio.read_dependence(tmp_path / "nope.dat")
    assert fio.read_dependences([tmp_path / "nope.dat"], zero_if_file_missing=True) == [0]
    with pytest.raises(ValueError, match="should be the same length"):
        fio.read_rate_coefficients(["a"], [])
    with pytest.raises(ValueError, match="dependence 'weird' is not"):
        fio.read_rate_coefficients(["a"], ["weird"])


# ---- source terms -----------------------------------------------------------------------------
def test_source_terms_match_reference_values(ref):
    from fedm_amd import functions as ff
    g, s = ref["gd_deck"], ref["source_term"]
    P, L, G = np.array(g["power"]), np.array(g["loss_m"]), np.array(g["gain_m"])
    f = ff.Source_term("coupled", "LMEA", P, L, G, s["k"], s["N0"], s["u"])
    assert np.allclose(f, s["f"], rtol=1e-14)
    fen = ff.Energy_Source_term("coupled", P, L, G, s["k"], g["energy_loss"], s["mean_energy"],
                                s["N0"], s["u"])
    assert fen == pytest.approx(s["f_energy"], rel=1e-14)
    # the decks' sentinel losses (fedm/functions.py:906-909): the facade and the oracle's statement against the value
    # the reference itself returned (tests/golden/make_fixtures.py)
    fen_s = ff.Energy_Source_term("coupled", P, L, G, s["k"], s["energy_loss_sentinels"], s["mean_energy"],
                                  s["N0"], s["u"], Ei=s["Ei"])
    assert fen_s == pytest.approx(s["f_energy_sentinels"], rel=1e-14)
    from oracle.gd import energy_loss_factor
    expn = [s["N0"]] + [float(np.exp(v)) for v in s["u"][1:4]]
    rates = [s["k"][j] * np.prod([expn[i] ** P[j, i] for i in range(4)]) for j in range(7)]
    mine = sum(-rates[j] * energy_loss_factor(s["energy_loss_sentinels"][j], s["Ei"], s["mean_energy"]) for j in range(7))
    assert mine == pytest.approx(s["f_energy_sentinels"], rel=1e-13)


def test_interpolation_semantics():
    from fedm_amd import functions as ff
    from fedm_amd.forms import Function
    from fedm_amd.physical_constants import elementary_charge, kB, kB_eV
    N0, Tgas = 2.0, 300.0
    energy, red = Function(values=[0.5, 2.0, 9.0]), Function(values=[10.0, 20.0, 30.0])
    kx, ky = [1.0, 3.0, 5.0], [10.0, 30.0, 20.0]
    ks = [Function(values=np.zeros(3)) for _ in range(4)]
    mu = Function(values=[1.0, 2.0, 3.0])
    ff.Transport_coefficient_interpolation("initial", ["const", "Umean", "E/N", "ESR"], N0, Tgas, ks,
                                           [0, kx, [10.0, 30.0], 0], [8.0, ky, [1.0, 3.0], 0],
                                           energy, red, [None, None, None, mu])
    assert np.array_equal(ks[0].vector(), [4.0] * 3)
    assert np.allclose(ks[1].vector(), np.interp([0.5, 2.0, 9.0], kx, ky) / N0)       # clamped ends
    assert np.allclose(ks[2].vector(), [0.5, 1.0, 1.5])
    assert np.allclose(ks[3].vector(), kB * Tgas * np.array([1.0, 2.0, 3.0]) / elementary_charge)
    r = [Function(values=np.zeros(3))]
    ff.Rate_coefficient_interpolation("update", ["Te"], r, [kx], [ky], energy, red)
    assert np.allclose(r[0].vector(), np.interp(2 * np.array([0.5, 2.0, 9.0]) / (3 * kB_eV), kx, ky))


# ---- coefficient expressions --------------------------------------------------------------------
def test_termsum_parser():
    import math
    from fedm_amd.termsum import TermSum, parse
    a = parse("(1.1944e6 + 4.3666e26 * E_m**(-3))*exp(-2.73e7/E_m)-340.75")
    for E in (3e5, 1.5e6, 4e7):
        assert a(E) == pytest.approx((1.1944e6 + 4.3666e26 * E ** -3) * math.exp(-2.73e7 / E) - 340.75, rel=1e-14)
        h = E * 1e-6
        assert a.derivative(E) == pytest.approx((a(E + h) - a(E - h)) / (2 * h), rel=1e-6)
    assert parse("sqrt(E_m)*2")(4.0) == pytest.approx(4.0)
    assert parse("0.00000E+00").is_const()
    for bad in ("__import__('os').system('true')", "E_m.real", "exp(E_m + E_m**2)", "x*2",
                "(1+E_m)**0.5", "exp(-1/E_m)*exp(-E_m)"):
        with pytest.raises(ValueError):
            parse(bad)


# ---- C ABI -----------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    import __graft_entry__ as entry
    entry.build()
    from fedm_amd import _lib
    header = (ROOT / "include" / "fedm_hip.h").read_text()
    declared = set(re.findall(r"\b(fedm_[a-z0-9_]+)\s*\(", header))
    declared -= {"fedm_allreduce_fn", "fedm_exchange_fn"}
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_lib.exported_symbols())
    # the version the library reports is the header's, and the one the Python binding was written against
    assert lib.fedm_abi_version() == int(re.search(r"#define FEDM_ABI_VERSION (\d+)", header).group(1)) == _lib.ABI_VERSION


def test_binding_refuses_a_library_of_another_abi_version(tmp_path, monkeypatch):
    """A stale .so built against an older header would have its descriptors read past their end: the
    binding compares fedm_abi_version() with its own constant before anything else."""
    from fedm_amd import _lib
    _lib.load()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        _lib.load()


def test_product_and_oracle_quadrature_tables_agree():
    from fedm_amd import quadrature as pq
    from oracle import quadrature as oq
    for d in (1, 2, 3, 4, 5, 6, 8):
        a, b = pq.triangle(d), oq.triangle_rule(d)
        assert np.allclose(a[0], b[0]) and np.allclose(a[1], b[1]) and a[1].sum() == pytest.approx(0.5)
    assert np.allclose(pq.interval(2)[0], oq.interval_rule(2)[0])


# ---- weak-form façade -----------------------------------------------------------------------------
def _streamer_forms(mesh):
    from fedm_amd import forms, functions as ff
    from fedm_amd.forms import (Expression, FiniteElement, Function, FunctionSpace, Measure,
                                MixedElement, TestFunctions, TrialFunction, dx, exp, grad, inner, sqrt)
    from fedm_amd.cases import streamer
    from fedm_amd.physical_constants import elementary_charge, epsilon_0
    from fedm_amd.termsum import parse
    forms.parameters["form_compiler"]["quadrature_degree"] = 2
    ME = FunctionSpace(mesh, MixedElement([FiniteElement()] * 3))
    u, v = TrialFunction(ME), TestFunctions(ME)
    E = -grad(u[2])
    E_m = sqrt(inner(-grad(u[2]), -grad(u[2])))
    mu1, D1 = parse(streamer.MU_E), parse(streamer.D_E)
    alpha = (1.1944e6 + 4.3666e26 * E_m ** (-3)) * exp(-2.73e7 / E_m) - 340.75
    f = [alpha * mu1 * E_m * exp(u[1]), alpha * mu1 * E_m * exp(u[1]),
         Function(FunctionSpace(mesh, FiniteElement()))]
    sign = [1.0, -1.0]
    for i in range(2):
        f[2] += sign[i] * exp(u[i]) * elementary_charge / epsilon_0
    dt = Expression("time_step", time_step=5e-12)
    r = Expression("x[0]", degree=1)
    eqt = ["reaction", "drift-diffusion-reaction"]
    F = 0.0
    F += ff.weak_form_balance_equation_log_representation(eqt[0], dt, dt, dx, u[0], None, None, v[0], f[0], 0.0, r, 0.0)
    F += ff.weak_form_balance_equation_log_representation(
        eqt[1], dt, dt, dx, u[1], None, None, v[1], f[1],
        ff.Flux(sign[1], u[1], D1, mu1, E, grad_diffusion=False), r, D1)
    F += ff.weak_form_Poisson_equation(dx, u[2], v[2], f[2], r)
    dsm = Measure("ds", domain=mesh, subdomain_data=ff.Marking_boundaries(mesh, streamer.BOUNDARIES))
    for i in range(4):
        for j in range(2):
            F += ff.Boundary_flux(streamer.BC_TYPE[i][j], eqt[j], "x", sign[j], None, E, None, u[j],
                                  0.0, v[j], dsm(i + 1), r)
    return F


def test_weak_forms_compile_to_the_streamer_model():
    """The calls of fedm-streamer.py:235-271 produce the same device model as the case module."""
    from fedm_amd import functions as ff
    from fedm_amd.cases import streamer
    from fedm_amd.mesh import RectangleMesh
    mesh = RectangleMesh((0, 0), (0.0125, 0.0125), 4, 4)
    m, mesh2, tags = ff.compile_forms(_streamer_forms(mesh))
    ref = streamer.model()
    assert mesh2 is mesh and tags.shape == (mesh.num_cells(), 3)
    assert (m.n_species, m.poisson, list(m.eq_type), list(m.Z)) == (2, True, list(ref.eq_type), [1.0, -1.0])
    assert m.bc_kind == ref.bc_kind and m.quadrature_degree == 2 and m.axisymmetric
    assert len(m.reactions) == 1 and list(m.reactions[0].net) == [1, 1] and list(m.reactions[0].power) == [0, 1]
    for E in (1e5, 2e6, 3e7):
        assert m.reactions[0].k(E) == pytest.approx(ref.reactions[0].k(E), rel=1e-14)
        assert m.mu[1](E) == pytest.approx(ref.mu[1](E), rel=1e-15)
        assert m.D[1](E) == pytest.approx(ref.D[1](E), rel=1e-15)
    a, b = m.to_c(), ref.to_c()
    assert bytes(a)[:64] == bytes(b)[:64] and a.n_qp == b.n_qp == 3


def test_weak_form_argument_checks():
    """Same ValueErrors / warning as fedm/functions.py:333-348, 477-512."""
    from fedm_amd import functions as ff
    with pytest.raises(ValueError, match="is not recognised"):
        ff.weak_form_balance_equation_log_representation("advection", 0, 0, 0, 0, 0, 0, 0, 0, 0)
    with pytest.raises(ValueError, match="must also supply the diffusion coefficient"):
        ff.weak_form_balance_equation_log_representation("diffusion-reaction", 0, 0, 0, 0, 0, 0, 0, 0, 0)
    with pytest.raises(ValueError, match="boundary condition type 'wall' not recognised"):
        ff.Boundary_flux("wall", "reaction", "Heavy", 1, 0, 0, 0, 0, 0, 0, None)
    with pytest.raises(ValueError, match="equation type 'advection' not recognised"):
        ff.Boundary_flux("Neumann", "advection", "Heavy", 1, 0, 0, 0, 0, 0, 0, None)
    with pytest.warns(UserWarning, match="should have spaces"):
        assert ff.Boundary_flux("zero_flux", "reaction", "Heavy", 1, 0, 0, 0, 0, 0, 0, None) == 0.0
    assert ff.Boundary_flux("Neumann", "reaction", "Heavy", 1, 0, 0, 0, 0, 0, 0, None) == 0.0
    assert ff.Max(3.0, 5.0) == 5.0 and ff.Min(3.0, 5.0) == 3.0


# ---- mesh ingestion / result output ---------------------------------------------------------------
def test_dolfin_xml_and_vtu_round_trip(tmp_path, golden_dir):
    from fedm_amd import mesh_io
    from fedm_amd.mesh import RectangleMesh
    m = RectangleMesh((0.0, 0.0), (2.5e-4, 5e-4), 40, 40)
    mesh_io.write_dolfin_xml(m, tmp_path / "mesh.xml")
    m2 = mesh_io.read_dolfin_xml(tmp_path / "mesh.xml")
    assert np.array_equal(m2.cells, m.cells) and np.array_equal(m2.coords, m.coords)
    gold = np.load(golden_dir / "tof_golden.npz")
    f = mesh_io.PVDFile(tmp_path / "number density" / "electrons" / "electrons.pvd", m)
    vtu = f.write(gold["n_e"], "f_52", 2.6e-9)
    assert vtu.name == "electrons000000.vtu"
    back = mesh_io.read_vtu(vtu, "f_52")
    assert np.array_equal(back, gold["n_e"])            # repr() round-trips doubles exactly
    assert "electrons000000.vtu" in (tmp_path / "number density" / "electrons" / "electrons.pvd").read_text()


def test_file_output_interpolates_like_the_reference(tmp_path):
    from fedm_amd import mesh_io
    from fedm_amd.mesh import RectangleMesh
    m = RectangleMesh((0, 0), (1, 1), 2, 2)
    f = mesh_io.PVDFile(tmp_path / "a.pvd", m)
    new, old = np.full(9, 3.0), np.full(9, 1.0)
    t_out, step = mesh_io.file_output(1.6e-11, 0.6e-11, 1e-11, 1e-11, [1e-11, 1e-10], [1e-11, 1e-10],
                                      ["pvd"], [f], ["x"], [new], [old], unit="us")
    assert len(f.entries) == 1 and f.entries[0][0] == pytest.approx(1e-5)
    assert np.allclose(mesh_io.read_vtu(tmp_path / "a000000.vtu", "x"), 1.0 + 0.4 * 2.0)
    assert t_out == pytest.approx(2e-11) and step == 1e-11
    with pytest.raises(ValueError, match="unit 'h' not valid"):
        mesh_io.file_output(1, 0, 1, 1, [1], [1], ["pvd"], [f], ["x"], [new], [old], unit="h")


def test_xdmf_checkpoint_layout_and_roundtrip(tmp_path, golden_dir):
    """XDMFFile.write_checkpoint writes DOLFIN's checkpoint layout (the datasets, ranks and
    dtypes recorded from the reference's glow-discharge goldens) and file_output drives it like
    fedm/file_io.py:597-604; read_h5 / read_checkpoint read it back."""
    import json
    from fedm_amd import h5, mesh_io
    from fedm_amd.mesh import RectangleMesh
    mesh = RectangleMesh((0.0, 0.0), (1.0, 2.0), 4, 6, "crossed")
    f = mesh_io.XDMFFile(tmp_path / "electrons.xdmf", mesh)
    u0 = np.sin(mesh.coords[:, 0]) + mesh.coords[:, 1]
    u1 = u0 + 1.0
    # two output times inside one step, as file_output interpolates them
    t_out, step = mesh_io.file_output(1.0, 0.0, 0.25, 0.5, [0.0, 10.0], [0.5, 0.5], ["xdmf"], [f],
                                      ["electrons"], [u1], [u0], unit="ns")
    assert t_out == pytest.approx(1.25) and step == 0.5
    vecs = mesh_io.read_h5(tmp_path / "electrons.h5", "electrons")
    assert len(vecs) == 2 and vecs[0].shape == (mesh.num_vertices(), 1)
    assert np.allclose(vecs[0][:, 0], u0 + 0.25) and np.allclose(vecs[1][:, 0], u0 + 0.75)
    coords, cells, fields = mesh_io.read_checkpoint(tmp_path / "electrons.h5", "electrons")
    assert np.array_equal(coords, mesh.coords) and np.array_equal(cells, mesh.cells)
    assert np.allclose(fields[1], u0 + 0.75)
    layout = json.loads((golden_dir / "reference_values.json").read_text())["h5_checkpoint_layout"]
    with h5.File(tmp_path / "electrons.h5") as hf:
        assert hf.keys("/electrons") == ["electrons_0", "electrons_1"]
        for ds, (rank, kind) in layout.items():
            arr = hf.read("/electrons/electrons_0/" + ds, np.int64 if kind == "i" else np.float64)
            assert arr.ndim == rank, ds
    xdmf = (tmp_path / "electrons.xdmf").read_text()
    assert "electrons.h5:/electrons/electrons_1/vector" in xdmf and 'Value="750000000.0"' in xdmf
    # append=False starts over
    f.write_checkpoint(u0, "electrons", 0.0, None, False)
    assert len(mesh_io.read_h5(tmp_path / "electrons.h5", "electrons")) == 1


def test_missing_library_fails_loudly(tmp_path):
    """No CPU fallback: without the HIP library the product refuses to load."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from fedm_amd import _lib\n"
            "try:\n    _lib.load()\nexcept (RuntimeError, OSError) as e:\n    print('REFUSED', e)\n") % str(root)
    env = dict(os.environ, FEDM_HIP_LIB=str(tmp_path / "libfedm_hip_missing.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert "REFUSED" in out.stdout and "not found" in out.stdout


# ---- C-ABI host logic that needs no GPU ----------------------------------------------------
def test_rccl_failure_is_latched_and_reported():
    """Every RCCL return code is checked (csrc/comm.hip): a stub transport whose k-th call fails
    drives the exchange / all-reduce code of the RCCL path.  Whatever call fails, the failure is
    latched with the call's name in fedm_last_error, nothing but the closing ncclGroupEnd of an
    open group is issued afterwards, and without a fault all ten calls of two Krylov-step rounds
    (group start, send, recv, group end, all-reduce) go through."""
    import ctypes as C
    from fedm_amd import _lib
    lib = _lib.load()
    out = (C.c_int64 * 6)()
    assert lib.fedm_debug_comm_fault(-1, out) == 0 and list(out) == [0, 10, 0, 0, 0, 1]   # healthy: destroyed
    names = ["ncclGroupStart", "ncclSend", "ncclRecv", "ncclGroupEnd", "ncclAllReduce"]
    for k in range(10):
        assert lib.fedm_debug_comm_fault(k, out) == 1
        failed, calls, after, reported, aborts, destroys = list(out)
        assert failed == 1 and reported == 1
        assert (aborts, destroys) == (1, 0)         # a failed communicator is aborted, not destroyed (no hang)
        msg = _lib.last_error()
        assert names[k % 5] in msg and "rank 1 of 2" in msg
        inside_group = k % 5 in (1, 2)              # send / recv failed: the group is still closed
        assert after <= (2 if k % 5 == 1 else 1 if inside_group else 0)
        assert calls <= k + 1 + (3 if inside_group else 0)


def test_patch_micro_colouring_removes_lds_bank_clashes(monkeypatch):
    """csrc/prep.cpp orders every assembly patch's cells so that the 16 lanes of an LDS lane group
    add into distinct accumulator banks (the `north_star`'s colouring, at workgroup scale).  On a
    graded structured mesh numbered along the Z-curve the clashing (cell, vertex) pairs drop from
    about half of all pairs to a few percent; the pattern itself is untouched."""
    from fedm_amd.cases import streamer
    from fedm_amd.device import pattern_stats
    msh = streamer.mesh(96, 4.0)
    monkeypatch.setenv("FEDM_PATCH_ORDER", "0")
    plain = pattern_stats(msh.coords, msh.cells)
    monkeypatch.delenv("FEDM_PATCH_ORDER")
    ordered = pattern_stats(msh.coords, msh.cells)
    for k in ("n_slices", "max_patch_cells", "max_patch_width", "cell_visits", "owned_pairs", "nnz_blocks"):
        assert plain[k] == ordered[k]
    assert plain["bank_clashes"] > 0.4 * plain["owned_pairs"]
    assert ordered["bank_clashes"] < 0.06 * ordered["owned_pairs"]


def test_tile_tables_report_the_lds_their_kernels_need():
    """The species-sweep tiles keep a tile's vertices and its layers in LDS; the bytes a workgroup of the
    kernels asks for come with the host-side tile statistics (a context refuses tiles beyond the device's limit
    and runs the sweeps one launch each).  A compact numbering fits the 64 KiB default easily; the SAME mesh with
    a scattered block of vertices at the end of the numbering -- what the ghost layers of a partition were before
    they were given a locality order of their own -- needs several times as much."""
    from fedm_amd.cases import streamer
    from fedm_amd.device import fieldsplit_tiles_stats, locality_order
    msh = streamer.mesh(160, 4.0)
    compact = fieldsplit_tiles_stats(msh.coords, msh.cells)
    assert compact["violations"] == 0
    need = 4 * 2 * compact["max_vertices"] * 2 + 4 * 4 * compact["max_rows"]           # two species, row width 7
    assert compact["lds_bytes_species_kernel"] == need < 64 * 1024
    assert compact["lds_bytes_multigrid_kernel"] == 8 * 2 * compact["max_vertices"] + 4 * 4 * compact["max_rows"]
    # the last 12 % of the vertices in a random order (by global id across an interface, a ghost band looks like that)
    order = locality_order(msh.coords, msh.cells)
    rng = np.random.default_rng(0)
    n_tail = msh.coords.shape[0] // 8
    order[-n_tail:] = order[-n_tail:][rng.permutation(n_tail)]
    inv = np.empty(order.size, dtype=np.int64)
    inv[order] = np.arange(order.size)
    scattered = fieldsplit_tiles_stats(msh.coords[order], inv[msh.cells].astype(np.int32), reorder=False)
    assert scattered["violations"] == 0
    assert scattered["lds_bytes_species_kernel"] > 3 * compact["lds_bytes_species_kernel"]


def test_ghost_layers_of_a_deep_halo_are_compact_patches():
    """Deep halos assemble, sweep and tile the ghost rows like owned rows, so 64 consecutive ghost vertices must be
    a compact patch: each halo link is numbered by the bisection of its own vertices (fedm_amd/partition.py; both
    ends compute the same order).  With the links in ascending global id the tiles of a four-rank split of the
    288 x 288 mesh hold twice the vertices."""
    from fedm_amd import partition
    from fedm_amd.cases import streamer
    from fedm_amd.device import fieldsplit_tiles_stats, locality_order
    g = streamer.mesh(288, 4.0)
    part = partition.partition_rcb(g.coords, 4, g.cells)

    def tiles(lm):
        nv = lm.coords.shape[0]
        order = np.arange(nv)
        own = lm.cells[(lm.cells < lm.n_owned).all(axis=1)]
        order[:lm.n_owned] = locality_order(lm.coords[:lm.n_owned], own)      # the device's numbering: ghosts keep their place
        inv = np.empty(nv, dtype=np.int64)
        inv[order] = np.arange(nv)
        return fieldsplit_tiles_stats(lm.coords[order], inv[lm.cells].astype(np.int32), reorder=False)
    lm = partition.local_mesh(g.coords, g.cells, part, 0, depth=8)
    ordered = tiles(lm)
    # the same local mesh with every link back in ascending global id
    ghosts = np.arange(lm.n_owned, lm.coords.shape[0])
    by_id = np.concatenate([ghosts[lm.recv_ptr[k]:lm.recv_ptr[k + 1]][np.argsort(lm.vertex_global[lm.n_owned + lm.recv_ptr[k]:lm.n_owned + lm.recv_ptr[k + 1]])]
                            for k in range(len(lm.neighbours))])
    perm = np.concatenate([np.arange(lm.n_owned), by_id])
    inv = np.empty(perm.size, dtype=np.int64)
    inv[perm] = np.arange(perm.size)
    import copy
    lm2 = copy.copy(lm)
    lm2.coords, lm2.cells = lm.coords[perm], inv[lm.cells].astype(np.int32)
    plain = tiles(lm2)
    assert ordered["violations"] == 0 and plain["violations"] == 0
    assert ordered["max_vertices"] < 0.7 * plain["max_vertices"]


def test_lmea_script_is_lowered_onto_the_device_model(tmp_path, monkeypatch):
    """examples/glow_discharge.py (our own driver for the case of fedm-gd.py) builds the LMEA form
    with the facade's functions; `Problem()` hands it to fedm_amd.lmea.compile_lmea, which must
    recover the parameters of the device's LMEA model from the script's own objects (and refuse
    what the kernels do not implement).  Runs up to the point where the device context would be
    created.  (The reference's script itself: tests/test_reference_scripts.py.)"""
    import importlib.util
    import fedm_amd.functions as ff
    from fedm_amd import lmea
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("gd_example_cpu", root / "examples" / "glow_discharge.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen = {}

    class Stop(Exception):
        pass

    def capture(J, F, bcs):
        seen.update(F=F, bcs=bcs)
        raise Stop
    with pytest.raises(Stop):
        mod.main(nx=6, ny=6, output_dir=tmp_path, stop_before_device=capture)
    model, mesh, tags = ff.compile_forms(seen["F"])
    assert model.n_species == 4 and model.n_eq == 5 and model.N0 == pytest.approx(3.21877e22)
    assert model.eq_type == ["reaction", "diffusion-reaction", "drift-diffusion-reaction", "drift-diffusion-reaction"]
    assert model.sign == [0.0, 0.0, 1.0, -1.0] and model.grad_diffusion == [False, False, False, True]
    assert model.is_ion == [False, False, True, False]
    assert model.electron_mass == pytest.approx(9.10938215e-31, rel=1e-6)
    assert model.vth[1] == pytest.approx(398.7495614691132) and model.vth[3] == 0.0
    assert model.gamma == [0.06, 0.06, 0.0, 0.0] and model.we_secondary == 5.0
    assert model.ref[0][1:] == [0.3, 5e-4, 0.3] and model.ref[2][1:] == [1.0, 1.0, 1.0]
    assert model.energy_loss == [11.55, 15.76, -11.55, 4.21, -7.34, 0.0, 1.0]
    assert np.array(model.power).shape == (7, 4) and model.quadrature_degree == 4
    rows = model.field_binding.rows
    assert len(rows) == model.n_fields == 33
    assert rows[0] is None and all(r is not None for r in rows[1:4])          # gas has no mobility row
    assert rows[-1] is not rows[-2] and rows[-3] is not rows[-2]              # me_old, me, u_old_e
    assert model.mean_energy_form is None and model.to_c().mean_energy_form == 0
    pieces = seen["F"].pieces
    energy = next(p for p in pieces if isinstance(getattr(p, "f", None), lmea.LmeaEnergySource))
    # the decks' sentinel losses (fedm/functions.py:906-909): with the script's own mean_energy argument, u[0] / u[n - 1]
    # (fedm-gd.py:358), they go to the device as they are; a numeric mean energy is folded into the losses; any
    # other expression is refused
    plain = list(energy.f.loss)
    energy.f.loss = [11.55, 7.77e77, -11.55, 4.21, -7.34, 0.0, 9.99e99]
    energy.f.Ei = 15.76
    m2, _, _ = ff.compile_forms(seen["F"])
    assert m2.energy_loss == energy.f.loss and m2.mean_energy_form == "unknown_ratio" and m2.energy_Ei == 15.76
    c2 = m2.to_c()
    assert c2.mean_energy_form == 1 and c2.energy_Ei == 15.76 and c2.energy_loss[1] == 7.77e77
    script_arg = energy.f.mean_energy
    energy.f.mean_energy = 5.0
    m3, _, _ = ff.compile_forms(seen["F"])
    assert m3.mean_energy_form is None and m3.energy_loss == [11.55, 15.76 - 5.0, -11.55, 4.21, -7.34, 0.0, 5.0]
    energy.f.mean_energy = script_arg * 2.0
    with pytest.raises(NotImplementedError, match="mean_energy must be a number or"):
        ff.compile_forms(seen["F"])
    energy.f.mean_energy, energy.f.loss = script_arg, plain
    # a script that deviates from what the kernels implement is refused, not approximated
    energy.Gamma.mu = energy.Gamma.mu * 2.0
    with pytest.raises(NotImplementedError, match="factor of the electron mobility in the energy flux"):
        ff.compile_forms(seen["F"])


def test_non_logarithmic_weak_forms_compile_and_representations_may_not_be_mixed():
    """fedm/functions.py:350-368 with log_representation=False: the facade lowers the form onto the
    device model with `log_representation = False` (densities enter the sources as u[i]); mixing
    the two representations in one model is an error, not a guess."""
    from fedm_amd import forms
    from fedm_amd import functions as ff
    from fedm_amd.mesh import RectangleMesh
    from fedm_amd.physical_constants import elementary_charge, epsilon_0
    from fedm_amd.termsum import parse
    forms.parameters["form_compiler"]["quadrature_degree"] = 2
    mesh = RectangleMesh((0.0, 0.0), (1.0, 1.0), 4, 4)
    ME = forms.FunctionSpace(mesh, forms.MixedElement([1, 2, 3]))
    V = forms.FunctionSpace(mesh, forms.FiniteElement("Lagrange", None, 1))
    u, v = forms.TrialFunction(ME), forms.TestFunctions(ME)
    r = forms.Expression("x[0]", degree=1, python=lambda x: x[..., 0])
    dt = forms.Expression("time_step", time_step=1e-12, degree=0)
    E = -forms.grad(u[2])
    E_m = forms.sqrt(forms.inner(E, E))
    mu_e, D_e = parse("2.3987*E_m**(-0.26)"), parse("4.3628e-3*E_m**(0.22)")

    def build(log, source_density):
        f_rate = 1e-16 * mu_e * E_m * source_density
        rho = ff.Function_definition(V, "Function", 1)[0]
        for i, z in enumerate((1.0, -1.0)):
            rho += z * (forms.exp(u[i]) if log else u[i]) * elementary_charge / epsilon_0
        F = 0.0
        F += ff.weak_form_balance_equation("reaction", dt, dt, forms.dx, u[0], None, None, v[0], f_rate, 0.0, r, None,
                                           log_representation=log)
        F += ff.weak_form_balance_equation("drift-diffusion-reaction", dt, dt, forms.dx, u[1], None, None, v[1], f_rate,
                                           ff.Flux(-1.0, u[1], D_e, mu_e, E, logarithm_representation=log), r, D_e,
                                           log_representation=log)
        F += ff.weak_form_Poisson_equation(forms.dx, u[2], v[2], rho, r)
        return F
    model, _, _ = ff.compile_forms(build(False, u[1]))
    assert model.log_representation is False and model.to_c().linear_representation == 1
    assert [r_.power for r_ in model.reactions] == [[0, 1]] and model.Z == [1.0, -1.0]
    model, _, _ = ff.compile_forms(build(True, forms.exp(u[1])))
    assert model.log_representation is True and model.to_c().linear_representation == 0
    with pytest.raises(ValueError, match="densities must enter as u\\[i\\]"):
        ff.compile_forms(build(False, forms.exp(u[1])))
    with pytest.raises(ValueError, match="densities must enter as exp"):
        ff.compile_forms(build(True, u[1]))
    forms.parameters["form_compiler"]["quadrature_degree"] = -1


def test_polynomial_smoother_recurrence_and_weights():
    """The polynomial-smoother hierarchy (fedm_amg_setup_poly) folds k Richardson sweeps from a zero
    guess into one product x = S b with S <- S + w Dinv (I - A S), and the post-smoother's error
    propagator into E = I - S_post A (csrc/capi.cpp builds C = R (I - A S_pre), G = E S_pre + S_post,
    Q = E P from them).  Restated with scipy: the recurrence is the sweeps, the composite level is the
    smoothed two-level cycle, and the weights of amg.chebyshev_smoother_weights are the reciprocals of
    Chebyshev roots inside [lambda_max / fraction, lambda_max]."""
    import scipy.sparse as sp
    from fedm_amd import amg
    rng = np.random.default_rng(3)
    n, nc = 60, 12
    A = sp.diags([-1.0, 2.4, -1.0], [-1, 0, 1], shape=(n, n)).tocsr() + sp.diags(rng.uniform(0, 0.3, n))
    P = sp.csr_matrix((np.ones(n), (np.arange(n), np.arange(n) // 5)), shape=(n, nc))
    dinv = 1.0 / A.diagonal()
    w = amg.chebyshev_smoother_weights([(A, P), (P.T @ A @ P, None)], 3, fraction=6.0)[0]
    lam = np.linalg.eigvals((sp.diags(dinv) @ A).toarray()).real.max()
    assert np.all(1.0 / w > 1.04 * lam / 6.0) and np.all(1.0 / w < 1.06 * lam)

    def smoother(ws):
        S = sp.diags(ws[0] * dinv).tocsr()
        for wk in ws[1:]:
            S = S + sp.diags(wk * dinv) @ (sp.identity(n) - A @ S)
        return S.tocsr()

    def sweeps(ws, b, x):
        for wk in ws:
            x = x + wk * dinv * (b - A @ x)
        return x
    b = rng.normal(size=n)
    Spre, Spost = smoother(w), smoother(w[::-1])
    assert np.allclose(Spre @ b, sweeps(w, b, np.zeros(n)), rtol=1e-12, atol=1e-14)
    # composite two-level cycle: b_c = C b, x = G b + Q x_c  ==  pre-smooth, correct, post-smooth
    Ac = (P.T @ A @ P).toarray()
    C = P.T @ (sp.identity(n) - A @ Spre)
    E = sp.identity(n) - Spost @ A
    G, Q = E @ Spre + Spost, E @ P
    xc = np.linalg.solve(Ac, C @ b)
    x_composite = G @ b + Q @ xc
    x = sweeps(w, b, np.zeros(n))
    x = x + P @ np.linalg.solve(Ac, P.T @ (b - A @ x))
    x = sweeps(w[::-1], b, x)
    assert np.allclose(x_composite, x, rtol=1e-11, atol=1e-13)


def test_cpp_expression_subset_evaluates_the_reference_strings_and_refuses_the_rest():
    """The C++ strings of fedm-tof.py:107,116,120 evaluate to the closed forms of
    cases/time_of_flight; anything outside the arithmetic subset is refused, never executed."""
    from fedm_amd import forms
    from fedm_amd.cases import time_of_flight as tof
    x = np.random.default_rng(3).uniform(0.0, 5e-4, (50, 2))
    kw = dict(D=tof.DE, w=tof.WEZ, alpha=tof.ALPHA_E, t=2.6e-9, pi=np.pi)
    u = forms.Expression('std::log(exp(-(pow(x[1]-w*t, 2)+pow(x[0], 2))/(4.0*D*t)+alpha*w*t)/pow(4*D*t*pi,1.5))',
                         degree=3, **kw)
    f = forms.Expression('exp(-(pow(x[1]-w*t, 2)+pow(x[0], 2))/(4.0*D*t)+alpha*w*t)*(w*alpha)'
                         '/(8*pow(pi,1.5)*pow(D*t, 1.5))', degree=2, **kw)
    np.testing.assert_allclose(u(x), tof.analytic_log_density(x, 2.6e-9), rtol=1e-14)
    np.testing.assert_allclose(f(x), tof.source(x, 2.6e-9), rtol=1e-14)
    f.t = 2.7e-9                                       # parameters are looked up at call time
    np.testing.assert_allclose(f(x), tof.source(x, 2.7e-9), rtol=1e-14)
    # the same strings as postfix programs for the device (fedm_ext_source_program): parameters stay
    # symbolic, pi is one here because the script passes it (pi=pi)
    ops, consts, names = forms.expression_program(f)
    assert set(names) == {"w", "t", "D", "alpha", "pi"} and ops.shape[1] == 2 and ops.dtype == np.int32
    for t in (2.6e-9, 2.9e-9):
        f.t = t
        values = forms.run_expression_program(ops, consts, [getattr(f, n) for n in names], x)
        np.testing.assert_array_equal(values, f(x))                 # the same operations in the same order
        np.testing.assert_allclose(values, tof.source(x, t), rtol=1e-12)
    ops_u, consts_u, names_u = forms.expression_program(u)
    np.testing.assert_allclose(forms.run_expression_program(ops_u, consts_u, [getattr(u, n) for n in names_u], x),
                               tof.analytic_log_density(x, 2.6e-9), rtol=1e-14)
    with pytest.raises(NotImplementedError):
        forms.expression_program(forms.Expression(degree=1, python=lambda x: x[..., 0]))
    # parameters that are nodal Functions (fedm-gd.py:258) and parameter-only arithmetic (:272)
    from fedm_amd.mesh import RectangleMesh
    mesh = RectangleMesh((0.0, 0.0), (1.0, 1.0), 3, 3)
    V = forms.FunctionSpace(mesh, "P", 1)
    a = forms.interpolate(forms.Expression("2.0 + x[0]", degree=1), V)
    b = forms.interpolate(forms.Expression("x[1]", degree=1), V)
    we = forms.interpolate(forms.Expression("std::log(a) + b", a=a, b=b, degree=1), V)
    np.testing.assert_allclose(we.vector(), np.log(2.0 + mesh.coords[:, 0]) + mesh.coords[:, 1], rtol=1e-15)
    with pytest.raises(NotImplementedError, match="mesh vertices"):
        forms.Expression("std::log(a) + b", a=a, b=b, degree=1)(x)            # other points than the vertices
    powered = forms.Expression("U0*(1-exp(-t/1e-9))", U0=250.0, t=2e-9, pi=np.pi, degree=0)
    assert powered.value() == pytest.approx(250.0 * (1.0 - np.exp(-2.0)), rel=1e-15)
    powered.t = 0.0
    assert powered.value() == 0.0
    assert forms.Expression('x[0] > 1e-4 ? 1.0 : 0.0', degree=1).code      # construction is free ...
    for bad in ('__import__("os").system("true")', 'x[0] > 1e-4 ? 1.0 : 0.0', 'x.shape', 'q*x[0]',
                'x[0]; x[1]', '(lambda: 1)()'):
        with pytest.raises((NotImplementedError, SyntaxError, NameError)): # ... evaluation is not
            forms.Expression(bad, degree=1)(x)


def test_integer_division_in_expression_strings_is_refused():
    """`x[0]*(1/2) + pow(x[1], 3/2)` is 4.0 at (2, 4) under DOLFIN's C++ JIT (1/2 == 0, 3/2 == 1) and would be
    9.0 with Python's division: refused on the host evaluator and on the device-program path alike."""
    from fedm_amd import forms
    x = np.array([[2.0, 4.0]])
    for bad in ("x[0]*(1/2) + pow(x[1], 3/2)", "pow(x[1], -3/2)", "x[0]*((1+1)/(2*2))"):
        with pytest.raises(NotImplementedError, match="integer division"):
            forms.Expression(bad, degree=1)(x)
        with pytest.raises(NotImplementedError, match="integer division"):
            forms.expression_program(_expression_with_code(forms, bad))
    assert forms.Expression("x[0]*(1.0/2) + pow(x[1], 3.0/2)", degree=1)(x)[0] == pytest.approx(9.0)
    assert forms.Expression("x[0]/2 + 3*x[1]/4", degree=1)(x)[0] == pytest.approx(4.0)     # float / int is fine


def _expression_with_code(forms, code):
    e = forms.Expression("1.0", degree=1)
    e.code = code
    return e


def test_time_of_flight_script_is_lowered_onto_the_device_model(tmp_path, monkeypatch):
    """examples/time_of_flight.py (our own driver for the case of fedm-tof.py).  With the device
    replaced by a recorder: the forms lower to the model of cases/time_of_flight, the states are the
    script's interpolated Functions, and every solve sees the script's step sizes and its source
    Expression at the advanced time, interpolated at the P2 lattice nodes."""
    import importlib.util
    import fedm_amd.device as fdev
    from fedm_amd.cases import time_of_flight as tof
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("tof_example_cpu", root / "examples" / "time_of_flight.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    calls = []

    class Recorder:
        def __init__(self, coords, cells, model, facet_tags=None, dirichlet_dofs=None, dirichlet_vals=None,
                     device=0):
            self.coords, self.cells, self.model, self.t = coords, cells, model, 2.5e-9
            self.state = {}
            assert dirichlet_dofs.size == 0
            calls.append(("create", model))

        def set_state(self, **kw):
            self.state.update({k: np.array(v) for k, v in kw.items()})
            calls.append(("set_state", sorted(kw)))

        def shift_state(self):
            self.state["u_old1"], self.state["u_old"] = self.state["u_old"], self.state["u_new"]
            calls.append(("shift",))

        def set_step(self, dt, dt_old):
            calls.append(("step", dt, dt_old))

        def set_ext_source(self, s, nodal):
            calls.append(("source", s, np.array(nodal)))

        def set_ext_source_program(self, s, ops, consts, n_params):     # the C++ string as a device program
            self.program = (np.array(ops), np.array(consts), n_params)

        def eval_ext_source(self, s, params):                           # what the device kernel computes
            from fedm_amd import forms
            assert len(params) == self.program[2]
            nodes = tof.p2_nodes(self.coords, self.cells)
            calls.append(("source", s, forms.run_expression_program(self.program[0], self.program[1], params, nodes)))

        def newton_solve(self, **kw):
            self.t += 1e-12
            self.state["u_new"] = tof.analytic_log_density(self.coords, self.t)   # "the solver is exact"
            calls.append(("solve", kw["rtol"], kw["max_it"]))

        def get_state(self):
            return self.state["u_new"].reshape(-1, 1)

    monkeypatch.setattr(fdev, "DeviceProblem", Recorder)
    n_num, n_exact, rel = mod.main(nx=8, ny=8, box_width=2.5e-4, box_height=5e-4, T_final=2.503e-9,
                                   t_output=2.503e-9, output_dir=tmp_path, quiet=True)
    model = calls[0][1]
    assert bytes(model.to_c()) == bytes(tof.model().to_c())          # the same descriptor, byte for byte
    assert (model.n_species, model.poisson, list(model.eq_type)) == (1, False, ["drift-diffusion-reaction"])
    assert list(model.drift_w) == [(0.0, tof.WEZ)] and model.D[0].const_value() == tof.DE
    assert model.quadrature_degree == 8 and list(model.ext_source_degree) == [2] and model.axisymmetric
    assert calls[1] == ("set_state", ["u_new", "u_old", "u_old1"])
    mesh = mod.fem.RectangleMesh((0.0, 0.0), (2.5e-4, 5e-4), 8, 8)
    nodes = tof.p2_nodes(mesh.coords, mesh.cells)
    per_step = [c for c in calls[2:] if c[0] != "set_state"]
    assert [c[0] for c in per_step] == ["shift", "step", "source", "solve"] * 3
    assert per_step[1] == ("step", 1e-12, 1e30) and per_step[5] == ("step", 1e-12, 1e30) \
        and per_step[9] == ("step", 1e-12, 1e-12)                 # BDF2 from the third step (fedm-tof.py:166-167)
    for k in range(3):
        np.testing.assert_allclose(per_step[4 * k + 2][2], tof.source(nodes, 2.5e-9 + (k + 1) * 1e-12), rtol=1e-13)
        assert per_step[4 * k + 3] == ("solve", 1e-10, 50)
    assert 0.0 < rel < 1.0 and n_num.shape == n_exact.shape == (81,)     # 8x8 cells across a 35 um pulse
    assert (tmp_path / "mesh" / "mesh info.txt").exists()
    assert "relative_error" in (tmp_path / "relative error.log").read_text()
    assert len(list((tmp_path / "number density" / "electrons").glob("*.vtu"))) == 1


def test_python_callable_expression_source_keeps_the_host_path(monkeypatch):
    """An Expression source given as a Python callable has no device program: `Problem` evaluates it
    on the host at the lattice nodes of its degree and uploads the table before every solve."""
    import fedm_amd.device as fdev
    from fedm_amd import forms, functions as ff
    from fedm_amd.mesh import RectangleMesh
    calls = []

    class Recorder:
        def __init__(self, coords, cells, model, **kw):
            self.model = model

        def set_state(self, **kw):
            pass

        def set_step(self, dt, dt_old):
            calls.append(("step", dt, dt_old))

        def set_ext_source(self, s, nodal):
            calls.append(("host", s, np.array(nodal)))

        def set_ext_source_program(self, *a):
            calls.append(("program",))

        def eval_ext_source(self, *a):
            calls.append(("device",))

        def newton_solve(self, **kw):
            calls.append(("solve",))

    monkeypatch.setattr(fdev, "DeviceProblem", Recorder)
    mesh = RectangleMesh((0.0, 0.0), (1.0, 2.0), 3, 4)
    V = forms.FunctionSpace(mesh, "P", 1)
    u, v = forms.TrialFunction(V), forms.TestFunction(V)
    u_old, u_old1, u_new = forms.Function(V), forms.Function(V), forms.Function(V)
    dt = forms.Expression("time_step", time_step=1e-3, degree=0)
    dt_old = forms.Expression("time_step", time_step=1e30, degree=0)
    r = forms.Expression("x[0]", degree=1)
    w = forms.interpolate(forms.Constant(("0", 2.0)), forms.VectorFunctionSpace(mesh, "P", 1))
    D = forms.interpolate(forms.Constant(0.5), V)
    f = forms.Expression(degree=1, python=lambda x, e: e.t * (x[..., 0] + 2.0 * x[..., 1]), t=3.0)
    Gamma = -forms.grad(D * forms.exp(u)) + w * forms.exp(u)
    F = ff.weak_form_balance_equation_log_representation("drift-diffusion-reaction", dt, dt_old, forms.dx, u, u_old,
                                                         u_old1, v, f, Gamma, r)
    forms.parameters["form_compiler"]["quadrature_degree"] = 4
    problem = ff.Problem(forms.derivative(forms.action(F, u_new), u_new, u), forms.action(F, u_new), [])
    assert list(problem.device.model.ext_source_degree) == [1] and problem.device.model.quadrature_degree == 4
    solver = ff.PETScSNESSolver()
    solver.solve(problem, None)
    f.t = 5.0
    solver.solve(problem, None)
    kinds = [c[0] for c in calls]
    assert "program" not in kinds and "device" not in kinds and kinds.count("host") == 2
    vertices = mesh.coords[mesh.cells]                                   # degree 1: the lattice nodes are the vertices
    expected = vertices[..., 0] + 2.0 * vertices[..., 1]
    host = [c for c in calls if c[0] == "host"]
    np.testing.assert_allclose(host[0][2], 3.0 * expected)
    np.testing.assert_allclose(host[1][2], 5.0 * expected)


def test_subdomain_predicates_are_what_marking_boundaries_tags():
    """LineSubDomain / CircleSubDomain (fedm/functions.py:49-84) as point predicates: the line one
    agrees with the facet tags of Marking_boundaries, the circle one carries the reference's defect."""
    from fedm_amd import functions as ff
    from fedm_amd.mesh import RectangleMesh
    mesh = RectangleMesh((0.0, 0.0), (1.0, 2.0), 4, 6)
    boundaries = [['line', 0.0, 0.0, 0.0, 1.0], ['line', 2.0, 2.0, 0.0, 1.0]]
    tags = ff.Marking_boundaries(mesh, boundaries)
    cell, local = mesh.exterior_facets()
    ends = (np.array([1, 0, 0])[local], np.array([2, 2, 1])[local])
    for idx, (_, z1, z2, r1, r2) in enumerate(boundaries):
        sub = ff.LineSubDomain((r1, r2), (z1, z2))
        for c, l, a, b in zip(cell, local, *ends):
            pa, pb = mesh.coords[mesh.cells[c, a]], mesh.coords[mesh.cells[c, b]]
            inside = all(sub.inside(p, True) for p in (pa, pb, 0.5 * (pa + pb)))
            assert inside == (tags[c, l] == idx + 1)
        assert not sub.inside((0.5, z1), False)                          # not on the boundary
    tip = ff.CircleSubDomain(-0.5, 0.0, 0.5, 0.01)
    assert tip.inside((0.0, 0.0), True) and not tip.inside((0.0, 0.0), False) and not tip.inside((0.3, 0.3), True)
    with pytest.raises(AttributeError, match="_gap_length"):
        ff.CircleSubDomain(2.5, 0.0, 0.5, 2.0).inside((0.0, 2.0), True)


def test_normal_vector_is_the_projected_outward_normal():
    """Normal_vector (fedm/functions.py:1133-1151): interior vertices zero, the exact normal along a
    straight side away from the corners, the boundary mass system satisfied."""
    from fedm_amd import functions as ff
    from fedm_amd.mesh import RectangleMesh
    mesh = RectangleMesh((0.0, 0.0), (1.0, 2.0), 16, 32)
    n = np.asarray(ff.Normal_vector(mesh).vector())
    x = mesh.coords
    interior = (x[:, 0] > 1e-12) & (x[:, 0] < 1 - 1e-12) & (x[:, 1] > 1e-12) & (x[:, 1] < 2 - 1e-12)
    assert n.shape == (mesh.num_vertices(), 2) and np.all(n[interior] == 0.0)
    mid = lambda m: m & ~interior
    bottom = mid((np.abs(x[:, 1]) < 1e-12) & (np.abs(x[:, 0] - 0.5) < 0.2))
    right = mid((np.abs(x[:, 0] - 1.0) < 1e-12) & (np.abs(x[:, 1] - 1.0) < 0.4))
    assert bottom.sum() >= 5 and right.sum() >= 5
    np.testing.assert_allclose(n[bottom], np.tile([0.0, -1.0], (bottom.sum(), 1)), atol=1e-3)
    np.testing.assert_allclose(n[right], np.tile([1.0, 0.0], (right.sum(), 1)), atol=1e-3)
    corner = np.argmin(np.abs(x[:, 0]) + np.abs(x[:, 1]))                 # (0, 0): both sides meet
    assert n[corner, 0] < -0.3 and n[corner, 1] < -0.3


def test_host_poisson_assemble_and_solve_like_the_scripts_initial_solve():
    """lhs / rhs / assemble / bc.apply / solve as fedm-streamer.py:203-215 uses them, on the host: a
    manufactured axisymmetric problem (Phi = z^2 + r^2 has -(1/r)(r Phi_r)_r - Phi_zz = -6) and the plane
    default r = 0.5/pi with a linear solution."""
    from fedm_amd import forms, functions as ff
    from fedm_amd.mesh import RectangleMesh
    mesh = RectangleMesh((0.0, 0.0), (1.0, 2.0), 24, 48)
    V = forms.FunctionSpace(mesh, "P", 1)
    u, v, Phi = forms.TrialFunction(V), forms.TestFunction(V), forms.Function(V)
    x = mesh.coords
    forms.parameters["form_compiler"]["quadrature_degree"] = 2
    exact = lambda p: p[..., 1] ** 2 + p[..., 0] ** 2
    every = lambda p, on_boundary: on_boundary
    r = forms.Expression("x[0]", degree=1)
    n0 = forms.interpolate(forms.Constant(np.log(6.0)), V)
    Fp = ff.weak_form_Poisson_equation(forms.dx, u, v, -forms.exp(n0), r)      # f = -6 through exp(Function)
    A, b = forms.assemble(forms.lhs(Fp)), forms.assemble(forms.rhs(Fp))
    bcs = [forms.DirichletBC(V, exact, every)]
    [bc.apply(A) for bc in bcs]
    [bc.apply(b) for bc in bcs]
    forms.solve(A, Phi.vector(), b, "mumps")
    assert np.abs(Phi.vector() - exact(x)).max() < 2e-3 * np.abs(exact(x)).max()   # second order, h = 1/24
    # plane problem, Laplace: Phi = 3 z between Phi(0) = 0 and Phi(2) = 6, exact in P1
    Fp = ff.weak_form_Poisson_equation(forms.dx, u, v, forms.Constant(0.0))
    A, b = forms.assemble(forms.lhs(Fp)), forms.assemble(forms.rhs(Fp))
    bottom = lambda p, on_boundary: forms.near(p[1], 0.0) and on_boundary
    top = lambda p, on_boundary: forms.near(p[1], 2.0) and on_boundary
    for bc in (forms.DirichletBC(V, forms.Constant(0.0), bottom), forms.DirichletBC(V, forms.Constant(6.0), top)):
        bc.apply(A)
        bc.apply(b)
    forms.solve(A, Phi.vector(), b)
    np.testing.assert_allclose(Phi.vector(), 3.0 * x[:, 1], atol=1e-12)
    # the same through Poisson_solver (fedm/functions.py:1154-1161): A with its boundary rows, L, bcs
    again = forms.Function(V)
    bcs = [forms.DirichletBC(V, forms.Constant(0.0), bottom), forms.DirichletBC(V, forms.Constant(6.0), top)]
    ff.Poisson_solver(A, forms.rhs(Fp), None, bcs, again, solver_type="gmres", preconditioner="hypre_amg")
    np.testing.assert_allclose(again.vector(), 3.0 * x[:, 1], atol=1e-12)


def test_streamer_script_runs_up_to_the_device_and_lowers_to_the_case_model(tmp_path, monkeypatch):
    """examples/streamer_discharge.py (our own driver for the case of fedm-streamer.py).  Up to the
    point where the device context would be created (no GPU here): the initial conditions from the C++
    strings, the host-side initial Poisson solve, the forms lowered to the model of cases/streamer, the
    initial values of the mixed Functions collected by the reverse assigner -- on the tensor-product
    mesh and on the unstructured one that is written to and loaded from `mesh.xml`."""
    import importlib.util
    import fedm_amd.functions as ff
    from fedm_amd.cases import streamer
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("streamer_example_cpu", root / "examples" / "streamer_discharge.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen = {}

    class Stop(Exception):
        pass

    def capture(J, F, bcs):
        seen.update(F=F, bcs=bcs)
        raise Stop
    with pytest.raises(Stop):
        mod.main(mesh_spacing=1.5e-4, output_dir=tmp_path / "unstructured", quiet=True, stop_before_device=capture)
    model_u, mesh_u, _ = ff.compile_forms(seen["F"])
    assert (tmp_path / "unstructured" / "mesh" / "mesh.xml").exists() and mesh_u.num_vertices() > 1000
    with pytest.raises(Stop):
        mod.main(cells=12, output_dir=tmp_path, quiet=True, stop_before_device=capture)
    model, mesh, tags = ff.compile_forms(seen["F"])
    assert bytes(model.to_c()) == bytes(model_u.to_c())            # the model does not depend on the mesh
    ref = streamer.model()
    assert (model.n_species, model.poisson, list(model.eq_type), list(model.Z)) == (2, True, list(ref.eq_type), [1.0, -1.0])
    assert model.bc_kind == ref.bc_kind and model.quadrature_degree == 2 and model.axisymmetric
    for E in (1e5, 3e7):
        assert model.reactions[0].k(E) == pytest.approx(ref.reactions[0].k(E), rel=1e-14)
    assert len(seen["bcs"]) == 2
    # what Problem() would upload: the parts the reverse assigner collected for u_new / u_old
    parts = seen["F"].u_new.parts
    assert len(parts) == 3
    x = mesh.coords
    ions, electrons = streamer.initial_log_densities(x)
    np.testing.assert_allclose(parts[0].vector(), ions, rtol=1e-14)
    np.testing.assert_allclose(parts[1].vector(), electrons, rtol=1e-14)
    phi = np.asarray(parts[2].vector())
    assert abs(phi[np.argmin(x[:, 1])]) < 1e-9 and phi[np.argmax(x[:, 1])] == pytest.approx(18750.0)
    assert np.all(np.diff(phi[np.argsort(x[:, 1], kind="stable")][::13]) >= -1e-6)      # rises from cathode to anode
    assert (tmp_path / "potential" / "Phi" / "Phi000000.vtu").exists()
    assert (tmp_path / "number density" / "electrons" / "electrons000000.vtu").exists()


def test_dolfin_like_names_of_the_star_import(tmp_path):
    """`from fedm_amd.forms import *` stands for `from dolfin import *` in the example scripts: the mesh
    constructors and file objects they use come with it (fedm-streamer.py:117-122,132; fedm-gd.py:163,265;
    fedm-tof.py:89)."""
    from fedm_amd import forms, mesh_io
    mesh = forms.RectangleMesh(forms.Point(0, 0), forms.Point(1.0, 2.0), 3, 4)
    assert mesh.num_cells() == 24 and mesh.ufl_cell() == "triangle"
    assert forms.RectangleMesh((0, 0), (1, 1), 2, 2, "crossed").num_cells() == 16
    mesh_io.write_dolfin_xml(mesh, tmp_path / "mesh.xml")
    again = forms.Mesh(str(tmp_path / "mesh.xml"))
    assert np.array_equal(again.cells, mesh.cells) and np.allclose(again.coords, mesh.coords)
    tags = np.arange(72, dtype=np.int8).reshape(24, 3) % 5
    forms.File(str(tmp_path / "mesh" / "boundary_mesh_function.pvd")) << tags
    assert np.array_equal(np.loadtxt(tmp_path / "mesh" / "boundary_mesh_function.txt", dtype=int), tags)
    f = forms.Function(forms.FunctionSpace(mesh, "P", 1))
    out = forms.File(str(tmp_path / "f.pvd"))
    out << (f, 0.5)
    out << (f, 1.0)
    assert sorted(p.name for p in tmp_path.glob("f*")) == ["f.pvd", "f000000.vtu", "f000001.vtu"]
    assert forms.XDMFFile.Encoding.HDF5 == "HDF5"


def test_models_beyond_the_instantiated_kernels_are_refused_not_truncated():
    """The descriptors are fixed-size structs and the LFA kernels are instantiated for 1-2 species without
    and 1-4 species with a Poisson equation (the four-species model on the device: tests/test_gpu_parity.py): a deck
    beyond that is refused with a message that names the limits -- by the Python descriptor (5 species do not fit
    the struct) and by fedm_ctx_create (3 species without a Poisson equation fit the struct but no kernel exists),
    before any device is touched."""
    import ctypes as C
    from fedm_amd import _lib
    from fedm_amd.device import Model, Reaction
    from fedm_amd.termsum import TermSum
    big = Model(n_species=5, poisson=True, eq_type=["reaction"] * 5, Z=[0.0] * 5)
    with pytest.raises(ValueError, match="n_species must be 1..4"):
        big.to_c()
    three = Model(n_species=3, poisson=False, eq_type=["reaction"] * 3, Z=[1.0, -1.0, 0.0])
    many = Model(n_species=2, poisson=True, eq_type=["reaction"] * 2, Z=[1.0, -1.0],
                 reactions=[Reaction(TermSum.const(1.0), power=[1, 0], net=[0, 1])] * 9)
    with pytest.raises(ValueError, match="at most 8 reactions"):
        many.to_c()
    lib = _lib.load()
    coords = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    cells = np.array([[0, 1, 2]], dtype=np.int32)
    mesh = _lib.MeshDesc()
    mesh.n_vertices, mesh.n_cells = 3, 1
    mesh.coords = coords.ctypes.data_as(C.POINTER(C.c_double))
    mesh.cells = cells.ctypes.data_as(C.POINTER(C.c_int32))
    handle = C.c_void_p()
    md = three.to_c()
    assert lib.fedm_ctx_create(C.byref(mesh), C.byref(md), 0, C.byref(handle)) == -2 and not handle
    msg = _lib.last_error()
    assert "unsupported LFA model: 3 species," in msg and "1-4 with a Poisson equation" in msg


def test_binding_brings_torch_in_before_the_hip_library():
    """Loaded before torch, libfedm_hip.so would pull /opt/rocm's HIP runtime in next to the one PyTorch-ROCm
    carries, and whichever initialises second finds no device (seen on the MI355X box).  _lib.load() therefore
    imports torch first; a fresh interpreter shows the order."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from fedm_amd import _lib; assert 'torch' not in sys.modules; "
            "_lib.load(); assert 'torch' in sys.modules; "
            "maps = open('/proc/self/maps').read(); "
            "hip = {l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}; print(len(hip))" % str(ROOT))
    done = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    assert done.stdout.strip() == "1"                      # one HIP runtime in the process, not two
