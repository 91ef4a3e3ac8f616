"""GPU parity on a locally refined UNSTRUCTURED mesh (Delaunay, vertex valence 4-9, arbitrary
numbering) that went through the DOLFIN XML reader -- the kind of mesh the reference's streamer case
loads (``Mesh('mesh.xml')``, examples/streamer_discharge/fedm-streamer.py:116; the file itself is a
missing blob, .MISSING_LARGE_BLOBS:2, so the mesh is `fedm_amd.cases.streamer.refined_mesh`).

Everything the device path derives from the mesh meets variable valence here for the first time:
the bisection vertex order, the 64-vertex patches and their micro-colouring, the sliced block-ELL
padding, the smoothed-aggregation hierarchy, RCB's cuts through an unstructured vertex cloud.
Tolerances as in test_gpu_parity.py (fp64; element-level 1e-11 / 1e-10)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent))
import mp_results  # noqa: E402
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
H_FINE = 30e-6             # 17 k vertices, 270 patches


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    from oracle import streamer as ost
    from oracle.mesh import Mesh as OMesh
    from fedm_amd.cases import streamer
    msh = streamer.refined_mesh(H_FINE, xml_path=tmp_path_factory.mktemp("mesh") / "mesh.xml")
    assert msh.num_vertices() >= 16000
    omodel = ost.build(OMesh(msh.coords, msh.cells))
    U0 = ost.initial_state(omodel)
    prob = streamer.device_problem(msh.coords, msh.cells)
    yield msh, omodel, U0, prob
    prob.close()


def _rel_rows(A, B):
    D = abs(A - B)
    scale = np.maximum(abs(B).max(axis=1).toarray().ravel(), 1e-300)
    return (sp.diags(1.0 / scale) @ D).max()


def _developed_state(msh, seed):
    """Steep density gradients and a non-uniform field, as behind a streamer head in the channel."""
    from oracle import streamer as ost
    r, z = msh.coords[:, 0], msh.coords[:, 1]
    rng = np.random.default_rng(seed)
    nv = msh.coords.shape[0]
    head = np.exp(-(r ** 2 + (z - 0.008) ** 2) / (0.6e-3) ** 2)
    U = np.zeros((nv, 3))
    U[:, 0] = np.log(1e13 + 4e19 * head) + 0.02 * rng.standard_normal(nv)
    U[:, 1] = np.log(1e13 + 3e19 * head) + 0.02 * rng.standard_normal(nv)
    U[:, 2] = ost.U_W * z / ost.BOX * (1.0 + 0.3 * head) + 5.0 * rng.standard_normal(nv)
    return U, U + 0.01 * rng.standard_normal(U.shape), U + 0.02 * rng.standard_normal(U.shape)


def test_the_lean_patch_kernels_run_on_this_mesh(setup):
    """Not a fallback path: patches of at most 192 cells, little ELL padding."""
    msh, omodel, U0, prob = setup
    sz = prob.sizes()
    assert sz["max_patch_cells"] <= 192
    assert sz["stored_blocks"] < 1.08 * sz["nnz_blocks"]
    assert prob.assembly_variant() == "lds-patches/one-pass"


@pytest.mark.parametrize("dt,dt_old", [(5e-12, 1e30), (5e-12, 4.977e-12)])
def test_residual_jacobian_and_product_against_both_cpu_statements(setup, dt, dt_old):
    from oracle import cpu_backend as cb
    msh, omodel, U0, prob = setup
    U, Uo, Uo1 = _developed_state(msh, 11)
    F_np, J_np = omodel.residual_jacobian(U, Uo, Uo1, dt, dt_old)
    cprob = cb.CpuProblem(omodel)
    cprob.set_state(U, Uo, Uo1)
    F_c, J_c = cprob.residual_jacobian(dt, dt_old)
    cprob.close()
    prob.set_state(U, Uo, Uo1)
    prob.set_step(dt, dt_old)
    F_gpu, fnorm = prob.residual()          # residual-only kernel
    prob.jacobian()                         # F + J kernel
    J_gpu = prob.jacobian_csr()
    scale = np.abs(F_np).reshape(-1, 3).max(axis=0)
    for F_ref in (F_np, F_c):
        assert (np.abs(F_gpu - F_ref).reshape(-1, 3) / scale).max() < 1e-11
    assert fnorm == pytest.approx(np.linalg.norm(F_np), rel=1e-11)
    assert _rel_rows(J_gpu, J_np) < 1e-10
    assert _rel_rows(J_gpu, J_c) < 1e-10
    x = np.random.default_rng(5).normal(size=prob.n)
    y, yc = prob.spmv(x), J_np @ x
    assert np.abs(y - yc).max() / np.abs(yc).max() < 1e-11


def test_every_assembly_variant_gives_the_same_system(setup, monkeypatch):
    """The bitwise reproducible global-colouring assembly (FEDM_ASSEMBLY=colour), the unrolled patch routine
    and the row-phase kernels against the default one-pass kernels, on the unstructured pattern."""
    from fedm_amd.cases import streamer
    msh, omodel, U0, prob = setup
    U, Uo, Uo1 = _developed_state(msh, 3)
    prob.set_state(U, Uo, Uo1)
    prob.set_step(5e-12, 4e-12)
    prob.jacobian()
    F_ref, J_ref = prob.residual()[0], prob.jacobian_csr()
    for env in (dict(FEDM_ASSEMBLY="colour"), dict(FEDM_ASSEMBLY_LEAN="0"), dict(FEDM_ASSEMBLY_LEAN="2")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        other = streamer.device_problem(msh.coords, msh.cells)
        for k in env:
            monkeypatch.delenv(k)
        other.set_state(U, Uo, Uo1)
        other.set_step(5e-12, 4e-12)
        other.jacobian()
        F, J = other.residual()[0], other.jacobian_csr()
        other.close()
        assert np.abs(F - F_ref).max() / np.abs(F_ref).max() < 1e-11
        assert _rel_rows(J, J_ref) < 1e-10


def test_poisson_solve_with_the_multigrid_built_on_this_mesh(setup):
    from fedm_amd.cases import streamer
    msh, omodel, U0, prob = setup
    U = U0.copy()
    U[:, 2] = 0.0
    prob.set_state(U, U, U)
    prob.setup_multigrid(**streamer.MULTIGRID)
    its = prob.poisson_solve(rtol=1e-13)
    Phi = prob.get_state()[:, 2]
    assert 0 < its < 40                                      # the V-cycle works as a preconditioner
    assert np.abs(Phi - U0[:, 2]).max() / np.abs(U0[:, 2]).max() < 1e-9


def test_newton_step_and_error_norm(setup):
    from oracle.controller import field_error
    from oracle.newton import newton_solve
    msh, omodel, U0, prob = setup
    prob.set_state(U0, U0, U0)
    prob.set_step(5e-12, 1e30)
    its, _ = prob.newton_solve(rtol=1e-8, max_it=20, ksp_rtol=1e-10)
    U_gpu = prob.get_state()
    U_cpu = U0.copy()
    its_cpu, _ = newton_solve(omodel, U_cpu, U0, U0, 5e-12, 1e30, 1e-8, 20)
    assert its == its_cpu
    d = np.abs(U_gpu - U_cpu).max(axis=0) / np.abs(U_cpu).max(axis=0)
    assert d.max() < 1e-9
    assert prob.field_error(1) == pytest.approx(field_error(U_cpu[:, 1], U0[:, 1]), rel=1e-7)


def test_error_log_of_five_adaptive_steps(setup):
    """The script-level loop (fedm-streamer.py:304-340) with the field split + multigrid solver
    against the oracle's direct solves: same rows in `relative error.log`."""
    from oracle import streamer as ost
    from fedm_amd.cases import streamer
    msh, omodel, U0, prob = setup
    _, st, _, _ = ost.run(mesh=omodel.mesh, max_steps=5)
    out = streamer.run(prob, max_steps=5)
    assert len(out["log"]) == len(st.log)
    assert np.allclose(np.array(out["log"]), np.array(st.log), rtol=2e-4)
    assert out["linear_iterations"] <= 12 * 5                # the preconditioner holds on this mesh


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from fedm_amd.cases import streamer, streamer_distributed
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run = streamer_distributed.Runner(None, rank, world, 0, transport="torch",
                                          mesh=streamer.refined_mesh(60e-6), relative_tolerance=1e-9)
        run.solver.parameters["krylov_relative_tolerance"] = 1e-11
        run.initialise()
        for _ in range(3):
            run.step()
        U = run.prob.get_state()[:run.lm.n_owned]
        q.put((rank, run.lm.vertex_global[:run.lm.n_owned], U, run.log_rows(), int(run.lm.n_ghost)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_one_gpu_on_the_unstructured_mesh():
    """Partition (RCB through an unstructured vertex cloud), ghost rows, halo exchanges, all-reduced
    dots and the distributed multigrid on this mesh: two processes on one GPU over the host-staged
    transport against the single-GPU solve."""
    import torch.multiprocessing as mp
    from fedm_amd.cases import streamer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = mp_results.collect(procs, q, len(procs), 300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    msh = streamer.refined_mesh(60e-6)
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob, relative_tolerance=1e-9)
    st.solver.parameters["krylov_relative_tolerance"] = 1e-11
    st.initialise()
    for _ in range(3):
        st.step()
    U_ref = prob.get_state()
    prob.close()
    U = np.zeros_like(U_ref)
    for _, gids, Uloc, _, n_ghost in res:
        U[gids] = Uloc
        assert n_ghost > 0
    scale = np.abs(U_ref).max(axis=0)
    assert (np.abs(U - U_ref) / scale).max() < 1e-8
    ref_log = np.array(st.log_rows())
    for r in res:
        assert np.allclose(np.array(r[3]), ref_log, rtol=1e-6)


def test_lmea_kernels_on_an_unstructured_mesh(monkeypatch):
    """The glow-discharge (LMEA) assembly -- element buffer + gather by stored matrix position, coloured
    dual-number cross-check, 'flux source' walls -- has only ever run on DOLFIN's crossed meshes: residual
    and Jacobian of all three device variants against the oracle on a locally refined Delaunay mesh of the
    1 cm x 1 cm discharge domain (refined towards the cathode sheath at z = 0)."""
    from oracle import gd as ogd
    from oracle.mesh import Mesh as OMesh
    from fedm_amd import functions as ff, meshgen
    from fedm_amd.cases import glow_discharge as gdc
    deck = ROOT / "decks" / "glow_discharge" / "file_input" / "4_particles"
    size = meshgen.box_distance_size((0.0, 0.01, 0.0, 0.0015), 1.0e-4, 0.3, 1.2e-3)
    msh = meshgen.refined_rectangle(0.01, 0.01, size, 1.0e-4, n_levels=5)
    assert 2000 < msh.num_vertices() < 20000
    for variant in ("3", "4", "2", "0"):
        monkeypatch.setenv("FEDM_GD_HAND", variant)
        case = gdc.Case(device_pipeline=False, mesh=msh)
        o = ogd.GlowDischarge(deck, mesh=OMesh(msh.coords, msh.cells))
        nv = o.mesh.nv
        rng = np.random.default_rng(0)
        me_old = 3.0 + rng.normal(0, 0.3, nv)
        me = me_old + rng.normal(0, 0.05, nv)
        U = case.U.copy()
        U[:, 0] = np.log(me) + U[:, 3] + rng.normal(0, 0.05, nv)
        U[:, 1:4] += rng.normal(0, 0.2, (nv, 3))
        U[:, 4] = -100.0 * (1 - o.mesh.coords[:, 1] / 0.01) + rng.normal(0, 3.0, nv)
        Uo = U + rng.normal(0, 0.02, U.shape)
        Uo1 = U + rng.normal(0, 0.02, U.shape)
        redE = o.reduced_field(U[:, 4])
        assert np.allclose(case.project_reduced_field(U[:, 4]), redE, rtol=1e-10)
        co = o.coefficients(me_old, redE)
        t, dt, dt_old = 2e-12, 1.1e-12, 0.7e-12
        F_cpu, J_cpu = o.residual_jacobian(U, Uo, Uo1, dt, dt_old, co, me_old, me, Uo[:, 3], o.dirichlet_values(t))
        case.mean_energy_old.vector()[:] = me_old
        case.mean_energy.vector()[:] = me
        case.redE.vector()[:] = redE
        ff.Transport_coefficient_interpolation("update", case.mu_dep, case.N0, case.Tgas, case.mu, case.mu_x,
                                               case.mu_y, case.mean_energy_old, case.redE)
        ff.Transport_coefficient_interpolation("update", case.D_dep, case.N0, case.Tgas, case.D, case.D_x,
                                               case.D_y, case.mean_energy_old, case.redE, case.mu)
        ff.Rate_coefficient_interpolation("update", case.k_dep, case.k, case.k_x, case.k_y,
                                          case.mean_energy_old, case.redE, Te=0, Tgas=0)
        case.U = Uo
        case.upload_fields()
        prob = case.prob
        prob.set_state(U, Uo, Uo1)
        prob.set_step(dt, dt_old)
        prob.set_dirichlet_values(case.dirichlet_values(t))
        F_gpu, _ = prob.residual()
        scale = np.abs(F_cpu).reshape(-1, 5).max(axis=0)
        assert (np.abs(F_gpu - F_cpu).reshape(-1, 5) / scale).max() < 1e-11, variant
        prob.jacobian()
        J_gpu = prob.jacobian_csr()
        assert _rel_rows(J_gpu, J_cpu) < 1e-9, variant
        prob.close()


def test_time_of_flight_kernels_on_an_unstructured_mesh():
    """The third model family's path -- one equation without Poisson, degree-8 quadrature, an Expression source
    given as a per-cell table of P2 nodal values (so the patch cell order may NOT turn cells), the unrolled patch
    routine -- on a refined Delaunay mesh of the time-of-flight domain: residual and Jacobian against the oracle."""
    from oracle import tof as otof
    from oracle.forms import LFAModel
    from oracle.mesh import Mesh as OMesh
    from fedm_amd import meshgen
    from fedm_amd.cases import time_of_flight as tof
    from fedm_amd.device import DeviceProblem
    w, h = 2.5e-4, 5e-4
    size = meshgen.box_distance_size((0.0, 0.6e-4, 3.5e-4, 5e-4), 2e-6, 0.25, 4e-5)      # fine around the pulse
    msh = meshgen.refined_rectangle(w, h, size, 2e-6, n_levels=5)
    assert msh.num_vertices() > 3000
    prob = DeviceProblem(msh.coords, msh.cells, tof.model())
    omesh = OMesh(msh.coords, msh.cells)
    om = LFAModel(omesh, 1, False, ["drift-diffusion-reaction"], [-1.0], D=[otof.DE], drift_w=[(0.0, otof.WEZ)], qdeg=8)
    t0, dt = 2.5e-9, 1e-12
    rng = np.random.default_rng(0)
    U = otof.log_density(omesh.coords, t0, 3e-16)[:, None] + rng.normal(0, 0.1, (omesh.nv, 1))
    Uo = otof.log_density(omesh.coords, t0)[:, None]
    Uo1 = Uo + rng.normal(0, 0.1, (omesh.nv, 1))
    assert np.allclose(tof.p2_nodes(msh.coords, msh.cells), otof.cell_nodes(omesh, 2))
    om.set_ext_source(0, 2, otof.source(otof.cell_nodes(omesh, 2), t0 + dt))
    prob.set_ext_source(0, tof.source(tof.p2_nodes(msh.coords, msh.cells), t0 + dt))
    for dt_old in (1e30, 2e-12):
        prob.set_state(U, Uo, Uo1)
        prob.set_step(dt, dt_old)
        F_gpu, _ = prob.residual()
        F_cpu, J_cpu = om.residual_jacobian(U, Uo, Uo1, dt, dt_old)
        assert np.abs(F_gpu - F_cpu).max() / np.abs(F_cpu).max() < 1e-11
        prob.jacobian()
        assert _rel_rows(prob.jacobian_csr(), J_cpu) < 1e-10
    prob.close()


def test_multi_gpu_example_script_against_the_one_gpu_run(tmp_path):
    """examples/streamer_discharge_multi_gpu.py as a user launches it (torch.distributed.run, two ranks, here
    sharing the one GPU over the host-staged transport): its error log and gathered fields against the
    single-GPU Stepper on the same mesh.  The script loads libfedm_hip before torch initialises the GPU --
    the order in which the two HIP runtimes of the image used to collide (fedm_amd/_lib.py)."""
    import subprocess
    from fedm_amd.cases import streamer
    out = tmp_path / "mgpu"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "examples" / "streamer_discharge_multi_gpu.py"), "--mesh-spacing", "6e-5", "--end", "3e-11",
           "--out", str(out), "--share-one-gpu"]
    done = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=tmp_path)
    assert done.returncode == 0, done.stderr[-3000:]
    assert "halo exchanges" in done.stdout
    U = np.load(out / "state.npy")
    rows = np.array([[float(v) for v in line.split()] for line in open(out / "relative error.log")])
    for name in ("Ions", "electrons", "Phi"):
        assert (out / name / f"{name}.pvd").exists()
    msh = streamer.refined_mesh(6e-5)
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob)
    st.initialise()
    while st.t < 3e-11 * (1.0 - 1e-6):
        st.step()
    U_ref = prob.get_state()
    prob.close()
    ref_rows = np.array(st.log_rows())
    assert rows.shape == ref_rows.shape and np.allclose(rows, ref_rows, rtol=2e-3)
    assert (np.abs(U - U_ref) / np.abs(U_ref).max(axis=0)).max() < 1e-5
