"""BASELINE.json's full sizes on the device, checked through size-independent properties
(no oracle run needed): the 1 M-DOF streamer workload of bench.py and the 51 681-DOF
time-of-flight mesh of examples/time_of_flight/fedm-tof.py:87."""
import numpy as np
import sys

import pytest

sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent))
import mp_results  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def streamer_1m():
    from fedm_amd.cases import streamer
    msh = streamer.mesh(576, 4.0)
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob)
    st.initialise()
    return prob, st


def test_sizes_and_layout(streamer_1m):
    prob, _ = streamer_1m
    sz = prob.sizes()
    assert sz["n_vertices"] == 577 * 577 and sz["n_cells"] == 2 * 576 * 576 and sz["n_eq"] == 3
    assert prob.n == 998787
    # P1 on a right-diagonal grid: 7-point vertex stencil; ELL padding stays small
    assert sz["nnz_blocks"] == 7 * 577 * 577 - 2 * (2 * 577 + 2 * 575) - 2 * 2 - 2  # interior 7, edges 5, corners 4/3
    assert sz["stored_blocks"] / sz["nnz_blocks"] < 1.05


def test_initial_potential_solves_poisson(streamer_1m):
    """After fedm_poisson_solve the potential rows of the residual vanish relative to the
    size of their two terms, and the Dirichlet values hold exactly."""
    from fedm_amd.cases import streamer
    prob, _ = streamer_1m
    U = prob.get_state()
    prob.set_step(5e-12, 1e30)
    F, _ = prob.residual()
    Fphi = F.reshape(-1, 3)[:, 2]
    z = prob.coords[:, 1]
    assert np.all(U[np.abs(z) < 3e-16, 2] == 0.0)
    assert np.all(U[np.abs(z - streamer.BOX) < 3e-16, 2] == streamer.U_W)
    # scale: |K Phi| row sums ~ 2 pi r * U_w
    scale = 2 * np.pi * prob.coords[:, 0].max() * streamer.U_W
    assert np.abs(Fphi).max() / scale < 1e-9


def test_jacobian_is_the_derivative_of_the_residual(streamer_1m):
    """(F(u + e v) - F(u - e v)) / 2e == J v on the device, all 998 787 rows."""
    prob, _ = streamer_1m
    U = prob.get_state()
    prob.set_state(U, U, U)
    prob.set_step(5e-12, 1e30)
    rng = np.random.default_rng(0)
    v = rng.normal(size=U.shape) * np.array([1e-3, 1e-3, 1e-1])
    z = prob.coords[:, 1]
    v[(np.abs(z) < 3e-16) | (np.abs(z - 0.0125) < 3e-16), 2] = 0.0     # keep Dirichlet values
    prob.jacobian()
    Jv = prob.spmv(v.ravel())
    e = 1e-3
    prob.set_state(u_new=U + e * v)
    Fp, _ = prob.residual()
    prob.set_state(u_new=U - e * v)
    Fm, _ = prob.residual()
    prob.set_state(u_new=U)
    fd = (Fp - Fm) / (2 * e)
    for c in range(3):
        a, b = fd.reshape(-1, 3)[:, c], Jv.reshape(-1, 3)[:, c]
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-6, c


def test_spmv_is_linear(streamer_1m):
    prob, _ = streamer_1m
    rng = np.random.default_rng(1)
    x, y = rng.normal(size=prob.n), rng.normal(size=prob.n)
    lhs = prob.spmv(2.5 * x - 0.75 * y)
    rhs = 2.5 * prob.spmv(x) - 0.75 * prob.spmv(y)
    assert np.abs(lhs - rhs).max() / np.abs(rhs).max() < 1e-13


def test_step_converges_and_controller_accepts(streamer_1m):
    """One adaptive step: Newton meets its tolerance, the step is accepted (error < ttol),
    the state stays finite and moves by about the logged relative error."""
    prob, st = streamer_1m
    U0 = prob.get_state()
    t = st.step()
    rep = prob.last_report
    assert rep.converged and rep.fnorm <= 1e-4 * rep.fnorm0 and 1 <= rep.iterations <= 20
    rows = st.log_rows()
    assert len(rows) == 1 and rows[0][0] < 1e-3 and rows[0][2] == 5e-12 and t == 5e-12
    U1 = prob.get_state()
    assert np.all(np.isfinite(U1))
    assert np.abs(U1[:, :2] - U0[:, :2]).max() < 0.2       # log densities move a little per step
    assert prob.field_error(1) == pytest.approx(rows[0][0], rel=1e-12)


def test_colour_and_patch_assembly_agree(streamer_1m):
    """The two assembly kernels (global colouring / LDS patches) give the same F and J."""
    prob, _ = streamer_1m
    prob.set_step(5e-12, 5e-12)
    x = np.random.default_rng(2).normal(size=prob.n)
    # a state that differs from the old ones: with u == u_old the BDF term is cancellation noise
    # (eps*|u|*n/dt) and two correct evaluations of it agree to 1e-8 of |F| only
    U = prob.get_state()
    Up = U.copy()
    Up[:, :2] += 0.01 * np.random.default_rng(5).normal(size=(U.shape[0], 2))
    prob.set_state(Up, U, U)
    out = {}
    for kind in ("colour", "patch"):
        prob.set_assembly(kind)
        F, _ = prob.residual()
        prob.jacobian()
        out[kind] = (F, prob.spmv(x))
    for a, b in zip(out["colour"], out["patch"]):
        assert np.abs(a - b).max() / np.abs(a).max() < 1e-12
    # the coloured kernel is bitwise reproducible
    prob.set_assembly("colour")
    F2, _ = prob.residual()
    assert np.array_equal(F2, out["colour"][0])
    prob.set_assembly("patch")
    prob.set_state(U, U, U)


def test_time_of_flight_full_mesh():
    """examples/time_of_flight/fedm-tof.py mesh (160 x 320, 51 681 DOFs): 20 steps stay
    within 2 % of the analytic pulse in the resolved region and conserve the BDF2 order
    start-up (two BDF1 steps)."""
    from fedm_amd.cases import time_of_flight as tof
    prob, mesh = tof.device_problem(160, 320, 5e-4, 1e-3)
    assert prob.n == 51681
    t0, dt = 2.5e-9, 1e-12
    x = mesh.coords
    prob.set_state(tof.analytic_log_density(x, t0, tof.DOLFIN_EPS),
                   tof.analytic_log_density(x, t0), tof.analytic_log_density(x, t0))
    nodes = tof.p2_nodes(mesh.coords, mesh.cells)
    t, dt_old = t0, 1e30
    for k in range(20):
        prob.shift_state()
        t += dt
        prob.set_ext_source(0, tof.source(nodes, t))
        prob.set_step(dt, dt_old)
        its, _ = prob.newton_solve(rtol=1e-10, max_it=50)
        assert its <= 50
        if t > t0 + dt:
            dt_old = dt
    u = prob.get_state()[:, 0]
    ua = tof.analytic_log_density(x, t)
    core = ua > ua.max() - 10.0
    assert np.abs(np.exp(u[core] - ua[core]) - 1.0).max() < 0.02


# ---- BASELINE configs[4]: the ~4 M-DOF mesh --------------------------------------------------------
def test_streamer_at_4m_dofs_jacobian_product_and_step():
    """configs[4]'s mesh (1152 x 1152, 3 988 227 DOFs; Jacobian 670 MB, beyond the Infinity Cache) on ONE
    GPU: (F(u + e v) - F(u - e v)) / 2e == J v on every row, the product is linear, and two adaptive
    steps are accepted with the iteration counts of the 1 M-DOF case."""
    from fedm_amd.cases import streamer
    msh = streamer.mesh(1152, 4.0)
    prob = streamer.device_problem(msh.coords, msh.cells)
    assert prob.n == 3988227
    st = streamer.Stepper(prob)
    st.initialise()
    U = prob.get_state()
    prob.set_step(5e-12, 1e30)
    rng = np.random.default_rng(0)
    v = rng.normal(size=U.shape) * np.array([1e-3, 1e-3, 1e-1])
    z = prob.coords[:, 1]
    v[(np.abs(z) < 3e-16) | (np.abs(z - 0.0125) < 3e-16), 2] = 0.0
    prob.jacobian()
    Jv = prob.spmv(v.ravel())
    e = 1e-3
    prob.set_state(u_new=U + e * v)
    Fp, _ = prob.residual()
    prob.set_state(u_new=U - e * v)
    Fm, _ = prob.residual()
    prob.set_state(U, U, U)
    fd = (Fp - Fm) / (2 * e)
    for c in range(3):
        a, b = fd.reshape(-1, 3)[:, c], Jv.reshape(-1, 3)[:, c]
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-6, c
    x, y = rng.normal(size=prob.n), rng.normal(size=prob.n)
    assert np.abs(prob.spmv(2.5 * x - 0.75 * y) - (2.5 * prob.spmv(x) - 0.75 * prob.spmv(y))).max() < 1e-12 * np.abs(prob.spmv(x)).max()
    for _ in range(2):
        st.step()
    rows = st.log_rows()
    assert len(rows) == 2 and all(r[0] < 1e-3 and r[2] == 5e-12 for r in rows)
    assert st.newton_iterations <= 8 and st.linear_iterations <= 16
    prob.close()


def _configs4_worker(rank, world, port, q):
    import os
    import sys
    from pathlib import Path
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import torch.distributed as dist
    from fedm_amd.cases import streamer_distributed
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run = streamer_distributed.Runner(None, rank, world, 0, grading=4.0, transport="torch", global_n=1152)
        run.initialise()
        for _ in range(2):
            run.step()
        stats = run.prob.comm_stats()
        q.put((rank, run.log_rows(), run.total_dofs, int(run.lm.n_owned), int(run.lm.n_ghost), run.linear_iterations,
               stats["halo_exchanges"], run.halo_depth))
    finally:
        dist.destroy_process_group()


def test_configs4_mesh_split_over_two_ranks_matches_one_gpu():
    """The 4 M-DOF mesh of configs[4] partitioned (two parts here: the pool's boxes have one GPU, so both
    ranks share it over the host-staged transport; the driver's 8-GPU run uses the same code over RCCL):
    deep halos of eight layers, distributed multigrid -- the same error-log rows as the single-GPU run."""
    import socket
    import torch.multiprocessing as mp
    from fedm_amd.cases import streamer
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_configs4_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = mp_results.collect(procs, q, len(procs), 900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][2] == 3988227 and sum(r[3] for r in res) == 1153 * 1153
    assert all(r[7] == 8 and 8 * 1153 * 0.9 < r[4] < 8 * 1153 * 1.2 for r in res)      # eight ghost layers along the cut
    msh = streamer.mesh(1152, 4.0)
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob)
    st.initialise()
    for _ in range(2):
        st.step()
    ref = np.array(st.log_rows())
    prob.close()
    for r in res:
        assert np.allclose(np.array(r[1]), ref, rtol=2e-4)        # Newton to rtol 1e-4 on both sides
        assert r[5] <= st.linear_iterations + 4
        assert r[6] < 4 * r[5] + 40                                # about 1.5 exchanges per Krylov step + set-up


# ---- the ~1 M-DOF locally refined unstructured mesh (the bench's second record, the 14 ns run) ----------
def test_unstructured_mesh_at_1m_dofs_properties(tmp_path):
    """The refined Delaunay mesh of bench.py's `unstructured` record (4 um spacing in the channel, 341 280
    vertices = 1 023 840 DOFs) through the DOLFIN XML reader: pattern figures, the initial potential solves
    Poisson's equation, (F(u + e v) - F(u - e v)) / 2e == J v on every row, the product is linear, the
    coloured and the patch assembly agree, and two adaptive steps are accepted."""
    from fedm_amd.cases import streamer
    msh = streamer.refined_mesh(4e-6, growth=0.1, channel=(0.0, 4e-4) + streamer.CHANNEL[2:], xml_path=tmp_path / "mesh.xml")
    prob = streamer.device_problem(msh.coords, msh.cells)
    assert prob.n == 1023840
    sz = prob.sizes()
    assert sz["assembly_variant"] == "lds-patches/one-pass" and sz["max_patch_cells"] <= 256
    assert sz["stored_blocks"] / sz["nnz_blocks"] < 1.05 and sz["cell_visits"] < 1.3 * sz["n_cells"]
    st = streamer.Stepper(prob)
    st.initialise()
    U = prob.get_state()
    z = prob.coords[:, 1]
    assert np.all(U[np.abs(z) < 3e-16, 2] == 0.0) and np.all(U[np.abs(z - streamer.BOX) < 3e-16, 2] == streamer.U_W)
    prob.set_step(5e-12, 1e30)
    F, _ = prob.residual()
    assert np.abs(F.reshape(-1, 3)[:, 2]).max() / (2 * np.pi * streamer.BOX * streamer.U_W) < 1e-9
    rng = np.random.default_rng(0)
    v = rng.normal(size=U.shape) * np.array([1e-3, 1e-3, 1e-1])
    v[(np.abs(z) < 3e-16) | (np.abs(z - streamer.BOX) < 3e-16), 2] = 0.0
    prob.jacobian()
    Jv = prob.spmv(v.ravel())
    e = 1e-3
    prob.set_state(u_new=U + e * v)
    Fp, _ = prob.residual()
    prob.set_state(u_new=U - e * v)
    Fm, _ = prob.residual()
    fd = (Fp - Fm) / (2 * e)
    for c in range(3):
        a, b = fd.reshape(-1, 3)[:, c], Jv.reshape(-1, 3)[:, c]
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-6, c
    x, y = rng.normal(size=prob.n), rng.normal(size=prob.n)
    lhs, rhs = prob.spmv(2.5 * x - 0.75 * y), 2.5 * prob.spmv(x) - 0.75 * prob.spmv(y)
    assert np.abs(lhs - rhs).max() / np.abs(rhs).max() < 1e-13
    Up = U.copy()
    Up[:, :2] += 0.01 * rng.normal(size=(U.shape[0], 2))
    prob.set_state(Up, U, U)
    prob.set_step(5e-12, 5e-12)
    out = {}
    for kind in ("colour", "patch"):
        prob.set_assembly(kind)
        Fk, _ = prob.residual()
        prob.jacobian()
        out[kind] = (Fk, prob.spmv(x))
    for a, b in zip(out["colour"], out["patch"]):
        assert np.abs(a - b).max() / np.abs(a).max() < 1e-12
    prob.set_state(U, U, U)
    for _ in range(2):
        st.step()
    rows = st.log_rows()
    assert len(rows) == 2 and all(r[0] < 1e-3 and r[2] == 5e-12 for r in rows)
    assert st.newton_iterations <= 6 and st.linear_iterations <= 14
    prob.close()
