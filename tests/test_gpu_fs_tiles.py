"""The species sweeps of the field split, several per launch on tiles of slices with their vertex layers in LDS
(fedm_amd/csrc/fs_tiles.hip), against the one-launch-per-sweep kernels they replace on one GPU: same operands,
same order of the sums -- the preconditioner's output must agree BIT FOR BIT, on the tensor-product mesh and on
the locally refined unstructured one, for every chunking of the sweeps into launches.

The field split itself stands for the sub-solvers PETSc would run on the blocks of the reference's Newton
systems (fedm-streamer.py:293-300 selects the solver; tests/integrated_tests/.../fedm_streamer.py:32 GMRES)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mesh(kind):
    from fedm_amd.cases import streamer
    return streamer.mesh(96, 4.0) if kind == "tensor" else streamer.refined_mesh(30e-6)


def _problem(msh, weights):
    """A streamer context with a developed state, its Jacobian assembled, the multigrid and the species
    polynomial installed."""
    from fedm_amd.cases import streamer
    prob = streamer.device_problem(msh.coords, msh.cells)
    r, z = msh.coords[:, 0], msh.coords[:, 1]
    rng = np.random.default_rng(5)
    head = np.exp(-(r ** 2 + (z - 0.008) ** 2) / (0.6e-3) ** 2)
    U = np.zeros((prob.nv, 3))
    U[:, 0] = np.log(1e13 + 4e19 * head) + 0.02 * rng.standard_normal(prob.nv)
    U[:, 1] = np.log(1e13 + 3e19 * head) + 0.02 * rng.standard_normal(prob.nv)
    U[:, 2] = streamer.U_W * z / streamer.BOX * (1.0 + 0.3 * head)
    prob.set_state(U, U + 0.01 * rng.standard_normal(U.shape), U)
    prob.set_step(5e-12, 4e-12)
    prob.setup_multigrid(**streamer.MULTIGRID)
    prob.set_fieldsplit(weights)
    prob.jacobian()
    return prob


@pytest.mark.parametrize("kind", ["tensor", "refined"])
@pytest.mark.parametrize("degree", [2, 4, 6, 8])
def test_tiled_sweeps_equal_the_sweeps_one_by_one(kind, degree):
    """One context, one assembled Jacobian (two assemblies differ in the last bits: the order of the LDS atomics
    is not fixed): the preconditioner with the sweeps one by one, then on tiles in three chunkings."""
    from fedm_amd.device import chebyshev_weights
    msh = _mesh(kind)
    prob = _problem(msh, chebyshev_weights(degree))
    t = np.random.default_rng(degree).standard_normal(3 * msh.num_vertices())
    prob.configure_fieldsplit_tiles(False)
    assert prob.fieldsplit_tiles() is None
    z_ref = prob.fieldsplit_apply(t)
    assert np.isfinite(z_ref).all() and np.abs(z_ref).max() > 0
    assert np.array_equal(prob.fieldsplit_apply(t), z_ref)        # (the comparison below is meaningful)
    for slices, layers, threads in ((0, 0, 0), (8, 2, 512), (4, 3, 128), (8, 5, 512), (2, 4, 64)):
        prob.configure_fieldsplit_tiles(True, slices, layers, threads, multigrid=False)
        info = prob.fieldsplit_tiles()
        assert info is not None and info["layers"] == (layers or 3) and info["slices_per_tile"] == (slices or 8)
        assert info["threads"] == (threads or 512) and info["max_rows"] <= 8 * info["threads"]
        # (the multigrid's finest-level sweeps stay kernels of their own here: see the test below)
        prob.configure_fieldsplit_tiles(True, slices, layers, threads, multigrid=False)
        z = prob.fieldsplit_apply(t)
        assert np.array_equal(z, z_ref), (slices, layers, threads, np.abs(z - z_ref).max())
    prob.close()


@pytest.mark.parametrize("kind", ["tensor", "refined"])
def test_polynomial_smoother_sweeps_on_the_same_tiles(kind):
    """The finest level's sweeps behind the polynomial smoother's product as one launch on the tiles, with the
    Jacobian's own potential-potential plane as the operator (double precision; the hierarchy's copy is its
    single-precision rounding): the species part of the preconditioner's output is untouched, the potential part agrees
    to the rounding of that copy.  (The V(1,1) cycle keeps its kernels: its output must not change at all.)"""
    from fedm_amd.device import chebyshev_weights
    msh = _mesh(kind)
    prob = _problem(msh, chebyshev_weights(6))
    t = np.random.default_rng(11).standard_normal(3 * msh.num_vertices())
    out = {}
    for cycle in ("V(1,1)", "polynomial smoother"):
        if cycle != "V(1,1)":
            prob.setup_multigrid(nu=1, omega=0.85, poly_degree=2)
            prob.jacobian()
        prob.configure_fieldsplit_tiles(True, multigrid=False)
        z_ref = prob.fieldsplit_apply(t).reshape(-1, 3)
        prob.configure_fieldsplit_tiles(True, multigrid=True)
        out[cycle] = (z_ref, prob.fieldsplit_apply(t).reshape(-1, 3))
    prob.close()
    z_ref, z = out["V(1,1)"]
    assert np.array_equal(z, z_ref)
    z_ref, z = out["polynomial smoother"]
    assert np.array_equal(z[:, :2], z_ref[:, :2])
    assert not np.array_equal(z[:, 2], z_ref[:, 2])          # (it did take the other path)
    assert np.abs(z[:, 2] - z_ref[:, 2]).max() < 1e-5 * np.abs(z_ref[:, 2]).max()


def test_tiled_sweeps_with_four_unknowns_per_vertex_equal_the_sweeps_one_by_one():
    """The glow-discharge case: the species block has four unknowns a vertex (energy + three species), a row's sixteen
    half-precision planes are eight registers an entry, a thread keeps one row (tiles of three slices).  Same operands,
    same order of the sums: bit for bit the sweeps one by one."""
    import contextlib, io
    from fedm_amd.cases import glow_discharge as gdc
    with contextlib.redirect_stdout(io.StringIO()):
        case = gdc.Case(nx=60, ny=60, T_final=1.0)
    for _ in range(3):
        case.step()
    prob = case.prob
    prob.jacobian()
    t = np.random.default_rng(3).standard_normal(prob.n)
    prob.configure_fieldsplit_tiles(False)
    assert prob.fieldsplit_tiles() is None
    z_ref = prob.fieldsplit_apply(t)
    assert np.isfinite(z_ref).all() and np.abs(z_ref).max() > 0
    assert np.array_equal(prob.fieldsplit_apply(t), z_ref)
    for slices, layers, threads in ((0, 0, 0), (2, 3, 512), (3, 2, 512), (1, 4, 256)):
        prob.configure_fieldsplit_tiles(True, slices, layers, threads, multigrid=False)
        info = prob.fieldsplit_tiles()
        assert info is not None and info["slices_per_tile"] == (slices or 3) and info["max_rows"] <= info["threads"]
        z = prob.fieldsplit_apply(t)
        assert np.array_equal(z, z_ref), (slices, layers, threads, np.abs(z - z_ref).max())
    prob.close()
