"""Locally refined unstructured meshes on the host side (no GPU): the generator that stands in for the
reference's missing ``mesh.xml`` (examples/streamer_discharge/fedm-streamer.py:116,
.MISSING_LARGE_BLOBS:2), its way through the DOLFIN XML reader, the device path's vertex ordering
and pattern on it, and the multi-GPU decomposition (2-rank gloo)."""
import os
import socket
import sys
import warnings
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent))
import mp_results  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
H_FINE = 30e-6          # ~17 k vertices


@pytest.fixture(scope="module")
def refined(tmp_path_factory):
    from fedm_amd.cases import streamer
    return streamer.refined_mesh(H_FINE, xml_path=tmp_path_factory.mktemp("mesh") / "mesh.xml")


def _valence(mesh):
    c = mesh.cells
    e = np.unique(np.sort(np.concatenate([c[:, [0, 1]], c[:, [1, 2]], c[:, [2, 0]]]), axis=1), axis=0)
    return np.bincount(e.ravel(), minlength=mesh.num_vertices())


def test_generator_is_deterministic_graded_and_of_good_quality(refined):
    from fedm_amd import meshgen
    from fedm_amd.cases import streamer
    again = streamer.refined_mesh(H_FINE)
    assert np.array_equal(again.coords, refined.coords) and np.array_equal(again.cells, refined.cells)
    q = meshgen.mesh_quality(refined)
    assert q["n_vertices"] >= 16000
    assert q["min_angle"] > 25.0
    assert q["hmax"] / q["hmin"] > 20.0                    # locally refined, not quasi-uniform
    # genuinely unstructured: interior vertices of valence 5, 6, 7 and more side by side
    x = refined.coords
    interior = (x[:, 0] > 0) & (x[:, 0] < streamer.BOX) & (x[:, 1] > 0) & (x[:, 1] < streamer.BOX)
    val = _valence(refined)
    assert val[interior].min() >= 4 and val[interior].max() <= 10
    assert len(np.unique(val[interior])) >= 4
    # positively oriented cells covering the box exactly once
    a, b, c = (x[refined.cells[:, k]] for k in range(3))
    det = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    assert det.min() > 0 and 0.5 * det.sum() == pytest.approx(streamer.BOX ** 2, rel=1e-12)
    # the numbering is not a sweep through space: consecutive vertices are far apart somewhere
    assert np.abs(np.diff(x[:, 1])).max() > 0.5 * streamer.BOX


def test_sides_are_exact_so_the_scripts_boundary_tests_apply(refined):
    """Marking_boundaries and the Dirichlet `near()` tests compare with DOLFIN_EPS
    (fedm/functions.py:73-124, fedm-streamer.py:186-200)."""
    from fedm_amd.cases import streamer
    from fedm_amd.mesh import Marking_boundaries
    tags = Marking_boundaries(refined, streamer.BOUNDARIES)
    cell, local = refined.exterior_facets()
    assert (tags[cell, local] > 0).all()                   # every exterior facet lies on one of the four lines
    assert set(np.unique(tags[cell, local])) == {1, 2, 3, 4}
    dofs, vals = streamer.dirichlet(refined.coords)
    assert (vals == 0).sum() > 10 and (vals == streamer.U_W).sum() > 10


def test_vertex_order_makes_compact_patches_on_the_unstructured_mesh(refined):
    from fedm_amd import device
    from fedm_amd.cases import streamer
    nv = refined.num_vertices()
    z = device.z_curve_order(refined.coords)
    kd = device.bisection_order(refined.coords, device._vertex_spacing(refined.coords, refined.cells))
    for o in (z, kd):
        assert np.array_equal(np.sort(o), np.arange(nv))
    vz, vk = (device.cell_visits(o, refined.cells, nv) for o in (z, kd))
    assert vk < vz                                          # the bisection wins here ...
    assert np.array_equal(device.locality_order(refined.coords, refined.cells), kd)
    st = device.pattern_stats(refined.coords, refined.cells)
    assert st["cell_visits"] == vk
    assert st["cell_visits"] < 1.3 * refined.num_cells()
    assert st["max_patch_cells"] <= 192                     # one cell per thread of the 192-thread patch kernel
    assert st["stored_blocks"] < 1.08 * st["nnz_blocks"]    # sliced block-ELL padding
    assert st["bank_clashes"] < 0.15 * st["owned_pairs"]
    # ... and the Z-curve on the tensor-product mesh of the headline bench
    tp = streamer.mesh(96, 4.0)
    assert np.array_equal(device.locality_order(tp.coords, tp.cells), device.z_curve_order(tp.coords))


def test_owned_rows_assemble_locally_on_the_unstructured_mesh(refined):
    from fedm_amd import partition
    from oracle import streamer as ost
    from oracle.forms import LFAModel
    from oracle.mesh import Mesh, mark_boundaries
    gm = Mesh(refined.coords, refined.cells)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gmodel = ost.build(gm)
        U = ost.initial_state(gmodel)
    rng = np.random.default_rng(0)
    U[:, 1] += rng.normal(0, 0.2, gm.nv)
    Uo = U + rng.normal(0, 0.01, U.shape)
    F_glob = gmodel.residual(U, Uo, Uo, 5e-12, 4e-12, apply_bc=False).reshape(gm.nv, 3)
    part = partition.partition_rcb(gm.coords, 4)
    cnt = np.bincount(part)
    assert cnt.max() - cnt.min() <= 4
    gtags = mark_boundaries(gm, ost.BOUNDARIES)
    scale = np.abs(F_glob).max(axis=0)
    for r in range(4):
        lm = partition.local_mesh(gm.coords, gm.cells, part, r)
        lmodel = LFAModel(Mesh(lm.coords, lm.cells), 2, True, ["reaction", "drift-diffusion-reaction"],
                          [1.0, -1.0], mu=[0.0, ost.MU_E], D=[0.0, ost.D_E],
                          reactions=[(ost.K_ION, [0, 1], [1, 1])],
                          facet_tags=gtags[lm.cell_global], bc_type=ost.BC_TYPE, qdeg=2)
        g = lm.vertex_global
        F_loc = lmodel.residual(U[g], Uo[g], Uo[g], 5e-12, 4e-12, apply_bc=False).reshape(-1, 3)
        own = slice(0, lm.n_owned)
        assert (np.abs(F_loc[own] - F_glob[g[own]]) / scale).max() < 1e-12


def test_bisection_that_counts_severed_edges_on_the_refined_mesh(refined):
    """RCB by physical extent cuts the long, thin refined channel along its length; with the mesh at
    hand each bisection takes the direction whose cut severs fewer edges: several times smaller edge
    cut and deep halos at the same (perfect) balance.  On a tensor-product mesh the two agree."""
    from fedm_amd import partition
    from fedm_amd.cases import streamer
    graph = partition.vertex_graph(refined.num_vertices(), refined.cells).tocoo()
    for k in (2, 4, 8):
        plain = partition.partition_rcb(refined.coords, k)
        aware = partition.partition_rcb(refined.coords, k, refined.cells)
        cut = lambda part: int((part[graph.row] != part[graph.col]).sum() // 2)
        assert np.bincount(aware).max() - np.bincount(aware).min() <= k
        assert cut(aware) < 0.5 * cut(plain)
        ghosts = lambda part: max(partition.local_mesh(refined.coords, refined.cells, part, r, depth=4).n_ghost
                                  for r in range(k))
        assert ghosts(aware) < 0.6 * ghosts(plain)
    tp = streamer.mesh(48, 4.0)
    g2 = partition.vertex_graph(tp.num_vertices(), tp.cells).tocoo()
    a, b = partition.partition_rcb(tp.coords, 4), partition.partition_rcb(tp.coords, 4, tp.cells)
    assert abs(int((a[g2.row] != a[g2.col]).sum()) - int((b[g2.row] != b[g2.col]).sum())) <= 0.05 * (a[g2.row] != a[g2.col]).sum()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from fedm_amd import partition
    from fedm_amd.cases import streamer
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = streamer.refined_mesh(60e-6)
        part = partition.partition_rcb(m.coords, world)
        lm = partition.local_mesh(m.coords, m.cells, part, rank)
        f = lambda g: np.stack([np.sin(g * 0.37), g.astype(float), -2.0 * g], axis=1)
        vals = np.zeros((lm.coords.shape[0], 3))
        vals[:lm.n_owned] = f(lm.vertex_global[:lm.n_owned])
        vals = partition.exchange_ghosts(lm, vals)
        q.put((rank, bool(np.array_equal(vals, f(lm.vertex_global))), int(lm.n_ghost), int(lm.n_owned)))
    finally:
        dist.destroy_process_group()


def test_gloo_ghost_exchange_world2_on_the_unstructured_mesh():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = mp_results.collect(procs, q, len(procs), 180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res)
    assert all(ng > 0 for _, _, ng, _ in res)
    assert abs(res[0][3] - res[1][3]) <= 1


def test_xml_reader_fast_path_and_parser_fallback_agree(tmp_path):
    """mesh_io.read_dolfin_xml scans files in DOLFIN's own attribute order with regular expressions and hands
    anything else to the XML parser: both give the mesh that was written."""
    from fedm_amd import mesh_io
    from fedm_amd.cases import streamer
    msh = streamer.refined_mesh(2e-4)
    mesh_io.write_dolfin_xml(msh, tmp_path / "a.xml")
    a = mesh_io.read_dolfin_xml(tmp_path / "a.xml")
    text = (tmp_path / "a.xml").read_text()
    # the same file with the vertex attributes in another order: not DOLFIN's layout -> parser path
    import re
    shuffled = re.sub(r'<vertex index="(\d+)" x="([^"]+)" y="([^"]+)"', r'<vertex y="\3" index="\1" x="\2"', text)
    assert shuffled != text
    (tmp_path / "b.xml").write_text(shuffled)
    b = mesh_io.read_dolfin_xml(tmp_path / "b.xml")
    for m in (a, b):
        assert np.array_equal(m.coords, msh.coords) and np.array_equal(m.cells, msh.cells)
    (tmp_path / "c.xml").write_text(text.replace('celltype="triangle"', 'celltype="tetrahedron"'))
    with pytest.raises(ValueError, match="not a DOLFIN XML triangle mesh"):
        mesh_io.read_dolfin_xml(tmp_path / "c.xml")


@pytest.mark.parametrize("n", [5, 64, 65, 1000])
def test_vertex_orders_are_permutations_for_any_size(n):
    """Fewer vertices than one slice, exactly one, one more, and a count that is no multiple of 64."""
    from fedm_amd import device
    rng = np.random.default_rng(n)
    pts = rng.random((n, 2))
    from scipy.spatial import Delaunay
    cells = Delaunay(pts).simplices.astype(np.int32)
    for order in (device.z_curve_order(pts), device.bisection_order(pts, device._vertex_spacing(pts, cells)),
                  device.locality_order(pts, cells), device.locality_order(pts)):
        assert np.array_equal(np.sort(order), np.arange(n))
    st = device.pattern_stats(pts, cells)
    assert st["n_slices"] == (n + 63) // 64 and st["nnz_blocks"] > n


@pytest.mark.parametrize("kind", ["tensor", "refined"])
@pytest.mark.parametrize("slices,layers", [(8, 3), (8, 5), (4, 2), (1, 1)])
def test_tiles_of_the_species_sweeps_are_consistent(kind, slices, layers, refined):
    """The host-built tables behind the tiled species sweeps (fedm_amd/csrc/fs_tiles.hip: vertex lists
    [tile | layer 1 | ...], 16-bit local column numbers) checked by the library itself, without a GPU:
    every vertex owned by exactly one tile, nested layers, every local column naming the pattern's vertex."""
    from fedm_amd import device
    from fedm_amd.cases import streamer
    msh = streamer.mesh(96, 4.0) if kind == "tensor" else refined
    st = device.fieldsplit_tiles_stats(msh.coords, msh.cells, slices, layers)
    nv = msh.num_vertices()
    assert st["violations"] == 0
    assert st["n_tiles"] == -(-(-(-nv // 64)) // slices)
    assert nv <= st["total_rows"] <= st["total_vertices"]
    assert st["max_rows"] <= st["max_vertices"] < 65536
    assert st["row_width"] == device.pattern_stats(msh.coords, msh.cells)["max_patch_width"]
    if (slices, layers) == (8, 3):
        # what the default costs: the rows of the layers are computed redundantly by every tile that needs them
        assert st["total_rows"] < (1.6 if kind == "tensor" else 2.0) * nv
        assert st["max_rows"] <= 3 * 512          # three rows a thread at most (the kernel's instantiations)


def test_tile_parameters_out_of_range_are_refused(refined):
    from fedm_amd import device
    for slices, layers in ((0, 3), (9, 3), (8, 0), (8, 9)):
        with pytest.raises(RuntimeError, match="tile parameters out of range"):
            device.fieldsplit_tiles_stats(refined.coords, refined.cells, slices, layers)
