"""Multi-GPU decomposition on CPU: partition, halo plan, gloo ghost exchange, and the
owned-row assembly property (local oracle residual rows == global oracle residual rows)."""
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


def test_rcb_partition_is_balanced_and_complete():
    from fedm_amd import partition
    from fedm_amd.cases import streamer
    m = streamer.mesh(24, 3.0)
    for k in (2, 3, 4, 8):
        part = partition.partition_rcb(m.coords, k)
        cnt = np.bincount(part, minlength=k)
        assert cnt.sum() == m.num_vertices() and cnt.min() > 0
        assert cnt.max() - cnt.min() <= k
        lms = [partition.local_mesh(m.coords, m.cells, part, r) for r in range(k)]
        owned = np.concatenate([lm.vertex_global[:lm.n_owned] for lm in lms])
        assert np.array_equal(np.sort(owned), np.arange(m.num_vertices()))
        # every cell lives on the ranks owning one of its vertices
        for lm in lms:
            assert set(lm.neighbours.tolist()) <= set(range(k)) - {lm.rank}
        # halo plans are pairwise consistent: what p sends to q is what q expects from p
        for p in range(k):
            for ip, q in enumerate(lms[p].neighbours):
                lq = lms[q]
                iq = list(lq.neighbours).index(p)
                sent = lms[p].vertex_global[lms[p].send_idx[lms[p].send_ptr[ip]:lms[p].send_ptr[ip + 1]]]
                expected = lq.vertex_global[lq.n_owned + lq.recv_ptr[iq]:lq.n_owned + lq.recv_ptr[iq + 1]]
                assert np.array_equal(sent, expected)


def test_owned_rows_assemble_locally():
    """Assembling the cells that touch owned vertices reproduces the global residual rows."""
    from fedm_amd import partition
    from oracle import streamer as ost
    from oracle.mesh import Mesh, graded_axis, rectangle_right, mark_boundaries
    from oracle.forms import LFAModel
    n = 12
    gm = rectangle_right(0.0, 0.0, ost.BOX, ost.BOX, n, n, xs=graded_axis(ost.BOX, n, 3.0))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gmodel = ost.build(gm)
        U = ost.initial_state(gmodel)
    rng = np.random.default_rng(0)
    U[:, 1] += rng.normal(0, 0.2, gm.nv)
    Uo = U + rng.normal(0, 0.01, U.shape)
    F_glob = gmodel.residual(U, Uo, Uo, 5e-12, 4e-12, apply_bc=False).reshape(gm.nv, 3)
    part = partition.partition_rcb(gm.coords, 3)
    gtags = mark_boundaries(gm, ost.BOUNDARIES)
    for r in range(3):
        lm = partition.local_mesh(gm.coords, gm.cells, part, r)
        lmesh = Mesh(lm.coords, lm.cells)
        lmodel = LFAModel(lmesh, 2, True, ["reaction", "drift-diffusion-reaction"], [1.0, -1.0],
                          mu=[0.0, ost.MU_E], D=[0.0, ost.D_E],
                          reactions=[(ost.K_ION, [0, 1], [1, 1])],
                          facet_tags=gtags[lm.cell_global], bc_type=ost.BC_TYPE, qdeg=2)
        g = lm.vertex_global
        F_loc = lmodel.residual(U[g], Uo[g], Uo[g], 5e-12, 4e-12, apply_bc=False).reshape(-1, 3)
        own = slice(0, lm.n_owned)
        scale = np.abs(F_glob).max(axis=0)
        assert (np.abs(F_loc[own] - F_glob[g[own]]) / scale).max() < 1e-13


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
        m = streamer.mesh(16, 2.0)
        part = partition.partition_rcb(m.coords, world)
        lm = partition.local_mesh(m.coords, m.cells, part, rank)
        f = lambda g: np.stack([np.sin(g * 0.37), g.astype(float), -2.0 * g], axis=1)
        vals = np.zeros((lm.coords.shape[0], 3))
        vals[:lm.n_owned] = f(lm.vertex_global[:lm.n_owned])
        vals = partition.exchange_ghosts(lm, vals)
        ok = np.array_equal(vals, f(lm.vertex_global))
        q.put((rank, bool(ok), int(lm.n_ghost)))
    finally:
        dist.destroy_process_group()


def test_gloo_ghost_exchange_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = mp_results.collect(procs, q, len(procs), 120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert all(ng > 0 for _, _, ng in res)


def test_deep_halo_layers_plans_and_redundantly_assembled_rows():
    """partition.local_mesh(depth=k): k ghost layers by edge distance; the halo plans of all ranks fit
    together; the outermost layer is the list of identity rows; and the cells a rank keeps are exactly
    what it needs to assemble the rows of its owned vertices AND of its ghost vertices of the inner
    layers -- the oracle's residual rows of those vertices on the local mesh equal the global ones
    (that is what lets a Krylov step run sweeps, smoothings and product after ONE exchange)."""
    from fedm_amd import partition
    from fedm_amd.cases import streamer
    from oracle import streamer as ost
    from oracle.forms import LFAModel
    from oracle.mesh import Mesh, mark_boundaries
    m = streamer.refined_mesh(2.5e-4)                       # unstructured, ~700 vertices
    nv = m.num_vertices()
    graph = partition.vertex_graph(nv, m.cells)
    depth, world = 4, 3
    part = partition.partition_rcb(m.coords, world)
    lms = [partition.local_mesh(m.coords, m.cells, part, r, depth=depth) for r in range(world)]
    # layers against a brute-force breadth-first search
    import scipy.sparse.csgraph as csg
    for lm in lms:
        owned = np.nonzero(part == lm.rank)[0]
        d = csg.shortest_path(graph.astype(float), unweighted=True, indices=owned).min(axis=0)
        inside = np.nonzero(d <= depth)[0]
        assert np.array_equal(np.sort(lm.vertex_global), inside)
        assert np.array_equal(lm.layer, d[lm.vertex_global].astype(int))
        assert np.array_equal(lm.identity_vertices, np.nonzero(lm.layer == depth)[0])
        assert (lm.layer[:lm.n_owned] == 0).all() and (lm.layer[lm.n_owned:] > 0).all()
    for p in range(world):
        for ip_, q in enumerate(lms[p].neighbours):
            lq = lms[q]
            iq = list(lq.neighbours).index(p)
            sent = lms[p].vertex_global[lms[p].send_idx[lms[p].send_ptr[ip_]:lms[p].send_ptr[ip_ + 1]]]
            expected = lq.vertex_global[lq.n_owned + lq.recv_ptr[iq]:lq.n_owned + lq.recv_ptr[iq + 1]]
            assert np.array_equal(sent, expected)
    # rows of the owned vertices and of the ghost layers 1 .. depth-1 assemble locally
    gm = Mesh(m.coords, m.cells)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gmodel = ost.build(gm)
        U = ost.initial_state(gmodel)
    rng = np.random.default_rng(1)
    U[:, 1] += rng.normal(0, 0.2, nv)
    Uo = U + rng.normal(0, 0.01, U.shape)
    F_glob = gmodel.residual(U, Uo, Uo, 5e-12, 4e-12, apply_bc=False).reshape(nv, 3)
    gtags = mark_boundaries(gm, ost.BOUNDARIES)
    scale = np.abs(F_glob).max(axis=0)
    for lm in lms:
        lmodel = LFAModel(Mesh(lm.coords, lm.cells), 2, True, ["reaction", "drift-diffusion-reaction"], [1.0, -1.0],
                          mu=[0.0, ost.MU_E], D=[0.0, ost.D_E], reactions=[(ost.K_ION, [0, 1], [1, 1])],
                          facet_tags=gtags[lm.cell_global], bc_type=ost.BC_TYPE, qdeg=2)
        g = lm.vertex_global
        F_loc = lmodel.residual(U[g], Uo[g], Uo[g], 5e-12, 4e-12, apply_bc=False).reshape(-1, 3)
        rows = lm.layer < depth
        assert rows.sum() > lm.n_owned
        assert (np.abs(F_loc[rows] - F_glob[g[rows]]) / scale).max() < 1e-12
        # ... and NOT those of the outermost layer (their cells beyond the local mesh are missing)
        outer = lm.layer == depth
        assert (np.abs(F_loc[outer] - F_glob[g[outer]]) / scale).max() > 1e-6


def _u_shaped_domain(n=60):
    """a 3 x 3 box with a slot cut out of it: the median planes of a coordinate bisection cross both arms"""
    xs = np.linspace(0.0, 3.0, n + 1)
    X, Y = np.meshgrid(xs, xs, indexing="ij")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    i, j = i.ravel(), j.ravel()
    cx, cy = xs[i] + 1.5 / n, xs[j] + 1.5 / n
    keep = ~((cx > 1.0) & (cx < 2.0) & (cy > 0.5))
    i, j = i[keep], j[keep]
    a, b, c, d = i * (n + 1) + j, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1, i * (n + 1) + j + 1
    cells = np.concatenate([np.stack([a, b, c], axis=1), np.stack([a, c, d], axis=1)])
    used = np.unique(cells)
    lookup = np.full(coords.shape[0], -1)
    lookup[used] = np.arange(used.size)
    return coords[used], lookup[cells]


def test_graph_partitioner_balances_and_never_cuts_more_than_the_planes():
    """`north_star` names a graph partitioner (METIS); `fedm_amd/graph_partition.py` is the multilevel scheme written
    out: balanced, deterministic, complete, and on every mesh tried its edge cut is at most the coordinate
    bisection's (each bisection keeps the better of the multilevel split and the refined median plane)."""
    from fedm_amd import partition, graph_partition as gp
    from fedm_amd.cases import streamer
    m = streamer.mesh(40, 3.0)
    r = streamer.refined_mesh(4e-5, growth=0.15, channel=(0.0, 100.0 * 4e-5) + streamer.CHANNEL[2:])
    cases = [(m.coords, m.cells), (r.coords, r.cells), _u_shaped_domain()]
    for coords, cells in cases:
        G = partition.vertex_graph(coords.shape[0], cells).astype(np.float64).tocsr()
        for k in (2, 3, 8):
            part = gp.partition_graph(coords, cells, k)
            assert np.array_equal(part, gp.partition_graph(coords, cells, k))          # every rank computes the same
            cnt = np.bincount(part, minlength=k)
            assert cnt.sum() == coords.shape[0] and cnt.min() > 0
            assert cnt.max() <= 1.03 * cnt.mean()
            assert gp.edge_cut(G, part) <= gp.edge_cut(G, partition.partition_rcb(coords, k, cells)) * 1.0001
    # where straight cuts are bad it does better: two parts of the U-shaped domain
    coords, cells = _u_shaped_domain()
    G = partition.vertex_graph(coords.shape[0], cells).astype(np.float64).tocsr()
    assert gp.edge_cut(G, gp.partition_graph(coords, cells, 2)) < gp.edge_cut(G, partition.partition_rcb(coords, 2, cells))
    # the halo plans built on a graph partition are pairwise consistent (deep halos too)
    part = gp.partition_graph(coords, cells, 4)
    for depth in (1, 3):
        lms = [partition.local_mesh(coords, cells, part, q, depth=depth) for q in range(4)]
        for p in range(4):
            for ip, q in enumerate(lms[p].neighbours):
                lq = lms[q]
                iq = list(lq.neighbours).index(p)
                sent = lms[p].vertex_global[lms[p].send_idx[lms[p].send_ptr[ip]:lms[p].send_ptr[ip + 1]]]
                expected = lq.vertex_global[lq.n_owned + lq.recv_ptr[iq]:lq.n_owned + lq.recv_ptr[iq + 1]]
                assert np.array_equal(sent, expected)


def test_matching_and_refinement_steps_of_the_graph_partitioner():
    """The pieces: heavy-edge matching halves the graph and conserves vertex and edge weight; a refinement pass never
    raises the cut and keeps the balance it is given."""
    from fedm_amd import partition, graph_partition as gp
    coords, cells = _u_shaped_domain(40)
    G = partition.vertex_graph(coords.shape[0], cells).astype(np.float64).tocsr()
    G.sort_indices()
    vw = np.ones(coords.shape[0])
    Ac, vc, xc, cmap = gp._coarsen(G, vw, coords)
    assert 0.5 * G.shape[0] <= Ac.shape[0] <= 0.62 * G.shape[0]
    assert vc.sum() == vw.sum() and np.bincount(cmap).max() <= 2
    inner = G.data[cmap[gp._rows(G)] == cmap[G.indices]].sum()
    assert abs(Ac.sum() + inner - G.sum()) < 1e-9
    rng = np.random.default_rng(3)
    side = (coords[:, 0] + 0.05 * rng.standard_normal(coords.shape[0]) > 1.5).astype(np.int8)     # a ragged cut
    target0 = float(np.count_nonzero(side == 0))
    out = gp._refine(G, vw, side, target0, tol=10.0)
    assert gp.edge_cut(G, out) < 0.7 * gp.edge_cut(G, side)
    assert abs(np.count_nonzero(out == 0) - target0) <= 10
