"""Multilevel graph partitioner for the multi-GPU path (the role SCOTCH/ParMETIS play behind DOLFIN's
``mpirun -np N``, README.md:63-67 of the reference; ``north_star``: "METIS-partitioned").

METIS is not in the image, so this is the textbook scheme written on numpy/scipy.sparse arrays: recursive bisection;
each bisection coarsens the vertex graph by heavy-edge matching (mutual proposals, three rounds a level) until a few
thousand vertices are left, bisects the coarsest graph along the direction (of eight) whose weighted median plane cuts
the least edge weight, and on the way back up refines every level with Fiduccia-Mattheyses-style passes in which all
movers of a pass move at once: a vertex moves when its gain (edge weight to the other side minus edge weight to its
own side) is positive and larger than that of every candidate neighbour on the OTHER side (two neighbours swapping
sides would both count the edge between them as a gain), the pass is cut back to what the balance allows, and passes
stop when the cut stops falling.  Everything is a deterministic function of (coords, cells, n_parts): every rank
computes the same partition without talking, as with `partition.partition_rcb`.

What is balanced is the number of vertices (the matrix rows a rank owns); what is minimised is the edge cut, the proxy
for the ghost layers a rank carries (`partition.local_mesh`)."""
import numpy as np
import scipy.sparse as sp

from .partition import vertex_graph

_COARSEST = 4000


def _rows(A):
    return np.repeat(np.arange(A.shape[0], dtype=np.int64), np.diff(A.indptr))


def _row_max(indptr, values, empty):
    """max of `values` (one per stored entry) over each CSR row; `empty` for rows without entries"""
    n = indptr.size - 1
    out = np.full(n, empty, dtype=values.dtype)
    lens = np.diff(indptr)
    has = lens > 0
    if values.size:
        out[has] = np.maximum.reduceat(values, indptr[:-1][has])
    return out


def _pair_key(i, j):
    """a symmetric pseudo-random 20-bit tie-break of the edge (i, j)"""
    lo, hi = np.minimum(i, j).astype(np.uint64), np.maximum(i, j).astype(np.uint64)
    h = (lo * np.uint64(2654435761) + hi * np.uint64(40503) + np.uint64(12345)) * np.uint64(0x9E3779B97F4A7C15)
    return (h >> np.uint64(44)).astype(np.int64)


def edge_cut(A, part):
    """total weight of the edges of the (symmetric) graph `A` whose ends lie in different parts"""
    rows = _rows(A)
    return float(A.data[part[rows] != part[A.indices]].sum()) / 2.0


def _coarsen(A, vw, xy):
    """one level of heavy-edge matching: (coarse graph, coarse vertex weights, coarse coordinates, map)"""
    n = A.shape[0]
    rows, cols = _rows(A), A.indices.astype(np.int64)
    key = (np.round(A.data).astype(np.int64) << 20) + _pair_key(rows, cols)
    match = np.full(n, -1, dtype=np.int64)
    ids = np.arange(n, dtype=np.int64)
    for _ in range(3):
        free = match < 0
        k = np.where(free[rows] & free[cols], key, -1)
        best = _row_max(A.indptr, k, -1)
        hit = (k == best[rows]) & (k >= 0)
        prop = np.full(n, -1, dtype=np.int64)
        prop[rows[hit]] = cols[hit]                    # (keys are unique per edge up to hash collisions; any hit will do)
        ok = prop >= 0
        mutual = ok.copy()
        mutual[ok] = prop[prop[ok]] == ids[ok]
        match[mutual] = prop[mutual]
    rep = np.where(match >= 0, np.minimum(ids, match), ids)
    _, cmap = np.unique(rep, return_inverse=True)
    nc = int(cmap.max()) + 1
    P = sp.csr_matrix((np.ones(n), (ids, cmap)), shape=(n, nc))
    Ac = (P.T @ A @ P).tocsr()
    Ac.setdiag(0.0)
    Ac.eliminate_zeros()
    Ac.sort_indices()
    vwc = np.bincount(cmap, weights=vw, minlength=nc)
    xyc = np.stack([np.bincount(cmap, weights=vw * xy[:, d], minlength=nc) / vwc for d in (0, 1)], axis=1)
    return Ac, vwc, xyc, cmap


def _gains(A, side, rows):
    same = side[rows] == side[A.indices]
    n = A.shape[0]
    ext = np.bincount(rows, weights=np.where(same, 0.0, A.data), minlength=n)
    inn = np.bincount(rows, weights=np.where(same, A.data, 0.0), minlength=n)
    return ext, inn, same


def _trim(movers, gain, vw, allowed):
    """the highest-gain prefix of `movers` whose weight stays within `allowed`"""
    if movers.size == 0 or allowed <= 0:
        return movers[:0]
    order = movers[np.argsort(-gain[movers], kind="stable")]
    keep = np.cumsum(vw[order]) <= allowed
    return order[keep]


def _rebalance(A, vw, side, target0, tol, rows):
    """moves boundary vertices of the heavy side (best gain first) until side 0 weighs target0 +- tol"""
    for _ in range(8):
        w0 = float(vw[side == 0].sum())
        if abs(w0 - target0) <= tol:
            return side
        heavy = 0 if w0 > target0 else 1
        need = abs(w0 - target0)
        ext, inn, _ = _gains(A, side, rows)
        cand = np.nonzero((side == heavy) & (ext > 0))[0]
        if cand.size == 0:
            return side
        order = cand[np.argsort(-(ext - inn)[cand], kind="stable")]
        cum = np.cumsum(vw[order])
        take = order[: int(np.searchsorted(cum, need, side="left")) + 1]
        side = side.copy()
        side[take] = 1 - heavy
    return side


def _refine(A, vw, side, target0, tol, passes=10):
    """simultaneous-move Fiduccia-Mattheyses passes on a two-way split (module docstring)"""
    rows = _rows(A)
    cols = A.indices
    side = _rebalance(A, vw, side.copy(), target0, tol, rows)
    tie = _pair_key(np.arange(A.shape[0]), np.arange(A.shape[0]) + 7)
    best_cut, best = None, side
    for _ in range(passes):
        ext, inn, same = _gains(A, side, rows)
        cut = float(ext.sum()) / 2.0
        if best_cut is not None and cut >= best_cut:
            break
        best_cut, best = cut, side
        gain = ext - inn
        cand = gain > 0
        if not cand.any():
            break
        pri = np.where(cand, (np.round(gain * 16).astype(np.int64) << 20) + tie, -1)
        other = _row_max(A.indptr, np.where(same, -1, pri[cols]), -1)
        move = cand & (pri > other)
        w0 = float(vw[side == 0].sum())
        m01, m10 = np.nonzero(move & (side == 0))[0], np.nonzero(move & (side == 1))[0]
        # side 0 may lose at most (w0 - target0 + tol) net and gain at most (target0 + tol - w0) net
        m01 = _trim(m01, gain, vw, (w0 - target0 + tol) + float(vw[m10].sum()))
        m10 = _trim(m10, gain, vw, (target0 + tol - w0) + float(vw[m01].sum()))
        m01 = _trim(m01, gain, vw, (w0 - target0 + tol) + float(vw[m10].sum()))
        if m01.size + m10.size == 0:
            break
        side = side.copy()
        side[m01], side[m10] = 1, 0
    ext, _, _ = _gains(A, side, rows)
    return side if float(ext.sum()) / 2.0 < best_cut else best


def _bfs_order(A, seed):
    """vertices in breadth-first order from `seed` (unreached components appended), and the last one reached"""
    n = A.shape[0]
    seen = np.zeros(n, dtype=bool)
    seen[seed] = True
    order, frontier = [np.array([seed])], np.array([seed])
    while frontier.size:
        lens = A.indptr[frontier + 1] - A.indptr[frontier]
        starts = np.repeat(A.indptr[frontier] - np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)
        nb = np.unique(A.indices[starts + np.arange(int(lens.sum()))])
        frontier = nb[~seen[nb]]
        seen[frontier] = True
        order.append(frontier)
    rest = np.nonzero(~seen)[0]
    reached = np.concatenate(order)
    return np.concatenate([reached, rest]), int(reached[-1])


def _initial(A, vw, xy, frac):
    """coarsest graph: the best of eight weighted-median planes and four breadth-first growths (from the ends of
    two pseudo-diameters), each refined"""
    total = float(vw.sum())
    target0 = frac * total
    tol = max(0.02 * total, float(vw.max()))
    orders = []
    for k in range(8):
        t = np.pi * k / 8.0
        orders.append(np.argsort(xy[:, 0] * np.cos(t) + xy[:, 1] * np.sin(t), kind="stable"))
    for start in (int(np.argmin(xy[:, 0] + xy[:, 1])), int(np.argmin(xy[:, 0] - xy[:, 1]))):
        _, far = _bfs_order(A, start)
        o1, far2 = _bfs_order(A, far)
        o2, _ = _bfs_order(A, far2)
        orders += [o1, o2]
    best, best_cut = None, None
    for order in orders:
        cum = np.cumsum(vw[order])
        n0 = int(np.searchsorted(cum, target0, side="left")) + 1
        side = np.ones(A.shape[0], dtype=np.int8)
        side[order[:n0]] = 0
        side = _refine(A, vw, side, target0, tol)
        c = edge_cut(A, side)
        if best_cut is None or c < best_cut:
            best, best_cut = side, c
    return best


def bisect(A, vw, xy, frac=0.5, imbalance=0.005):
    """two-way split of the weighted graph: side[v] in {0, 1}, weight(side 0) = frac * total +- imbalance * total"""
    levels = []
    Ak, vk, xk = A, vw, xy
    while Ak.shape[0] > _COARSEST:
        Ac, vc, xc, cmap = _coarsen(Ak, vk, xk)
        if Ac.shape[0] > 0.9 * Ak.shape[0]:
            break
        levels.append((Ak, vk, cmap))
        Ak, vk, xk = Ac, vc, xc
    side = _initial(Ak, vk, xk, frac)
    total = float(vw.sum())
    for Af, vf, cmap in reversed(levels):
        side = side[cmap]
        tol = max(imbalance * total, float(vf.max()))
        side = _refine(Af, vf, side, frac * total, tol)
    return side


def _plane_split(A, xy, h, frac, imbalance):
    """the coordinate bisection of `partition.partition_rcb` (the axis whose median plane severs fewer edges), its
    boundary then refined on the graph"""
    n = A.shape[0]
    n0 = int(round(n * frac))
    best, best_cut = None, None
    for d in (0, 1):
        order = np.argsort(xy[:, d], kind="stable")
        side = np.ones(n, dtype=np.int8)
        side[order[:n0]] = 0
        c = edge_cut(A, side)
        if best_cut is None or c < best_cut:
            best, best_cut = side, c
    return _refine(A, np.ones(n), best, frac * n, max(imbalance * n, 1.0))


def partition_graph(coords, cells, n_parts, imbalance=0.005):
    """part[v] in [0, n_parts): recursive bisection of the mesh's vertex graph; every bisection is the better (by edge
    cut) of the multilevel split (module docstring) and the graph-refined median plane.  The vertex counts of the parts
    differ by at most about `imbalance` * (levels of bisection) of the mean."""
    coords = np.asarray(coords, dtype=np.float64)
    nv = coords.shape[0]
    G = vertex_graph(nv, cells).astype(np.float64).tocsr()
    G.sort_indices()
    part = np.zeros(nv, dtype=np.int32)

    def split(idx, A, first, count):
        if count == 1:
            part[idx] = first
            return
        left = count // 2
        side = bisect(A, np.ones(idx.size), coords[idx], frac=left / count, imbalance=imbalance)
        plane = _plane_split(A, coords[idx], None, left / count, imbalance)
        if edge_cut(A, plane) < edge_cut(A, side):
            side = plane
        for s, f, c in ((0, first, left), (1, first + left, count - left)):
            sel = np.nonzero(side == s)[0]
            if c == 1:
                part[idx[sel]] = f
            else:
                split(idx[sel], A[sel][:, sel].tocsr(), f, c)

    split(np.arange(nv), G, 0, int(n_parts))
    return part


def quality(coords, cells, part, depth=1):
    """edge cut, largest part / mean, most neighbours of a part, largest ghost set (`depth` layers) of a part"""
    from .partition import _layers
    nv = np.asarray(coords).shape[0]
    G = vertex_graph(nv, cells).astype(np.float64).tocsr()
    n_parts = int(part.max()) + 1
    counts = np.bincount(part, minlength=n_parts)
    rows = _rows(G)
    cross = part[rows] != part[G.indices]
    pairs = np.unique(part[rows[cross]].astype(np.int64) * n_parts + part[G.indices[cross]])
    neighbours = np.bincount(pairs // n_parts, minlength=n_parts)
    ghosts = [int(np.count_nonzero(_layers(G, part == q, depth) > 0)) for q in range(n_parts)]
    return dict(edge_cut=edge_cut(G, part), imbalance=float(counts.max() / counts.mean()),
                max_neighbours=int(neighbours.max()), max_ghosts=int(max(ghosts)), sum_ghosts=int(sum(ghosts)))
