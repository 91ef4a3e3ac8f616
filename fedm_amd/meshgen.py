"""Locally refined, unstructured triangle meshes of a rectangle (cold path, host).

The reference's streamer case loads an externally generated, locally refined mesh
(``Mesh('mesh.xml')``, examples/streamer_discharge/fedm-streamer.py:116) that is not in its
checkout (.MISSING_LARGE_BLOBS:2).  This module makes a stand-in of the same kind: a Delaunay
triangulation of a point set whose spacing follows a size function -- fine along the axis and
the streamer channel, doubling level by level away from it.  Deterministic (no random numbers):
the same arguments give the same vertices, cells and numbering on every machine.

Construction: nested hexagonal lattices ``L_0 > L_1 > ...`` (spacing ``h0 * 2**l``); a lattice
point of ``L_l`` is kept where the size function asks for level ``l`` or finer lattices do not
reach, so the union is a graded point set whose Delaunay triangulation has equilateral cells in
the uniform regions and short transition bands (vertex valence 4-10) between them.  The vertex
numbering is that of the point set (level by level, lattice order inside a level): "arbitrary"
from the device path's point of view, which renumbers internally (``device.locality_order``).
"""
import numpy as np

from .mesh import Mesh


def _levels(size, h0, n_levels):
    with np.errstate(divide="ignore"):
        lv = np.floor(np.log2(np.maximum(size, h0) / h0) + 1e-12).astype(np.int64)
    return np.clip(lv, 0, n_levels - 1)


def box_distance_size(fine_box, h_fine, growth, h_max):
    """Size function: ``h_fine`` inside ``fine_box = (r0, r1, z0, z1)``, growing linearly with the
    distance from it (``growth`` = dh/d distance) up to ``h_max``."""
    r0, r1, z0, z1 = fine_box

    def size(p):
        dr = np.maximum(np.maximum(r0 - p[:, 0], p[:, 0] - r1), 0.0)
        dz = np.maximum(np.maximum(z0 - p[:, 1], p[:, 1] - z1), 0.0)
        return np.minimum(h_fine + growth * np.hypot(dr, dz), h_max)
    return size


def refined_rectangle(width, height, size, h_fine, n_levels=6, smooth=2, retriangulate=True):
    """Delaunay mesh of ``[0, width] x [0, height]`` graded by ``size(points) -> h``.

    Returns a :class:`fedm_amd.mesh.Mesh`.  All four sides carry vertices exactly on them
    (``Marking_boundaries`` and the Dirichlet values of the scripts test coordinates against the
    box with DOLFIN_EPS).  ``retriangulate=False``: the smoothing sweeps keep the first triangulation
    (a third of the time on meshes of millions of vertices, where Qhull dominates; the sweeps move a vertex by
    a fraction of its cell, so the cells stay positive -- checked, with the Delaunay pass as the fallback)."""
    from scipy.spatial import Delaunay
    top = 2 ** (n_levels - 1)
    # level-0 lattice: columns h0 apart, rows dz0 apart, both dividing the box into a multiple of
    # 2**(n_levels-1) intervals so that every level's lattice fits the box exactly
    nx = max(1, int(round(width / (h_fine * top)))) * top
    nz = max(1, int(round(height / (h_fine * np.sqrt(0.75) * top)))) * top
    h0, dz0 = width / nx, height / nz
    pts, on_side = [], []
    for lv in range(n_levels):
        step = 2 ** lv
        k = np.arange(0, nz + 1, step)                       # rows of this level
        odd = (k // step) % 2 == 1
        for rows, shift in ((k[~odd], 0), (k[odd], step)):   # shift in units of h0/2
            if rows.size == 0:
                continue
            i = np.arange(shift, 2 * nx + 1, 2 * step)       # r in units of h0/2
            if shift:                                         # odd rows: close the two sides
                i = np.concatenate([[0], i, [2 * nx]])
            I, K = np.meshgrid(i, rows, indexing="xy")
            p = np.stack([I.ravel() * (0.5 * h0), K.ravel() * dz0], axis=1)
            p[I.ravel() == 2 * nx, 0] = width                # the far sides exactly, whatever the rounding
            p[K.ravel() == nz, 1] = height
            want = _levels(size(p), h0, n_levels)
            # a point of L_lv is kept where level lv is wanted; coarser lattices are subsets of the
            # finer ones, so "finer wanted" points were emitted by an earlier level already
            keep = want == lv
            pts.append(p[keep])
            on_side.append(((I.ravel() == 0) | (I.ravel() == 2 * nx) | (K.ravel() == 0) | (K.ravel() == nz))[keep])
    p = np.concatenate(pts)
    side = np.concatenate(on_side)
    # the same lattice point can be wanted by two levels only through the closing side points of
    # odd rows: drop exact duplicates, keeping the first
    key = np.round(p[:, 0] / (0.25 * h0)).astype(np.int64) * (4 * nz + 8) + np.round(p[:, 1] / (0.5 * dz0)).astype(np.int64)
    _, first = np.unique(key, return_index=True)
    first.sort()
    p, side = p[first], side[first]
    tri = Delaunay(p).simplices
    for _ in range(smooth):
        p = _laplace_smooth(p, tri, side)
        if retriangulate:
            tri = Delaunay(p).simplices
    if not retriangulate and smooth:
        a, b, c = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
        det = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
        d0 = np.abs(det)
        if (d0 < 1e-3 * np.median(d0)).sum() > (d0 == 0).sum() + 64:     # squashed cells beyond the collinear boundary ones
            tri = Delaunay(p).simplices
    tri = _positive_cells(p, tri)
    return Mesh(p, tri.astype(np.int32))


def _positive_cells(p, tri):
    """Counter-clockwise cells; cells of zero area (collinear boundary points) are dropped."""
    a, b, c = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
    det = (b[:, 0] - a[:, 0]) * (c[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (c[:, 0] - a[:, 0])
    scale = np.abs(det).max()
    good = np.abs(det) > 1e-12 * scale
    tri = tri[good].copy()
    flip = det[good] < 0
    tri[flip] = tri[flip][:, [0, 2, 1]]
    return tri


def _laplace_smooth(p, tri, fixed):
    """One Jacobi sweep of Laplacian smoothing (vertices on the sides stay)."""
    n = p.shape[0]
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]).astype(np.int64)
    key = np.unique(np.concatenate([e[:, 0] * n + e[:, 1], e[:, 1] * n + e[:, 0]]))     # directed edges, once each
    a, b = key // n, key % n
    cnt = np.bincount(a, minlength=n).astype(float)
    acc = np.stack([np.bincount(a, weights=p[b, d], minlength=n) for d in (0, 1)], axis=1)
    q = p.copy()
    move = ~fixed & (cnt > 0)
    q[move] = 0.5 * p[move] + 0.5 * acc[move] / cnt[move, None]
    return q


def mesh_quality(mesh):
    """Valence range, smallest angle (degrees), edge-length range: what the tests assert on."""
    c, x = mesh.cells, mesh.coords
    val = np.bincount(np.unique(np.sort(np.concatenate([c[:, [0, 1]], c[:, [1, 2]], c[:, [2, 0]]]), axis=1),
                                axis=0).ravel(), minlength=x.shape[0])
    ang = []
    for i in range(3):
        a, b, d = x[c[:, i]], x[c[:, (i + 1) % 3]], x[c[:, (i + 2) % 3]]
        u, v = b - a, d - a
        cosang = (u * v).sum(axis=1) / (np.linalg.norm(u, axis=1) * np.linalg.norm(v, axis=1))
        ang.append(np.degrees(np.arccos(np.clip(cosang, -1.0, 1.0))))
    return dict(n_vertices=int(x.shape[0]), n_cells=int(c.shape[0]), valence_min=int(val.min()),
                valence_max=int(val.max()), min_angle=float(np.min(ang)), hmin=mesh.hmin(), hmax=mesh.hmax())
