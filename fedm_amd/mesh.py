"""Triangle meshes and boundary marking for the device path.

Stands in for the DOLFIN objects a FEDM script creates before the time loop:
``RectangleMesh`` (examples/time_of_flight/fedm-tof.py:87,
examples/glow_discharge/fedm-gd.py:157) with DOLFIN's vertex/cell numbering,
and ``Marking_boundaries`` (fedm/functions.py:86-124).
"""
import numpy as np

DOLFIN_EPS = 3.0e-16


class Mesh:
    """coords (Nv,2) as (r,z) and cells (Nc,3)."""

    def __init__(self, coords, cells):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        if self.coords.ndim != 2 or self.coords.shape[1] != 2:
            raise ValueError("coords must have shape (n_vertices, 2)")
        if self.cells.ndim != 2 or self.cells.shape[1] != 3:
            raise ValueError("cells must have shape (n_cells, 3)")

    def num_vertices(self):
        return self.coords.shape[0]

    def num_cells(self):
        return self.cells.shape[0]

    def _edge_lengths(self):
        x = self.coords[self.cells]
        return np.stack([np.linalg.norm(x[:, i] - x[:, j], axis=1)
                         for i, j in ((0, 1), (1, 2), (0, 2))])

    def hmax(self):
        return float(self._edge_lengths().max())

    def hmin(self):
        return float(self._edge_lengths().min())

    def ufl_cell(self):
        """``FiniteElement("Lagrange", mesh.ufl_cell(), 1)`` (fedm-streamer.py:132)."""
        return "triangle"

    def exterior_facets(self):
        """(cell, local facet) of every boundary edge; facet i is opposite vertex i."""
        c = self.cells.astype(np.int64)
        nv = self.num_vertices()
        pairs = np.concatenate([c[:, [1, 2]], c[:, [0, 2]], c[:, [0, 1]]])
        pairs.sort(axis=1)
        key = pairs[:, 0] * nv + pairs[:, 1]
        _, inverse, counts = np.unique(key, return_inverse=True, return_counts=True)
        boundary = counts[inverse] == 1
        cell = np.tile(np.arange(c.shape[0]), 3)[boundary]
        local = np.repeat(np.arange(3), c.shape[0])[boundary]
        return cell, local


def _grid_lines(a, b, n, lines):
    if lines is not None:
        lines = np.asarray(lines, dtype=np.float64)
        if lines.size != n + 1:
            raise ValueError("graded axis must have n+1 points")
        return lines
    return a + np.arange(n + 1, dtype=np.float64) * (b - a) / n


def RectangleMesh(p0, p1, nx, ny, diagonal="right", x_lines=None, y_lines=None):
    """DOLFIN's RectangleMesh numbering for the "right" and "crossed" patterns.

    ``x_lines`` / ``y_lines`` optionally replace the uniform grid lines (a graded
    tensor-product mesh with the same connectivity)."""
    xs = _grid_lines(p0[0], p1[0], nx, x_lines)
    ys = _grid_lines(p0[1], p1[1], ny, y_lines)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    corners = np.stack([X.ravel(), Y.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    a = (j * (nx + 1) + i).ravel()
    b, c, d = a + 1, a + nx + 1, a + nx + 2
    if diagonal == "right":
        cells = np.empty((2 * nx * ny, 3), dtype=np.int32)
        cells[0::2] = np.stack([a, b, d], axis=1)
        cells[1::2] = np.stack([a, c, d], axis=1)
        return Mesh(corners, cells)
    if diagonal == "crossed":
        xm, ym = 0.5 * (xs[:-1] + xs[1:]), 0.5 * (ys[:-1] + ys[1:])
        XM, YM = np.meshgrid(xm, ym, indexing="xy")
        centres = np.stack([XM.ravel(), YM.ravel()], axis=1)
        m = (nx + 1) * (ny + 1) + (j * nx + i).ravel()
        cells = np.empty((4 * nx * ny, 3), dtype=np.int32)
        cells[0::4] = np.stack([a, b, m], axis=1)
        cells[1::4] = np.stack([a, c, m], axis=1)
        cells[2::4] = np.stack([b, d, m], axis=1)
        cells[3::4] = np.stack([c, d, m], axis=1)
        return Mesh(np.concatenate([corners, centres]), cells)
    raise ValueError(f"unknown diagonal '{diagonal}', options are 'right', 'crossed'")


def geometric_lines(length, n, ratio):
    """n+1 grid lines on [0,length]; interval sizes grow geometrically, last/first = ratio."""
    if n < 2 or ratio == 1.0:
        return np.linspace(0.0, length, n + 1)
    q = ratio ** (1.0 / (n - 1))
    h = np.concatenate([[0.0], np.cumsum(q ** np.arange(n))])
    return length * h / h[-1]


class SubDomain:
    """Point predicate of a boundary part (DOLFIN's SubDomain as the two classes below use it):
    ``inside(x, on_boundary)`` for one point ``x = (r, z)``."""

    def inside(self, x, on_boundary):
        raise NotImplementedError


class LineSubDomain(SubDomain):
    """Axis-parallel boundary segment, fedm/functions.py:73-84: the closed box ``r_range x z_range``
    widened by DOLFIN_EPS (``df.between``), on the boundary only."""

    def __init__(self, r_range, z_range):
        self._r_range, self._z_range = tuple(r_range), tuple(z_range)

    def inside(self, x, on_boundary):
        within = all(lo - DOLFIN_EPS <= c <= hi + DOLFIN_EPS
                     for c, (lo, hi) in zip((x[0], x[1]), (self._r_range, self._z_range)))
        return bool(within and on_boundary)


class CircleSubDomain(SubDomain):
    """Circular electrode tip, fedm/functions.py:49-70.  As in the reference the constructor does not
    keep ``gap_length``, so ``inside`` fails with the same AttributeError for a tip at z > 0 -- the
    reason no reference example uses the 'circle' boundary type."""

    def __init__(self, center_z, center_r, radius, gap_length, submesh=False, tol=1e-8):
        self._center_z, self._center_r, self._radius = float(center_z), float(center_r), float(radius)
        self._submesh, self._tol = bool(submesh), float(tol)

    def inside(self, x, on_boundary):
        d2 = (x[0] - self._center_r) ** 2 + (x[1] - self._center_z) ** 2
        on_circle = abs(d2 - self._radius ** 2) <= self._tol
        side = x[1] <= 0 if self._center_z <= 0 else x[1] >= self._gap_length
        return bool(on_circle and side and (on_boundary or self._submesh))


def Marking_boundaries(mesh, boundaries, submesh=False, gap_length=0.01):
    """Facet tags (n_cells, 3), int8; tag idx+1 for ``boundaries[idx]``.

    Same arguments and the same ValueError as fedm/functions.py:86-124.  A
    boundary edge is tagged when its two end points and its midpoint lie in the
    closed box widened by DOLFIN_EPS (``LineSubDomain.inside`` + DOLFIN's
    ``SubDomain.mark``).  The reference's 'circle' type cannot work there
    (``CircleSubDomain`` reads an attribute it never sets, functions.py:49-69);
    it raises the same AttributeError here."""
    tags = np.zeros((mesh.num_cells(), 3), dtype=np.int8)
    cell, local = mesh.exterior_facets()
    first = np.array([1, 0, 0])[local]
    second = np.array([2, 2, 1])[local]
    pa = mesh.coords[mesh.cells[cell, first]]
    pb = mesh.coords[mesh.cells[cell, second]]
    pm = 0.5 * (pa + pb)
    for idx, boundary in enumerate(boundaries):
        kind = boundary[0]
        if kind == "circle":
            raise AttributeError("'CircleSubDomain' object has no attribute '_gap_length'")
        if kind != "line":
            raise ValueError(f"fedm.Marking_boundaries: Invalid boundary_type '{kind}'. "
                             "Possible values are 'circle', 'line'.")
        z1, z2 = boundary[1] - DOLFIN_EPS, boundary[2] + DOLFIN_EPS
        r1, r2 = boundary[3] - DOLFIN_EPS, boundary[4] + DOLFIN_EPS
        hit = np.ones(cell.size, dtype=bool)
        for p in (pa, pb, pm):
            hit &= (r1 <= p[:, 0]) & (p[:, 0] <= r2) & (z1 <= p[:, 1]) & (p[:, 1] <= z2)
        tags[cell[hit], local[hit]] = idx + 1
    return tags
