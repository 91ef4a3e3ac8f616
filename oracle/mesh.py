"""Triangle meshes in DOLFIN's built-in numbering (oracle; test infra).

Numbering follows what the reference's golden files show:
``RectangleMesh(P0, P1, nx, ny)`` ("right" diagonal), evidence
tests/integrated_tests/time_of_flight/20220707_results/electrons000000.vtu,
and ``RectangleMesh(..., "crossed")``, evidence
tests/integrated_tests/glow_discharge/20220707_results/electrons.h5.
Boundary marking restates fedm/functions.py:73-124 (``LineSubDomain`` /
``Marking_boundaries``): an exterior facet gets tag ``idx+1`` when both end
points and the midpoint lie in the closed, DOLFIN_EPS-widened box.
"""
import numpy as np

DOLFIN_EPS = 3.0e-16


class Mesh:
    def __init__(self, coords, cells):
        self.coords = np.ascontiguousarray(coords, dtype=np.float64)   # (Nv,2) = (r,z)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)       # (Nc,3)
        self.nv = self.coords.shape[0]
        self.nc = self.cells.shape[0]
        self._facets = None

    # -- exterior facets -------------------------------------------------
    def exterior_facets(self):
        """(cell index, local opposite-vertex index) of every exterior facet.

        Local facet i is the edge opposite local vertex i (UFC convention)."""
        if self._facets is None:
            c = self.cells
            edges = np.concatenate([c[:, [1, 2]], c[:, [0, 2]], c[:, [0, 1]]])
            loc = np.repeat(np.arange(3), self.nc)
            cell = np.tile(np.arange(self.nc), 3)
            key = np.sort(edges, axis=1).astype(np.int64)
            key = key[:, 0] * self.nv + key[:, 1]
            _, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
            ext = cnt[inv] == 1
            self._facets = (cell[ext], loc[ext])
        return self._facets

    def hmax(self):
        x = self.coords[self.cells]
        e = [np.linalg.norm(x[:, i] - x[:, j], axis=1) for i, j in ((0, 1), (1, 2), (0, 2))]
        return float(np.max(e))


def _axis(a, b, n):
    """DOLFIN's grid line formula a + i*(b-a)/n."""
    return a + np.arange(n + 1, dtype=np.float64) * (b - a) / n


def rectangle_right(x0, y0, x1, y1, nx, ny, xs=None, ys=None):
    """DOLFIN RectangleMesh, default ("right") diagonal.  Optional graded axes."""
    xs = _axis(x0, x1, nx) if xs is None else np.asarray(xs, dtype=np.float64)
    ys = _axis(y0, y1, ny) if ys is None else np.asarray(ys, dtype=np.float64)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    a = (j * (nx + 1) + i).ravel()
    b, c, d = a + 1, a + nx + 1, a + nx + 2
    cells = np.empty((2 * nx * ny, 3), dtype=np.int32)
    cells[0::2] = np.stack([a, b, d], axis=1)
    cells[1::2] = np.stack([a, c, d], axis=1)
    return Mesh(coords, cells)


def rectangle_crossed(x0, y0, x1, y1, nx, ny, xs=None, ys=None):
    """DOLFIN RectangleMesh(..., "crossed"): corner vertices, then cell centres."""
    xs = _axis(x0, x1, nx) if xs is None else np.asarray(xs, dtype=np.float64)
    ys = _axis(y0, y1, ny) if ys is None else np.asarray(ys, dtype=np.float64)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    xm, ym = 0.5 * (xs[:-1] + xs[1:]), 0.5 * (ys[:-1] + ys[1:])
    XM, YM = np.meshgrid(xm, ym, indexing="xy")
    coords = np.concatenate([np.stack([X.ravel(), Y.ravel()], axis=1),
                             np.stack([XM.ravel(), YM.ravel()], axis=1)])
    i, j = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    a = (j * (nx + 1) + i).ravel()
    b, c, d = a + 1, a + nx + 1, a + nx + 2
    m = (nx + 1) * (ny + 1) + (j * nx + i).ravel()
    cells = np.empty((4 * nx * ny, 3), dtype=np.int32)
    cells[0::4] = np.stack([a, b, m], axis=1)
    cells[1::4] = np.stack([a, c, m], axis=1)
    cells[2::4] = np.stack([b, d, m], axis=1)
    cells[3::4] = np.stack([c, d, m], axis=1)
    return Mesh(coords, cells)


def graded_axis(length, n, ratio):
    """n intervals on [0,length], geometric growth so that last/first = ratio."""
    if ratio == 1.0:
        return np.linspace(0.0, length, n + 1)
    q = ratio ** (1.0 / (n - 1))
    h = np.cumsum(np.concatenate([[0.0], q ** np.arange(n)]))
    return length * h / h[-1]


def mark_boundaries(mesh, boundaries):
    """Facet tags per (cell, local facet): int8 (Nc,3), 0 = interior/unmarked.

    ``boundaries`` is FEDM's list ``['line', z1, z2, r1, r2]``
    (fedm/functions.py:86-124).  Later entries overwrite earlier ones, as
    ``SubDomain.mark`` does."""
    tags = np.zeros((mesh.nc, 3), dtype=np.int8)
    cell, loc = mesh.exterior_facets()
    ends = {0: (1, 2), 1: (0, 2), 2: (0, 1)}
    va = np.array([ends[l][0] for l in range(3)])[loc]
    vb = np.array([ends[l][1] for l in range(3)])[loc]
    pa = mesh.coords[mesh.cells[cell, va]]
    pb = mesh.coords[mesh.cells[cell, vb]]
    pm = 0.5 * (pa + pb)
    for idx, bnd in enumerate(boundaries):
        if bnd[0] != "line":
            raise ValueError(
                f"fedm.Marking_boundaries: Invalid boundary_type '{bnd[0]}'. "
                "Possible values are 'circle', 'line'.")
        z1, z2 = bnd[1] - DOLFIN_EPS, bnd[2] + DOLFIN_EPS
        r1, r2 = bnd[3] - DOLFIN_EPS, bnd[4] + DOLFIN_EPS
        inside = np.ones(len(cell), dtype=bool)
        for p in (pa, pb, pm):
            inside &= (p[:, 0] >= r1) & (p[:, 0] <= r2) & (p[:, 1] >= z1) & (p[:, 1] <= z2)
        tags[cell[inside], loc[inside]] = idx + 1
    return tags
