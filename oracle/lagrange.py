"""Lagrange P_k interpolation on the reference triangle (oracle; test infra).

DOLFIN represents ``Expression(..., degree=k)`` inside a form by its nodal
interpolant in the degree-k Lagrange element of each cell (point evaluation at
the equispaced lattice nodes).  The interpolating polynomial does not depend on
the basis, so this module works with a monomial Vandermonde matrix.
Used for fedm-tof.py:107 (degree 3), :116 (degree 2), :120 (degree 2).
"""
import numpy as np


def lattice(k):
    """Equispaced P_k nodes on the reference triangle, (n,2)."""
    if k == 0:
        return np.array([[1.0 / 3.0, 1.0 / 3.0]])
    pts = [(i / k, j / k) for j in range(k + 1) for i in range(k + 1 - j)]
    return np.array(pts, dtype=np.float64)


def _vandermonde(k, pts):
    cols = [pts[:, 0] ** a * pts[:, 1] ** b
            for b in range(k + 1) for a in range(k + 1 - b)]
    return np.stack(cols, axis=1)


def interpolation_matrix(k, xq):
    """B (nq, nnodes): value at xq of the P_k interpolant of nodal values."""
    nodes = lattice(k)
    vn = _vandermonde(k, nodes)
    vq = _vandermonde(k, np.asarray(xq, dtype=np.float64))
    return vq @ np.linalg.inv(vn)


def p1_basis(xq):
    """phi (nq,3) of the P1 basis 1-x-y, x, y at reference points xq."""
    xq = np.asarray(xq, dtype=np.float64)
    return np.stack([1.0 - xq[:, 0] - xq[:, 1], xq[:, 0], xq[:, 1]], axis=1)
