"""Quadrature rules used by FFC/FIAT 2019.1.0 "default" scheme (oracle; test infra).

The reference never names a rule; it sets ``parameters["form_compiler"]
["quadrature_degree"]`` (examples/streamer_discharge/fedm-streamer.py:23 -> 2,
examples/glow_discharge/fedm-gd.py:28 -> 4) or leaves the degree to UFL's
estimator (examples/time_of_flight/fedm-tof.py:18-20).  FFC then asks FIAT
``create_quadrature(cell, degree, "default")`` which (published algorithm,
FIAT 2019.1.0 ``quadrature_schemes.py``) returns hard-coded symmetric rules
for triangles up to degree 6 and the collapsed Gauss-Jacobi rule with
``m = (degree + 2) // 2`` points per direction above that.  Intervals always
get Gauss-Legendre with ``m = (degree + 2) // 2`` points.

Reference-cell convention: triangle (0,0),(1,0),(0,1), weights sum to 1/2;
interval [0,1], weights sum to 1.
"""
import numpy as np
from scipy.special import roots_jacobi


def _perm3(a, b):
    """The three points (a,b),(b,a),(b,b)-style orbit of barycentric (a,b,b)."""
    return [(a, b), (b, a), (b, b)]


def triangle_rule(degree):
    """Return (points (nq,2), weights (nq,)) on the reference triangle."""
    if degree < 0:
        raise ValueError("negative quadrature degree")
    if degree <= 1:
        x = [(1.0 / 3.0, 1.0 / 3.0)]
        w = [0.5]
    elif degree == 2:
        x = [(1.0 / 6.0, 1.0 / 6.0), (1.0 / 6.0, 2.0 / 3.0), (2.0 / 3.0, 1.0 / 6.0)]
        w = [1.0 / 6.0] * 3
    elif degree == 3:
        x = [(0.659027622374092, 0.231933368553031),
             (0.659027622374092, 0.109039009072877),
             (0.231933368553031, 0.659027622374092),
             (0.231933368553031, 0.109039009072877),
             (0.109039009072877, 0.659027622374092),
             (0.109039009072877, 0.231933368553031)]
        w = [1.0 / 12.0] * 6
    elif degree == 4:
        x = [(0.816847572980459, 0.091576213509771),
             (0.091576213509771, 0.816847572980459),
             (0.091576213509771, 0.091576213509771),
             (0.108103018168070, 0.445948490915965),
             (0.445948490915965, 0.108103018168070),
             (0.445948490915965, 0.445948490915965)]
        w = [0.109951743655322 / 2.0] * 3 + [0.223381589678011 / 2.0] * 3
    elif degree == 5:
        x = [(0.33333333333333333, 0.33333333333333333),
             (0.79742698535308720, 0.10128650732345633),
             (0.10128650732345633, 0.79742698535308720),
             (0.10128650732345633, 0.10128650732345633),
             (0.05971587178976981, 0.47014206410511505),
             (0.47014206410511505, 0.05971587178976981),
             (0.47014206410511505, 0.47014206410511505)]
        w = [0.22500000000000000 / 2.0] + [0.12593918054482717 / 2.0] * 3 \
            + [0.13239415278850616 / 2.0] * 3
    elif degree == 6:
        x = [(0.873821971016996, 0.063089014491502),
             (0.063089014491502, 0.873821971016996),
             (0.063089014491502, 0.063089014491502),
             (0.501426509658179, 0.249286745170910),
             (0.249286745170910, 0.501426509658179),
             (0.249286745170910, 0.249286745170910),
             (0.636502499121399, 0.310352451033785),
             (0.636502499121399, 0.053145049844816),
             (0.310352451033785, 0.636502499121399),
             (0.310352451033785, 0.053145049844816),
             (0.053145049844816, 0.636502499121399),
             (0.053145049844816, 0.310352451033785)]
        w = [0.050844906370207 / 2.0] * 3 + [0.116786275726379 / 2.0] * 3 \
            + [0.082851075618374 / 2.0] * 6
    else:
        return collapsed_triangle_rule((degree + 2) // 2)
    return np.array(x, dtype=np.float64), np.array(w, dtype=np.float64)


def collapsed_triangle_rule(m):
    """FIAT ``CollapsedQuadratureTriangleRule``: Gauss-Jacobi (0,0) x (1,0)."""
    ptx, wx = roots_jacobi(m, 0.0, 0.0)
    pty, wy = roots_jacobi(m, 1.0, 0.0)
    pts, wts = [], []
    for x, w1 in zip(ptx, wx):
        for y, w2 in zip(pty, wy):
            xi1 = 0.5 * (1.0 + x) * (1.0 - y) - 1.0   # collapse to [-1,1] triangle
            xi2 = y
            pts.append((0.5 * (xi1 + 1.0), 0.5 * (xi2 + 1.0)))
            wts.append(0.5 * 0.25 * w1 * w2)
    return np.array(pts), np.array(wts)


def interval_rule(degree):
    """Gauss-Legendre on [0,1] with m = (degree+2)//2 points."""
    m = max((degree + 2) // 2, 1)
    x, w = roots_jacobi(m, 0.0, 0.0)
    return 0.5 * (x + 1.0), 0.5 * w
