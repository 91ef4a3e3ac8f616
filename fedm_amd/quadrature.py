"""FIAT 2019.1.0 "default" quadrature rules for the element kernels.

FEDM scripts either set ``parameters["form_compiler"]["quadrature_degree"]``
(examples/streamer_discharge/fedm-streamer.py:23, glow_discharge/fedm-gd.py:28)
or leave the degree to UFL (examples/time_of_flight/fedm-tof.py).  FFC then
uses FIAT's ``create_quadrature(cell, degree, "default")``: fixed symmetric
rules on triangles up to degree 6, collapsed Gauss-Jacobi with
``(degree + 2) // 2`` points per direction above; Gauss-Legendre on facets.
The tables travel to the device inside ``fedm_model_desc``.
"""
import numpy as np
from scipy.special import roots_jacobi

_T = {
    1: ([(1 / 3, 1 / 3)], [0.5]),
    2: ([(1 / 6, 1 / 6), (1 / 6, 2 / 3), (2 / 3, 1 / 6)], [1 / 6] * 3),
    3: ([(0.659027622374092, 0.231933368553031), (0.659027622374092, 0.109039009072877),
         (0.231933368553031, 0.659027622374092), (0.231933368553031, 0.109039009072877),
         (0.109039009072877, 0.659027622374092), (0.109039009072877, 0.231933368553031)],
        [1 / 12] * 6),
    4: ([(0.816847572980459, 0.091576213509771), (0.091576213509771, 0.816847572980459),
         (0.091576213509771, 0.091576213509771), (0.108103018168070, 0.445948490915965),
         (0.445948490915965, 0.108103018168070), (0.445948490915965, 0.445948490915965)],
        [0.109951743655322 / 2] * 3 + [0.223381589678011 / 2] * 3),
    5: ([(0.33333333333333333, 0.33333333333333333),
         (0.79742698535308720, 0.10128650732345633), (0.10128650732345633, 0.79742698535308720),
         (0.10128650732345633, 0.10128650732345633), (0.05971587178976981, 0.47014206410511505),
         (0.47014206410511505, 0.05971587178976981), (0.47014206410511505, 0.47014206410511505)],
        [0.225 / 2] + [0.12593918054482717 / 2] * 3 + [0.13239415278850616 / 2] * 3),
    6: ([(0.873821971016996, 0.063089014491502), (0.063089014491502, 0.873821971016996),
         (0.063089014491502, 0.063089014491502), (0.501426509658179, 0.249286745170910),
         (0.249286745170910, 0.501426509658179), (0.249286745170910, 0.249286745170910),
         (0.636502499121399, 0.310352451033785), (0.636502499121399, 0.053145049844816),
         (0.310352451033785, 0.636502499121399), (0.310352451033785, 0.053145049844816),
         (0.053145049844816, 0.636502499121399), (0.053145049844816, 0.310352451033785)],
        [0.050844906370207 / 2] * 3 + [0.116786275726379 / 2] * 3 + [0.082851075618374 / 2] * 6),
}


def triangle(degree):
    """(points (n,2), weights (n,)) on the reference triangle; weights sum to 1/2."""
    degree = max(int(degree), 1)
    if degree in _T:
        x, w = _T[degree]
        return np.array(x, dtype=np.float64), np.array(w, dtype=np.float64)
    m = (degree + 2) // 2
    px, wx = roots_jacobi(m, 0.0, 0.0)
    py, wy = roots_jacobi(m, 1.0, 0.0)
    pts = [(0.25 * (1 + x) * (1 - y), 0.5 * (1 + y)) for x in px for y in py]
    wts = [0.125 * a * b for a in wx for b in wy]
    return np.array(pts), np.array(wts)


def interval(degree):
    """Gauss-Legendre on [0,1], (degree+2)//2 points."""
    m = max((int(degree) + 2) // 2, 1)
    x, w = roots_jacobi(m, 0.0, 0.0)
    return 0.5 * (x + 1.0), 0.5 * w


def lagrange_interpolation_matrix(k, xq):
    """B (nq, n_nodes): value at xq of the P_k nodal interpolant on the equispaced
    lattice (how DOLFIN represents Expression(degree=k) inside a form)."""
    nodes = np.array([(i / k, j / k) for j in range(k + 1) for i in range(k + 1 - j)])
    mono = lambda p: np.stack([p[:, 0] ** a * p[:, 1] ** b
                               for b in range(k + 1) for a in range(k + 1 - b)], axis=1)
    return mono(np.asarray(xq, dtype=np.float64)) @ np.linalg.inv(mono(nodes)), nodes
