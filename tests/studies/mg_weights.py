"""V(1,1) with DIFFERENT damping for the pre- and the post-smoothing sweep (the two sweeps of a
cycle as the two roots of a degree-2 Chebyshev polynomial) -- same cost as the shipped cycle.
python tests/studies/mg_weights.py [n=288]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
src = open(os.path.join(ROOT, "tests", "studies", "mg_quality.py")).read()
g = {"__file__": os.path.join(ROOT, "tests", "studies", "mg_quality.py")}
exec(src[:src.index("t0 = time.time()")], g)
A, fixed, nv, quality, hierarchy = g["A"], g["fixed"], g["nv"], g["quality"], g["hierarchy"]
lv = hierarchy(A, fixed)
print("levels", [a.shape[0] for a, _ in lv])


def cycle(levels, w_pre, w_post):
    lu = spla.splu(levels[-1][0].tocsc())
    dinv = [1.0 / a.diagonal() for a, _ in levels]

    def cyc(l, b):
        if l == len(levels) - 1:
            return lu.solve(b)
        a, p = levels[l]
        x = w_pre * dinv[l] * b
        x = x + p @ cyc(l + 1, p.T @ (b - a @ x))
        return x + w_post * dinv[l] * (b - a @ x)
    return lambda b: cyc(0, b)


quality("V(1,1) 0.85 / 0.85 (shipped)", cycle(lv, 0.85, 0.85))
lam = 1.97
for frac in (3, 4, 6, 8, 12):
    lo, hi = 1.05 * lam / frac, 1.05 * lam
    th, de = 0.5 * (hi + lo), 0.5 * (hi - lo)
    r1, r2 = th + de * np.cos(np.pi / 4), th + de * np.cos(3 * np.pi / 4)
    for w1, w2 in ((1 / r1, 1 / r2), (1 / r2, 1 / r1)):
        quality(f"V(1,1) weights {w1:.3f} / {w2:.3f}  (Chebyshev roots on lmax/{frac})", cycle(lv, w1, w2))
for w1, w2 in ((0.6, 1.1), (1.1, 0.6), (0.7, 1.0), (1.0, 0.7), (0.55, 1.3), (1.3, 0.55)):
    quality(f"V(1,1) weights {w1} / {w2}", cycle(lv, w1, w2))
