"""How many outer Krylov steps a MORE ACCURATE potential solve saves on the Newton systems of a late
streamer step: k multigrid cycles per preconditioner application, combined by the Chebyshev semi-iteration
on the interval [1 - rho, 1] of the cycle's contraction rho (no inner products, a fixed linear operator).
python tests/studies/potential_acceleration.py [n=288]   (needs gpurun_out/late_<n>.npz from late_systems.py)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "288", "late", "noexec"]
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import precond_structure as ps     # hierarchy, cycles and the stored systems (study loop switched off)
from oracle import cpu_backend as cb

N, nv, iu, ip, Jpp0 = ps.N, ps.nv, ps.iu, ps.ip, ps.Jpp0


def contraction(cycle):
    """spectral radius of I - M^-1 A by power iteration"""
    x = np.random.default_rng(1).standard_normal(nv)
    x[ps.fixed] = 0.0
    lam = 0.0
    for _ in range(40):
        y = x - cycle(Jpp0 @ x)
        lam = np.linalg.norm(y) / np.linalg.norm(x)
        x = y / np.linalg.norm(y)
    return lam


def accelerated(cycle, k, rho):
    """k cycles, Chebyshev semi-iteration for eig(M^-1 A) in [1 - rho, 1]"""
    lmin, lmax = 1.0 - rho, 1.0
    th, de = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
    sigma = th / de

    def f(b):
        z = cycle(b)
        d = z / th
        x = d.copy()
        r_ = 1.0 / sigma
        for _ in range(k - 1):
            rn = 1.0 / (2.0 * sigma - r_)
            z = cycle(b - Jpp0 @ x)
            d = rn * r_ * d + 2.0 * rn / de * z
            x = x + d
            r_ = rn
        return x
    return f


def stationary(cycle, k):
    def f(b):
        x = cycle(b)
        for _ in range(k - 1):
            x = x + cycle(b - Jpp0 @ x)
        return x
    return f


cycles = {"V(1,1) Jacobi .85 [early default]": ps.V11, "V(2,2) Chebyshev smoother [~ hard-mode cycle]": ps.C22}
rhos = {name: contraction(c) for name, c in cycles.items()}
print({k: round(v, 3) for k, v in rhos.items()}, flush=True)
variants = []
for name, c in cycles.items():
    variants.append((f"1 x {name}", c, 1))
    for k in (2, 3, 4):
        variants.append((f"{k} x {name}, Chebyshev-accelerated", accelerated(c, k, min(rhos[name] * 1.05, 0.95)), k))
    variants.append((f"2 x {name}, stationary", stationary(c, 2), 2))
variants.append(("exact potential solve", ps.lpp.solve, 0))

total = {name: 0 for name, _, _ in variants}
for s_, (J, F) in enumerate(ps.systems):
    Juu, Jpu = J[iu][:, iu].tocsr(), J[ip][:, iu].tocsr()
    blocks = [np.linalg.inv(Juu[2 * v:2 * v + 2, 2 * v:2 * v + 2].toarray()) for v in range(nv)]
    D = sp.block_diag(blocks).tocsr()

    def cheb(ru, deg=4):
        w = cb.chebyshev_weights(deg)
        g = D @ ru
        z = w[0] * g
        for k in range(1, deg):
            z = z + w[k] * (g - D @ (Juu @ z))
        return z

    for name, pot, _ in variants:
        def apply(r):
            z = np.empty(N)
            zu = cheb(r[iu])
            z[iu], z[ip] = zu, pot(r[ip] - Jpu @ zu)
            return z
        its = [0]
        A = spla.LinearOperator((N, N), matvec=lambda y: J @ apply(y))
        spla.gmres(A, -F, rtol=1e-5, restart=30, maxiter=10, callback=lambda _: its.__setitem__(0, its[0] + 1),
                   callback_type="pr_norm")
        total[name] += its[0]
        print(f"  system {s_}: {name:64s} {its[0]:3d} outer steps", flush=True)
print("\nouter Krylov steps over the time step's Newton systems, and cycles spent:")
for name, _, k in variants:
    print(f"  {name:64s} {total[name]:3d} steps, {total[name] * k:3d} cycles")
