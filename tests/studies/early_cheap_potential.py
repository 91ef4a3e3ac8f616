"""Early in the run the potential block is easy: does the species-first field split need the whole
V-cycle there?  Error and residual of the iterates with cheaper potential solves.
python tests/studies/early_cheap_potential.py [n=288] [tag=early]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.argv = (sys.argv + ["288", "early"])[:3] + ["noexec"]
src = open(os.path.join(ROOT, "tests", "studies", "precond_structure.py")).read()
g = {"__file__": os.path.join(ROOT, "tests", "studies", "precond_structure.py")}
exec(src[:src.index("def study(J, F, label):")], g)
systems, iu, ip, N, nv, cb = g["systems"], g["iu"], g["ip"], g["N"], g["nv"], g["cb"]
V11, V01, V10, truncated, jacobi_only, lpp = g["V11"], g["V01"], g["V10"], g["truncated"], g["jacobi_only"], g["lpp"]
e = {}
exec(open(os.path.join(ROOT, "tests", "studies", "precond_error.py")).read().split("for ksys in")[0].split("systems, iu, ip")[0].replace("exec(src", "pass #"), e) if False else None


def fgmres_iterates(J, b, M, kmax):
    beta = np.linalg.norm(b)
    V = [b / beta]
    Zs, H = [], np.zeros((kmax + 1, kmax))
    out = []
    for k in range(kmax):
        z = M(V[k])
        w = J @ z
        Zs.append(z)
        for i in range(k + 1):
            H[i, k] = V[i] @ w
            w = w - H[i, k] * V[i]
        H[k + 1, k] = np.linalg.norm(w)
        V.append(w / H[k + 1, k])
        e1 = np.zeros(k + 2); e1[0] = beta
        y, *_ = np.linalg.lstsq(H[:k + 2, :k + 1], e1, rcond=None)
        x = sum(yi * zi for yi, zi in zip(y, Zs))
        out.append((x, np.linalg.norm(e1 - H[:k + 2, :k + 1] @ y) / beta))
    return out


for ksys, (J, F) in enumerate(systems):
    b = -F
    xs = spla.splu(J.tocsc()).solve(b)
    Juu, Jup, Jpu = J[iu][:, iu].tocsr(), J[iu][:, ip].tocsr(), J[ip][:, iu].tocsr()
    D = sp.block_diag([np.linalg.inv(Juu[2 * v:2 * v + 2, 2 * v:2 * v + 2].toarray()) for v in range(nv)]).tocsr()

    def cheb(ru, deg=6):
        w = cb.chebyshev_weights(deg)
        gg = D @ ru
        z = w[0] * gg
        for k in range(1, deg):
            z = z + w[k] * (gg - D @ (Juu @ z))
        return z

    def lower(pot):
        def f(r):
            zu = cheb(r[iu])
            z = np.empty(N); z[iu] = zu; z[ip] = pot(r[ip] - Jpu @ zu)
            return z
        return f
    print(f"Newton system {ksys}: max|x*| per field", [f"{np.abs(xs[c::3]).max():.2e}" for c in range(3)], flush=True)
    for name, pot in (("V(1,1)  [shipped]", V11), ("exact", lpp.solve), ("V(0,1)", V01), ("V(1,0)", V10),
                      ("two levels, 2 Jacobi sweeps on level 1", truncated(2, 2)),
                      ("2 Jacobi sweeps, no multigrid", jacobi_only(2)), ("1 Jacobi sweep", jacobi_only(1)),
                      ("nothing (z_phi = 0)", lambda r: 0.0 * r)):
        its = fgmres_iterates(J, b, lower(pot), 6)
        line = []
        for k, (x, res) in enumerate(its):
            err = [np.abs(x[c::3] - xs[c::3]).max() / np.abs(xs[c::3]).max() for c in range(3)]
            line.append(f"k={k + 1}: res {res:.0e} err {max(err[:2]):.0e}/{err[2]:.0e}")
        k5 = next((k + 1 for k, (x, r) in enumerate(its) if r <= 1e-5), None)
        print(f"  {name:40s} stops at {k5};  " + "  ".join(line[:5]), flush=True)
