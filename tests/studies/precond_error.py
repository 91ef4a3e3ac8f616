"""Error (not residual) of right-preconditioned GMRES iterates against the direct solution, per
field, for the two orders of the field split and with the potential row scaled in the Krylov
norm.  python tests/studies/precond_error.py [n=288] [tag=late]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.argv = sys.argv[:3] + ["noexec"]
src = open(os.path.join(ROOT, "tests", "studies", "precond_structure.py")).read()
g = {"__file__": os.path.join(ROOT, "tests", "studies", "precond_structure.py")}
exec(src[:src.index("def study(J, F, label):")], g)
systems, iu, ip, N, nv, V11, lpp, cb = g["systems"], g["iu"], g["ip"], g["N"], g["nv"], g["V11"], g["lpp"], g["cb"]


def fgmres_iterates(J, b, M, kmax, scale=None):
    """x_k, k = 1..kmax, of right-preconditioned GMRES on (S J) x = S b (S = diag(scale))"""
    S = np.ones(N) if scale is None else scale
    bs = S * b
    beta = np.linalg.norm(bs)
    V = [bs / beta]
    Zs, H = [], np.zeros((kmax + 1, kmax))
    out = []
    for k in range(kmax):
        z = M(V[k] / S)        # Minv acts on the unscaled residual direction
        w = S * (J @ z)
        Zs.append(z)
        for i in range(k + 1):
            H[i, k] = V[i] @ w
            w = w - H[i, k] * V[i]
        H[k + 1, k] = np.linalg.norm(w)
        V.append(w / H[k + 1, k])
        e1 = np.zeros(k + 2); e1[0] = beta
        y, *_ = np.linalg.lstsq(H[:k + 2, :k + 1], e1, rcond=None)
        x = sum(yi * zi for yi, zi in zip(y, Zs))
        res = np.linalg.norm(e1 - H[:k + 2, :k + 1] @ y) / beta
        out.append((x, res))
    return out


for ksys in (0, nsys_ := len(systems) - 1):
    J, F = systems[ksys]
    b = -F
    xs = spla.splu(J.tocsc()).solve(b)
    Juu, Jup, Jpu = J[iu][:, iu].tocsr(), J[iu][:, ip].tocsr(), J[ip][:, iu].tocsr()
    D = sp.block_diag([np.linalg.inv(Juu[2 * v:2 * v + 2, 2 * v:2 * v + 2].toarray()) for v in range(nv)]).tocsr()

    def cheb(ru, deg):
        w = cb.chebyshev_weights(deg)
        gg = D @ ru
        z = w[0] * gg
        for k in range(1, deg):
            z = z + w[k] * (gg - D @ (Juu @ z))
        return z

    def join(zu, zp):
        z = np.empty(N); z[iu], z[ip] = zu, zp
        return z

    def lower(deg):
        def f(r):
            zu = cheb(r[iu], deg)
            return join(zu, V11(r[ip] - Jpu @ zu))
        return f

    def upper(deg):
        def f(r):
            zp = V11(r[ip])
            return join(cheb(r[iu] - Jup @ zp, deg), zp)
        return f

    def sym(deg):   # lower, then the upper factor with the same species polynomial
        def f(r):
            zu = cheb(r[iu], deg)
            zp = V11(r[ip] - Jpu @ zu)
            return join(zu - cheb(Jup @ zp, deg), zp)
        return f

    dj = np.abs(J.diagonal())
    s_row = np.ones(N)
    s_row[ip] = np.median(dj[iu]) / np.median(dj[ip])
    print(f"Newton system {ksys}: potential-row scale {s_row[ip][0]:.2e}; max|x*| per field",
          [f"{np.abs(xs[c::3]).max():.2e}" for c in range(3)], flush=True)
    for name, M, sc in (("lower Chebyshev(4)  [round 1]", lower(4), None), ("lower Chebyshev(6)", lower(6), None),
                        ("upper Chebyshev(6)", upper(6), None), ("upper Chebyshev(8)", upper(8), None),
                        ("lower Chebyshev(6), potential row scaled", lower(6), s_row),
                        ("upper Chebyshev(6), potential row scaled", upper(6), s_row),
                        ("upper Chebyshev(8), potential row scaled", upper(8), s_row),
                        ("lower+upper factor Chebyshev(4), scaled", sym(4), s_row)):
        its = fgmres_iterates(J, b, M, 14, sc)
        rows = []
        for k, (x, res) in enumerate(its):
            err = [np.abs(x[c::3] - xs[c::3]).max() / np.abs(xs[c::3]).max() for c in range(3)]
            rows.append((k + 1, res, err))
        k5 = next((k for k, r, e in rows if r <= 1e-5), None)
        print(f"  {name:44s} stops (residual 1e-5) at step {k5}", flush=True)
        for k, r, e in rows:
            if k in (2, 3, 4, 6, 8, 10, 12, 14) or k == k5:
                print(f"      step {k:2d}: residual {r:.1e}  error ions {e[0]:.1e} electrons {e[1]:.1e} potential {e[2]:.1e}"
                      + ("   <- stop" if k == k5 else ""), flush=True)
