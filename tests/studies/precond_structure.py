"""Right-preconditioned GMRES step counts (rtol 1e-5 on the true residual) on the stored Newton
systems of a late streamer step for variants of the field-split preconditioner.
python tests/studies/precond_structure.py [n=288]   (needs gpurun_out/late_<n>.npz from late_systems.py)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from oracle import cpu_backend as cb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 288
tag = sys.argv[2] if len(sys.argv) > 2 else "late"
Z = np.load(os.path.join(ROOT, "gpurun_out", f"{tag}_{n}.npz"))
nsys = int(Z["n_systems"])
lib = cb.load()


def hierarchy(A, fixed, theta=0.08, omega_p=4.0 / 3.0, max_coarse=2000):
    levels, free = [], ~fixed
    while A.shape[0] > max_coarse:
        idx = np.nonzero(free)[0]
        Af = A[idx][:, idx].tocsr()
        dg = np.abs(Af.diagonal())
        rows = np.repeat(np.arange(Af.shape[0]), np.diff(Af.indptr))
        strong = (np.abs(Af.data) >= theta * np.sqrt(dg[rows] * dg[Af.indices])).astype(np.uint8)
        strong[Af.indices == rows] = 0
        agg = np.empty(Af.shape[0], dtype=np.int32)
        indptr = np.ascontiguousarray(Af.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(Af.indices, dtype=np.int32)
        nagg = lib.cpu_aggregate(Af.shape[0], indptr.ctypes.data_as(C.POINTER(C.c_int64)), cb._ip(indices),
                                 strong.ctypes.data_as(C.POINTER(C.c_uint8)), cb._ip(agg))
        if nagg >= 0.8 * idx.size:
            break
        T = sp.csr_matrix((np.ones(idx.size), (idx, agg)), shape=(A.shape[0], nagg))
        DinvA = sp.diags(1.0 / A.diagonal()) @ A
        rho = np.abs(DinvA).sum(axis=1).max()
        P = (sp.diags(free.astype(np.float64)) @ (T - (omega_p / rho) * (DinvA @ T))).tocsr()
        P.eliminate_zeros()
        levels.append((A, P))
        A = (P.T @ A @ P).tocsr()
        free = np.ones(A.shape[0], dtype=bool)
    levels.append((A, None))
    return levels


def make_cycle(levels, nus, omega=0.85, cheb_frac=None):
    lu = spla.splu(levels[-1][0].tocsc())
    dinv = [1.0 / a.diagonal() for a, _ in levels]
    lam = []
    for (a, _), di in zip(levels, dinv):
        x = np.random.default_rng(0).standard_normal(a.shape[0])
        for _ in range(30):
            x = di * (a @ x)
            x /= np.linalg.norm(x)
        lam.append(float(x @ (di * (a @ x))))

    def smooth(l, x, b, k):
        a = levels[l][0]
        if cheb_frac is None:
            for _ in range(k):
                x = x + omega * dinv[l] * (b - a @ x)
            return x
        if k == 0:
            return x
        lmax, lmin = 1.1 * lam[l], 1.1 * lam[l] / cheb_frac
        th, de = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = th / de
        rho_ = 1.0 / sigma
        r = dinv[l] * (b - a @ x)
        dv = r / th
        x = x + dv
        for _ in range(k - 1):
            rn = 1.0 / (2.0 * sigma - rho_)
            r = dinv[l] * (b - a @ x)
            dv = rn * rho_ * dv + 2.0 * rn / de * r
            x = x + dv
            rho_ = rn
        return x

    def cyc(l, b):
        if l == len(levels) - 1:
            return lu.solve(b)
        a, p = levels[l]
        x = smooth(l, np.zeros_like(b), b, nus[l][0])
        x = x + p @ cyc(l + 1, p.T @ (b - a @ x))
        return smooth(l, x, b, nus[l][1])
    return lambda b: cyc(0, b)


systems = []
for k in range(nsys):
    J = sp.csr_matrix((Z[f"J{k}_data"], Z[f"J{k}_indices"], Z[f"J{k}_indptr"]))
    systems.append((J, Z[f"F{k}"]))
N = systems[0][0].shape[0]
nv = N // 3
iu = np.array([3 * v + s for v in range(nv) for s in (0, 1)])
ip = np.arange(2, N, 3)
fixed = np.zeros(nv, dtype=bool)
d = Z["dirichlet_dofs"]
fixed[d[d % 3 == 2] // 3] = True
Jpp0 = systems[0][0][ip][:, ip].tocsr()
lv = hierarchy(Jpp0, fixed)
print("levels", [a.shape[0] for a, _ in lv], flush=True)
nl = len(lv) - 1
V11 = make_cycle(lv, [(1, 1)] * nl)
V22 = make_cycle(lv, [(2, 2)] * nl)
C22 = make_cycle(lv, [(2, 2)] * nl, cheb_frac=8.0)
C33 = make_cycle(lv, [(3, 3)] * nl, cheb_frac=10.0)
lpp = spla.splu(Jpp0.tocsc())
V10 = make_cycle(lv, [(1, 0)] * nl)
V01 = make_cycle(lv, [(0, 1)] * nl)
dinv0 = 1.0 / Jpp0.diagonal()


def truncated(levels_kept, sweeps):
    """V(1,1) on the first levels; the last kept level gets `sweeps` damped-Jacobi sweeps instead of a solve"""
    sub = lv[:levels_kept]
    A_last = lv[levels_kept - 1][0]
    dl = 1.0 / A_last.diagonal()

    def cyc(l, b):
        a, p = lv[l]
        if l == levels_kept - 1:
            x = 0.85 * dl * b
            for _ in range(sweeps - 1):
                x = x + 0.85 * dl * (b - a @ x)
            return x
        d = 1.0 / a.diagonal()
        x = 0.85 * d * b
        x = x + p @ cyc(l + 1, p.T @ (b - a @ x))
        return x + 0.85 * d * (b - a @ x)
    return lambda b: cyc(0, b)


def jacobi_only(k):
    def f(b):
        x = 0.85 * dinv0 * b
        for _ in range(k - 1):
            x = x + 0.85 * dinv0 * (b - Jpp0 @ x)
        return x
    return f


def study(J, F, label):
    Juu, Jup, Jpu = J[iu][:, iu].tocsr(), J[iu][:, ip].tocsr(), J[ip][:, iu].tocsr()
    blocks = [np.linalg.inv(Juu[2 * v:2 * v + 2, 2 * v:2 * v + 2].toarray()) for v in range(nv)]
    D = sp.block_diag(blocks).tocsr()
    luu = spla.splu(Juu.tocsc())

    def cheb(ru, deg, stop_after=None):
        w = cb.chebyshev_weights(deg)
        g = D @ ru
        z = w[0] * g
        for k in range(1, deg):
            z = z + w[k] * (g - D @ (Juu @ z))
        return z

    def join(zu, zp):
        z = np.empty(N)
        z[iu], z[ip] = zu, zp
        return z

    def run(name, apply):
        its = [0]
        A = spla.LinearOperator((N, N), matvec=lambda y: J @ apply(y))
        y, info = spla.gmres(A, -F, rtol=1e-5, restart=30, maxiter=10,
                             callback=lambda _: its.__setitem__(0, its[0] + 1), callback_type="pr_norm")
        x = apply(y)
        true = np.linalg.norm(J @ x + F) / np.linalg.norm(F)
        print(f"  {label} {name:66s} {its[0]:3d} steps (true residual {true:.1e})", flush=True)
        return its[0]

    def lower(sp_solve, pot):
        def f(r):
            zu = sp_solve(r[iu])
            return join(zu, pot(r[ip] - Jpu @ zu))
        return f

    def approx_lower(deg, pot, first=1):
        """potential right-hand side from the first Chebyshev stage(s) only (so that the V-cycle
        can run beside the remaining sweeps)"""
        def f(r):
            w = cb.chebyshev_weights(deg)
            g = D @ r[iu]
            z = w[0] * g
            z0 = z.copy() if first == 1 else None
            for k in range(1, deg):
                z = z + w[k] * (g - D @ (Juu @ z))
                if k + 1 == first:
                    z0 = z.copy()
            return join(z, pot(r[ip] - Jpu @ z0))
        return f

    def diag(deg, pot):
        return lambda r: join(cheb(r[iu], deg), pot(r[ip]))

    def approx_upper(deg, pot):
        def f(r):
            zp = pot(r[ip])
            zu = cheb(r[iu], deg) - D @ (Jup @ zp)
            return join(zu, zp)
        return f

    def upper(deg, pot):
        def f(r):
            zp = pot(r[ip])
            return join(cheb(r[iu] - Jup @ zp, deg), zp)
        return f

    def ldu_approx(deg, pot):
        def f(r):
            zu = cheb(r[iu], deg)
            zp = pot(r[ip] - Jpu @ zu)
            return join(zu - D @ (Jup @ zp), zp)
        return f

    run("lower: exact species, exact potential", lower(luu.solve, lpp.solve))
    run("lower: Chebyshev(4), exact potential", lower(lambda x: cheb(x, 4), lpp.solve))
    run("lower: Chebyshev(4), V(1,1) Jacobi .85            [shipped]", lower(lambda x: cheb(x, 4), V11))
    run("lower: Chebyshev(6), V(1,1)", lower(lambda x: cheb(x, 6), V11))
    run("lower: Chebyshev(4), V(2,2) Jacobi", lower(lambda x: cheb(x, 4), V22))
    run("lower: Chebyshev(4), V(2,2) Chebyshev", lower(lambda x: cheb(x, 4), C22))
    run("lower: Chebyshev(4), V(3,3) Chebyshev", lower(lambda x: cheb(x, 4), C33))
    run("lower: Chebyshev(4), two V(1,1) cycles", lower(lambda x: cheb(x, 4), lambda b: (lambda x1: x1 + V11(b - Jpp0 @ x1))(V11(b))))
    run("block diagonal: Chebyshev(4), V(1,1)", diag(4, V11))
    run("approximate lower (coupling from stage 1): Chebyshev(4), V(1,1)", approx_lower(4, V11, 1))
    run("approximate lower (coupling from stage 2): Chebyshev(4), V(1,1)", approx_lower(4, V11, 2))
    run("upper: V(1,1), then Chebyshev(4)", upper(4, V11))
    run("upper: V(1,1), then Chebyshev(6)", upper(6, V11))
    run("upper: V(2,2) Chebyshev, then Chebyshev(4)", upper(4, C22))
    run("upper: V(2,2) Jacobi, then Chebyshev(4)", upper(4, V22))
    run("upper: exact potential, then Chebyshev(4)", upper(4, lpp.solve))
    run("upper: V(1,1), then Chebyshev(3)", upper(3, V11))
    run("upper: V(1,1), then Chebyshev(2)", upper(2, V11))
    run("upper: V(1,1), then Chebyshev(8)", upper(8, V11))
    run("upper: exact potential, then Chebyshev(8)", upper(8, lpp.solve))
    run("upper: V(1,0), then Chebyshev(8)", upper(8, V10))
    run("upper: V(0,1), then Chebyshev(8)", upper(8, V01))
    run("upper: V(2,2) Chebyshev, then Chebyshev(8)", upper(8, C22))
    run("upper: two levels, 2 Jacobi sweeps on level 1, then Chebyshev(8)", upper(8, truncated(2, 2)))
    run("upper: two levels, 4 Jacobi sweeps on level 1, then Chebyshev(8)", upper(8, truncated(2, 4)))
    if nl >= 2:
        run("upper: three levels, 2 Jacobi sweeps on level 2, then Chebyshev(8)", upper(8, truncated(3, 2)))
    run("upper: 2 Jacobi sweeps (no multigrid), then Chebyshev(8)", upper(8, jacobi_only(2)))
    run("upper: exact potential, exact species", lambda r: (lambda zp: join(luu.solve(r[iu] - Jup @ zp), zp))(lpp.solve(r[ip])))
    run("approximate upper (zu -= D^-1 Jup zp): Chebyshev(4), V(1,1)", approx_upper(4, V11))
    run("lower + approximate upper factor: Chebyshev(4), V(1,1)", ldu_approx(4, V11))
    run("lower + approximate upper factor: Chebyshev(4), exact potential", ldu_approx(4, lpp.solve))


for k, (J, F) in enumerate(systems if "noexec" not in sys.argv else []):
    print(f"Newton system {k}: |F| = {np.linalg.norm(F):.3e}", flush=True)
    study(J, F, f"[{k}]")
