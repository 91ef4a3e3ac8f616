"""Convergence quality of V-cycle variants on the constant potential block of the bench mesh
(CPU study; the hierarchy is the one the device installs).  Prints the asymptotic convergence
factor of the stationary iteration and the preconditioned-CG step count to 1e-5 / 1e-10.
python tests/studies/mg_quality.py [n=288]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from oracle import cpu_backend as cb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 288
prob, mesh = cb.streamer_problem(n, 4.0)
A = prob._potential_block().tocsr()
nv = A.shape[0]
fixed = np.zeros(nv, dtype=bool)
d = prob.model.dirichlet_dofs
fixed[d[d % 3 == 2] // 3] = True
print("rows", nv, "fixed", fixed.sum(), flush=True)


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
        import ctypes as C
        indptr = np.ascontiguousarray(Af.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(Af.indices, dtype=np.int32)
        nagg = prob.lib.cpu_aggregate(Af.shape[0], indptr.ctypes.data_as(C.POINTER(C.c_int64)), cb._ip(indices),
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


def make_cycle(levels, nus, omega=0.85, cheb=False):
    """nus[l] = (pre, post) damped-Jacobi sweeps on level l (cheb: Chebyshev polynomial of that
    degree on [lam_max/4?]...)."""
    lu = spla.splu(levels[-1][0].tocsc())
    dinv = [1.0 / a.diagonal() for a, _ in levels]
    lam = []
    for (a, _), di in zip(levels, dinv):
        # largest eigenvalue of D^-1 A by a few power iterations
        x = np.random.default_rng(0).standard_normal(a.shape[0])
        for _ in range(30):
            x = di * (a @ x)
            x /= np.linalg.norm(x)
        lam.append(float(x @ (di * (a @ x))))

    def smooth(l, x, b, k):
        a = levels[l][0]
        if not cheb:
            for _ in range(k):
                x = x + omega * dinv[l] * (b - a @ x)
            return x
        if k == 0:
            return x
        # Chebyshev on [lam/cheb_lo, 1.1 lam]
        lmax, lmin = 1.1 * lam[l], 1.1 * lam[l] / 6.0
        theta_, delta = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        sigma = theta_ / delta
        rho_ = 1.0 / sigma
        r = dinv[l] * (b - a @ x)
        dvec = r / theta_
        x = x + dvec
        for _ in range(k - 1):
            rho_new = 1.0 / (2.0 * sigma - rho_)
            r = dinv[l] * (b - a @ x)
            dvec = rho_new * rho_ * dvec + 2.0 * rho_new / delta * r
            x = x + dvec
            rho_ = rho_new
        return x

    def cyc(l, b):
        if l == len(levels) - 1:
            return lu.solve(b)
        a, p = levels[l]
        x = smooth(l, np.zeros_like(b), b, nus[l][0])
        r = b - a @ x
        x = x + p @ cyc(l + 1, p.T @ r)
        return smooth(l, x, b, nus[l][1])
    return lambda b: cyc(0, b), lam


def quality(name, M):
    rng = np.random.default_rng(1)
    # stationary iteration on A x = 0 from a random start: error contraction per cycle
    x = rng.standard_normal(nv)
    x[fixed] = 0.0
    e0 = np.linalg.norm(x)
    fac = []
    for k in range(25):
        x = x - M(A @ x)
        e1 = np.linalg.norm(x)
        fac.append(e1 / e0)
        e0 = e1
    b = rng.standard_normal(nv)
    res = []
    spla.cg(A, b, rtol=1e-10, maxiter=200, M=spla.LinearOperator((nv, nv), matvec=M),
            callback=lambda xk: res.append(np.linalg.norm(b - A @ xk) / np.linalg.norm(b)))
    k5 = next((i + 1 for i, r in enumerate(res) if r < 1e-5), None)
    print(f"{name:52s} factor {np.mean(fac[-5:]):.3f}  CG steps to 1e-5: {k5}  to 1e-10: {len(res)}", flush=True)


t0 = time.time()
lv = hierarchy(A, fixed)
print("levels", [a.shape[0] for a, _ in lv], "nnz/row", [round(a.nnz / a.shape[0], 1) for a, _ in lv],
      f"({time.time() - t0:.1f} s)", flush=True)
nl = len(lv) - 1
for om in (0.85,):
    M, lam = make_cycle(lv, [(1, 1)] * nl, omega=om)
    print("lambda_max(D^-1 A) per level", [round(v, 3) for v in lam])
    quality(f"V(1,1) Jacobi {om} (shipped)", M)
    quality("V(2,2) Jacobi everywhere", make_cycle(lv, [(2, 2)] * nl, omega=om)[0])
    quality("V(3,3) Jacobi everywhere", make_cycle(lv, [(3, 3)] * nl, omega=om)[0])
    quality("fine V(1,1), coarser levels (2,2)", make_cycle(lv, [(1, 1)] + [(2, 2)] * (nl - 1), omega=om)[0])
    quality("fine V(1,1), coarser levels (3,3)", make_cycle(lv, [(1, 1)] + [(3, 3)] * (nl - 1), omega=om)[0])
    quality("fine V(2,2), coarser levels (1,1)", make_cycle(lv, [(2, 2)] + [(1, 1)] * (nl - 1), omega=om)[0])
    quality("fine V(1,2), coarser (3,3)", make_cycle(lv, [(1, 2)] + [(3, 3)] * (nl - 1), omega=om)[0])
quality("Chebyshev V(1,1)", make_cycle(lv, [(1, 1)] * nl, cheb=True)[0])
quality("Chebyshev V(2,2)", make_cycle(lv, [(2, 2)] * nl, cheb=True)[0])
quality("Chebyshev V(3,3)", make_cycle(lv, [(3, 3)] * nl, cheb=True)[0])
quality("Chebyshev fine (1,1), coarser (3,3)", make_cycle(lv, [(1, 1)] + [(3, 3)] * (nl - 1), cheb=True)[0])
for th in (0.02, 0.15, 0.25):
    l2 = hierarchy(A, fixed, theta=th)
    print("theta", th, "levels", [a.shape[0] for a, _ in l2])
    quality(f"V(1,1) Jacobi .85 theta={th}", make_cycle(l2, [(1, 1)] * (len(l2) - 1))[0])
print("--- fine-level variants (coarser levels V(1,1) Jacobi)")
for pre, post in ((2, 1), (1, 2), (2, 2), (3, 3)):
    quality(f"Jacobi fine ({pre},{post})", make_cycle(lv, [(pre, post)] + [(1, 1)] * (nl - 1))[0])
    quality(f"Chebyshev fine ({pre},{post})", make_cycle(lv, [(pre, post)] + [(1, 1)] * (nl - 1), cheb=True)[0])
for op in (0.8, 1.0, 1.6, 2.0):
    l2 = hierarchy(A, fixed, omega_p=op)
    quality(f"V(1,1) prolongator damping {op}", make_cycle(l2, [(1, 1)] * (len(l2) - 1))[0])
for om in (0.6, 0.7, 1.0):
    quality(f"V(1,1) Jacobi omega {om}", make_cycle(lv, [(1, 1)] * nl, omega=om)[0])
