"""Does a Schur-complement treatment of the electron-potential coupling cut the outer Krylov count on the Newton systems
of a late streamer step?  S^ = J_pp - J_pu D_uu^-1 J_up (D_uu: the 2x2 vertex blocks of the species block, or its
diagonal), used in the block factorisation  zp = S^-1 (r_p - J_pu Juu~^-1 r_u),  zu = Juu~^-1 (r_u - J_up zp).
python tests/studies/schur_complement.py [n=288]   (needs gpurun_out/late_<n>.npz from late_systems.py)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "288", "late", "noexec"]
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import precond_structure as ps
from oracle import cpu_backend as cb

N, nv, iu, ip, Jpp0 = ps.N, ps.nv, ps.iu, ps.ip, ps.Jpp0


def galerkin(levels0, A):
    """the hierarchy of `levels0` (fixed prolongators) with the operators recomputed from A"""
    out = []
    for _, P in levels0:
        out.append((A, P))
        if P is not None:
            A = (P.T @ A @ P).tocsr()
    return out


def study(J, F, label):
    Juu, Jup, Jpu = J[iu][:, iu].tocsr(), J[iu][:, ip].tocsr(), J[ip][:, iu].tocsr()
    Jpp = J[ip][:, ip].tocsr()
    blocks = [np.linalg.inv(Juu[2 * v:2 * v + 2, 2 * v:2 * v + 2].toarray()) for v in range(nv)]
    D = sp.block_diag(blocks).tocsr()
    Dd = sp.diags(1.0 / Juu.diagonal())
    luu = spla.splu(Juu.tocsc())
    lpp = spla.splu(Jpp.tocsc())
    xs = spla.splu(J.tocsc()).solve(-F)

    def cheb(ru, deg):
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
        eu = np.linalg.norm((x - xs)[iu]) / np.linalg.norm(xs[iu])
        ep = np.linalg.norm((x - xs)[ip]) / np.linalg.norm(xs[ip])
        print(f"  {label} {name:74s} {its[0]:3d} steps (true residual {true:.1e}; error species {eu:.1e}, potential {ep:.1e})", flush=True)
        return its[0]

    def lower(sp_solve, pot):
        def f(r):
            zu = sp_solve(r[iu])
            return join(zu, pot(r[ip] - Jpu @ zu))
        return f

    def schur(sp_solve, spot, sp_solve2=None):
        sp2 = sp_solve2 or sp_solve
        def f(r):
            zp = spot(r[ip] - Jpu @ sp_solve(r[iu]))
            return join(sp2(r[iu] - Jup @ zp), zp)
        return f

    def schur_upper(sp_solve, spot):
        """upper factor only: zp = S^-1 r_p (no species solve in front)"""
        def f(r):
            zp = spot(r[ip])
            return join(sp_solve(r[iu] - Jup @ zp), zp)
        return f

    S_blk = (Jpp - Jpu @ D @ Jup).tocsr()
    S_dia = (Jpp - Jpu @ Dd @ Jup).tocsr()
    lump = sp.diags(np.asarray(Jpu.sum(axis=1)).ravel())     # row sums of the coupling: one electron + one ion entry
    print(f"  {label} nnz: Jpp {Jpp.nnz}, S(block) {S_blk.nnz}, S(diag) {S_dia.nnz}; |S-Jpp|/|Jpp| = "
          f"{spla.norm(S_blk - Jpp) / spla.norm(Jpp):.2e}; asym |S-S^T|/|S| = {spla.norm(S_blk - S_blk.T) / spla.norm(S_blk):.2e}", flush=True)
    lS_blk = spla.splu(S_blk.tocsc())
    lS_dia = spla.splu(S_dia.tocsc())
    c4 = lambda x: cheb(x, 4)
    c6 = lambda x: cheb(x, 6)

    run("lower: exact species, exact potential                      [floor of the shipped split]", lower(luu.solve, lpp.solve))
    run("lower: Chebyshev(6), V(1,1)                                 [shipped]", lower(c6, ps.V11))
    run("Schur(block D): exact species, exact S^", schur(luu.solve, lS_blk.solve))
    run("Schur(diag D): exact species, exact S^", schur(luu.solve, lS_dia.solve))
    run("Schur(block D): Chebyshev(6), exact S^", schur(c6, lS_blk.solve))
    run("Schur(block D): Chebyshev(4), exact S^", schur(c4, lS_blk.solve))
    run("Schur(block D): D^-1 in front, Chebyshev(6) behind, exact S^", schur(lambda x: D @ x, lS_blk.solve, c6))
    run("Schur upper only (block D): exact S^, then Chebyshev(6)", schur_upper(c6, lS_blk.solve))
    run("Schur upper only (block D): exact S^, then exact species", schur_upper(luu.solve, lS_blk.solve))
    # multigrid on S^: aggregates of the constant block, operators re-formed
    lvS = galerkin(ps.lv, S_blk)
    nl = len(lvS) - 1
    VS11 = ps.make_cycle(lvS, [(1, 1)] * nl)
    VS22 = ps.make_cycle(lvS, [(2, 2)] * nl)
    run("Schur(block D): Chebyshev(6), V(1,1) on S^ (fixed prolongators, Galerkin)", schur(c6, VS11))
    run("Schur(block D): Chebyshev(6), V(2,2) on S^ (fixed prolongators, Galerkin)", schur(c6, VS22))
    run("Schur upper only: V(1,1) on S^ (fixed prolongators), then Chebyshev(6)", schur_upper(c6, VS11))
    lvR = ps.hierarchy(S_blk, ps.fixed)
    VR11 = ps.make_cycle(lvR, [(1, 1)] * (len(lvR) - 1))
    run("Schur(block D): Chebyshev(6), V(1,1) on S^ (hierarchy rebuilt)", schur(c6, VR11))
    # S^ on the finest level only, coarse levels from the constant block
    def mixed(b):
        a, p = S_blk, ps.lv[0][1]
        d = 1.0 / a.diagonal()
        x = 0.85 * d * b
        rc = p.T @ (b - a @ x)
        sub = ps.make_cycle(ps.lv[1:], [(1, 1)] * (len(ps.lv) - 2)) if not hasattr(mixed, "sub") else mixed.sub
        mixed.sub = sub
        x = x + p @ sub(rc)
        return x + 0.85 * d * (b - a @ x)
    run("Schur(block D): Chebyshev(6), V(1,1) with S^ on level 0 only", schur(c6, mixed))
    # how accurate must the inner solve be?  k stationary cycles on S^ (level 0 only); work = cycles and species sweeps
    def repeated(cycle, A, k):
        def f(b):
            x = cycle(b)
            for _ in range(k - 1):
                x = x + cycle(b - A @ x)
            return x
        return f
    for k in (2, 3, 4):
        n_ = run(f"Schur(block D): Chebyshev(6), {k} cycles with S^ on level 0 only", schur(c6, repeated(mixed, S_blk, k)))
        print(f"      work: {n_ * k} cycles, {2 * 6 * n_} species sweeps (shipped: 1 cycle and 6 sweeps per step)")
        n_ = run(f"Schur(block D): Chebyshev(6) in front, (4) behind, {k} cycles, level 0 only", schur(c6, repeated(mixed, S_blk, k), c4))
    # the coupling lumped to the diagonal: S^ keeps the pattern of the potential block
    Ll = sp.diags(np.asarray(np.abs(Jpu).sum(axis=1)).ravel())
    for sname, col in (("electron", 0), ("ion", 1)):
        pass
    sel = [sp.csr_matrix((np.ones(nv), (np.arange(nv), 2 * np.arange(nv) + c)), shape=(nv, 2 * nv)) for c in (0, 1)]
    S_lmp = Jpp.copy()
    for c in (0, 1):
        lc = np.asarray((Jpu @ sel[c].T).sum(axis=1)).ravel()          # lumped coupling of species c
        dc = 1.0 / (sel[c] @ Juu @ sel[c].T).diagonal()
        S_lmp = S_lmp - sp.diags(lc * dc) @ (sel[c] @ Jup)
    S_lmp = S_lmp.tocsr()
    print(f"  {label} lumped S^: nnz {S_lmp.nnz}; |S_lumped - S_block| / |S_block - Jpp| = {spla.norm(S_lmp - S_blk) / spla.norm(S_blk - Jpp):.2e}")
    lS_lmp = spla.splu(S_lmp.tocsc())
    run("Schur(lumped, pattern of Jpp): Chebyshev(6), exact S^", schur(c6, lS_lmp.solve))


for k, (J, F) in list(enumerate(ps.systems))[::2]:
    print(f"Newton system {k}: |F| = {np.linalg.norm(F):.3e}", flush=True)
    study(J, F, f"[{k}]")
