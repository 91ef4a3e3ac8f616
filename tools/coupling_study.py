"""Which block couplings does the field split need late in the streamer run?  Takes the Jacobian of
a late state from the device (scalar CSR), forms block preconditioners with EXACT block solves on
the host (SuperLU) and counts right-preconditioned GMRES steps to 1e-5.  (Host-heavy: minutes of
CPU time on the GPU box for n = 192.)
python tools/coupling_study.py [n=192] [steps=250]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from fedm_amd.cases import streamer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 250
msh = streamer.mesh(n, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
st = streamer.Stepper(prob)
st.initialise()
for k in range(steps):
    l0, n0 = st.linear_iterations, st.newton_iterations
    st.step()
    if k % 50 == 0 or k == steps - 1:
        print(f"step {k}: newton {st.newton_iterations - n0} gmres {st.linear_iterations - l0}", flush=True)
# Jacobian and residual at the start of the next step's first Newton iteration
prob.shift_state()
prob.set_step(st.dt.time_step, st.dt_old.time_step)
F, fn = prob.residual()
prob.jacobian()
J = prob.jacobian_csr().tocsr()
N = J.shape[0]
nv = N // 3
iu = np.array([3 * v + s for v in range(nv) for s in (0, 1)])
ip = np.arange(2, N, 3)
Juu, Jup, Jpu, Jpp = J[iu][:, iu].tocsc(), J[iu][:, ip].tocsc(), J[ip][:, iu].tocsc(), J[ip][:, ip].tocsc()
print("blocks", Juu.shape, Jpp.shape, "|F|", fn, flush=True)
luu, lpp = spla.splu(Juu), spla.splu(Jpp)
D = sp.block_diag([np.linalg.inv(Juu[2 * v:2 * v + 2, 2 * v:2 * v + 2].toarray()) for v in range(nv)]).tocsr()
S = (Jpp - Jpu @ D @ Jup).tocsc()
ls = spla.splu(S)
ratio = np.abs((Jpu @ D @ Jup).diagonal()) / np.abs(Jpp.diagonal())
print("Schur correction / Jpp on the diagonal: max %.3g, 99th percentile %.3g, median %.3g"
      % (ratio.max(), np.percentile(ratio, 99), np.median(ratio)))


xs = spla.splu(J.tocsc()).solve(-F)   # the direct solve the reference would do


def run(name, apply):
    its = [0]
    def cb(_):
        its[0] += 1
    # right preconditioning: solve (J M) y = b in the plain residual norm, as the device does
    A = spla.LinearOperator((N, N), matvec=lambda y: J @ apply(y))
    y, info = spla.gmres(A, -F, rtol=1e-5, restart=60, maxiter=5, callback=cb, callback_type="pr_norm")
    x = apply(y)
    err = [np.abs(x[c::3] - xs[c::3]).max() / max(np.abs(xs).max(), 1e-300) for c in range(3)]
    print(f"{name:58s} {its[0]:3d} steps; max error per component / max|direct solution|: "
          + " ".join(f"{e:.1e}" for e in err), flush=True)


def split(r):
    return r[iu], r[ip]


def join(zu, zp):
    z = np.empty(N)
    z[iu], z[ip] = zu, zp
    return z


def lower(r, schur=False, upper=False, species=None):
    ru, rp = split(r)
    su = species or luu.solve
    zu = su(ru)
    zp = (ls if schur else lpp).solve(rp - Jpu @ zu)
    if upper:
        zu = zu - su(Jup @ zp)
    return join(zu, zp)


def upper_tri(r):
    ru, rp = split(r)
    zp = lpp.solve(rp)
    zu = luu.solve(ru - Jup @ zp)
    return join(zu, zp)


def cheb(ru, deg=4):
    from fedm_amd.device import chebyshev_weights
    w = chebyshev_weights(deg)
    g = D @ ru
    z = w[0] * g
    for k in range(1, deg):
        z = z + w[k] * (g - D @ (Juu @ z))
    return z


run("block diagonal, exact blocks", lambda r: join(luu.solve(split(r)[0]), lpp.solve(split(r)[1])))
run("lower triangular (shipped structure), exact blocks", lower)
run("upper triangular, exact blocks", upper_tri)
run("lower + upper factor (LDU with S = Jpp), exact blocks", lambda r: lower(r, upper=True))
run("lower, Schur S = Jpp - Jpu D^-1 Jup, exact blocks", lambda r: lower(r, schur=True))
run("LDU with that Schur complement", lambda r: lower(r, schur=True, upper=True))
run("lower, species by Chebyshev(4) block Jacobi, Jpp exact", lambda r: lower(r, species=cheb))
run("lower, Chebyshev(6), Jpp exact", lambda r: lower(r, species=lambda x: cheb(x, 6)))
run("lower, Chebyshev(4), Schur", lambda r: lower(r, schur=True, species=cheb))
run("LDU, Chebyshev(4), Schur", lambda r: lower(r, schur=True, upper=True, species=cheb))
