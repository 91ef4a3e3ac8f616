"""Stores the Newton systems (J_k, F_k) of one late time step of the streamer case (CPU statement),
for preconditioner studies: python tests/studies/late_systems.py [n=288] [steps=200] -> gpurun_out/late_<n>.npz"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from oracle import cpu_backend as cb, controller, streamer as ost

n = int(sys.argv[1]) if len(sys.argv) > 1 else 288
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tag = sys.argv[3] if len(sys.argv) > 3 else "late"
prob, mesh = cb.streamer_problem(n, 4.0)
U = np.zeros((mesh.nv, 3))
U[:, 0], U[:, 1] = ost.initial_log_densities(mesh.coords)
prob.set_state(U, U, U)
prob.setup_multigrid()
prob.poisson_solve()
U = prob.get_state()
st = controller.StepState(5e-12, 1e30, n_error=2)
t = 0.0
U_old, U_old1 = U.copy(), U.copy()
cnt = [0, 0]


def solve(Uw, dt, dt_old):
    prob.set_state(Uw, U_old, U_old1)
    its, lits = prob.newton_solve(dt, dt_old, 1e-4, 20)
    cnt[0] += its; cnt[1] += lits
    Uw[:] = prob.get_state()


t0 = time.time()
for k in range(steps):
    U_old1[:] = U_old
    U_old[:] = U
    c0 = list(cnt)
    t = controller.adaptive_solve(solve, U, U_old, t, st, 1e-3, 1e-15, error_component=1)
    st.dt_old = st.dt
    st.dt = controller.adaptive_timestep(st.dt, st.max_error, 1e-3, 1e-15, 5e-12)
    st.max_error[2] = st.max_error[1]
    st.max_error[1] = st.max_error[0]
    if k % 20 == 0 or k == steps - 1:
        print(f"step {k} t={t:.3e} newton {cnt[0]-c0[0]} gmres {cnt[1]-c0[1]} ({time.time()-t0:.0f} s)", flush=True)
# the next step's Newton systems, with direct solves
U_old1[:] = U_old
U_old[:] = U
out = {}
Uw = U.copy()
f0 = None
for k in range(6):
    prob.set_state(Uw, U_old, U_old1)
    F, J = prob.residual_jacobian(st.dt, st.dt_old)
    fn = np.linalg.norm(F)
    f0 = f0 or fn
    print(f"newton {k}: |F| = {fn:.4e} ({fn / f0:.2e} of the first)", flush=True)
    if fn < 1e-4 * f0:
        break
    J = J.tocsr()
    out[f"J{k}_data"], out[f"J{k}_indices"], out[f"J{k}_indptr"], out[f"F{k}"] = J.data, J.indices, J.indptr, F
    d = spla.splu(J.tocsc()).solve(-F)
    Uw = Uw + d.reshape(-1, 3)
out["n_systems"] = k
out["coords"] = mesh.coords
out["dirichlet_dofs"] = prob.model.dirichlet_dofs
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez(os.path.join(ROOT, "gpurun_out", f"{tag}_{n}.npz"), **out)
print("saved", k, "systems")
