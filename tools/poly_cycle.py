"""Polynomial-smoother V-cycle (fedm_amg_setup_poly): the Poisson-only CG count, and the bench run
late (from step 201) with the cycle as the alternative for hard systems.
python tools/poly_cycle.py [n=576] [late_start=200]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from fedm_amd import amg
from fedm_amd.cases import streamer
from fedm_amd.device import chebyshev_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
late = int(sys.argv[2]) if len(sys.argv) > 2 else 200
msh = streamer.mesh(n, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
U0 = np.zeros((prob.nv, 3))
U0[:, 0], U0[:, 1] = streamer.initial_log_densities(prob.coords)
for name, kw in (("V(1,1) Jacobi 0.85", dict(nu=1, omega=0.85)), ("Chebyshev degree 1", dict(poly_degree=1)),
                 ("Chebyshev degree 2, lmax/8", dict(poly_degree=2)), ("Chebyshev degree 2, lmax/5", dict(poly_degree=2, poly_fraction=5.0)),
                 ("Chebyshev degree 2, lmax/12", dict(poly_degree=2, poly_fraction=12.0)),
                 ("Chebyshev degree 3, lmax/10", dict(poly_degree=3, poly_fraction=10.0))):
    prob.set_state(U0, U0, U0)
    levels = prob.setup_multigrid(**kw)
    t0 = time.perf_counter()
    its = prob.poisson_solve(rtol=1e-12)
    dt = time.perf_counter() - t0
    t = min(prob.time_kernel(3, 50) for _ in range(3)) if hasattr(prob, "time_kernel") else 0.0
    print(f"{name:32s} levels {levels}: CG steps to 1e-12: {its:3d}  ({1e3 * dt:.2f} ms)", flush=True)

st = streamer.Stepper(prob)
st.initialise()
while st.t < late * 5e-12:
    st.step()
snap = (prob.get_state(), prob.get_state_old(), st.t, st.dt.time_step, st.dt_old.time_step, list(st.max_error), list(st.error))
print(f"late window (t = {st.t:.2e} s); every variant starts from the same stored state", flush=True)


def run(name, mg, deg, hard, steps=20, **kw):
    U, Uold, st.t, st.dt.time_step, st.dt_old.time_step, me, er = snap
    prob.set_state(U, Uold, Uold)
    st.max_error[:], st.error[:] = me, er
    prob.setup_multigrid(**mg)
    prob.set_fieldsplit(chebyshev_weights(deg), hard_weights=None if hard is None else chebyshev_weights(hard), **kw)
    st.step(); st.step()
    n0, l0 = st.newton_iterations, st.linear_iterations
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step()
    dt = time.perf_counter() - t0
    print(f"  {name:60s} newton/step {(st.newton_iterations - n0) / steps:5.2f}  gmres/step "
          f"{(st.linear_iterations - l0) / steps:6.2f}  ms/step {1e3 * dt / steps:7.3f}", flush=True)


V11 = dict(nu=1, omega=0.85)
run("V(1,1); species Chebyshev(6) / (4) when hard   [default]", V11, 6, 4)
run("Chebyshev(2,2) cycle always; species (6)/(4)", dict(poly_degree=2), 6, 4)
run("Chebyshev(2,2) cycle always; species (6)", dict(poly_degree=2), 6, None)
run("Chebyshev(2,2) cycle always; species (4)", dict(poly_degree=2), 4, None)
run("Chebyshev(3,3) cycle always; species (6)/(4)", dict(poly_degree=3, poly_fraction=10.0), 6, 4)
run("V(1,1) + Chebyshev(2,2) when hard; species (6)/(4)", dict(nu=1, omega=0.85, hard_poly_degree=2), 6, 4)
run("V(1,1) + Chebyshev(2,2) when hard; species (6)/(6)", dict(nu=1, omega=0.85, hard_poly_degree=2), 6, 6)
run("V(1,1) + Chebyshev(3,3) when hard; species (6)/(4)", dict(nu=1, omega=0.85, hard_poly_degree=3, poly_fraction=10.0), 6, 4)
