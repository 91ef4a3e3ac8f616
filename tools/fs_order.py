"""Order of the block-triangular field split (lower: species first; upper: potential first) and the
degree of the species polynomial, early (steps 6..25) and late (from step 201) in the bench run.
python tools/fs_order.py [n=576] [late_start=200]     (tests/studies/precond_structure.py is the
CPU study behind it; tools/fs_order_accuracy.py shows what the faster order costs in accuracy).
Round-2 results: DESIGN.md section 4."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
from fedm_amd.device import chebyshev_weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 576
late = int(sys.argv[2]) if len(sys.argv) > 2 else 200
msh = streamer.mesh(n, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
st = streamer.Stepper(prob)
st.initialise()

CONFIGS = [("lower, Chebyshev(6) / (4) when hard  [default]", "lower", 6, 4),
           ("upper, Chebyshev(6)", "upper", 6, None),
           ("upper, Chebyshev(6) / (4) when hard", "upper", 6, 4),
           ("upper, Chebyshev(4)", "upper", 4, None),
           ("upper, Chebyshev(8)", "upper", 8, None),
           ("lower, Chebyshev(6)", "lower", 6, None),
           ("upper, Chebyshev(6)  (again)", "upper", 6, None)]


def measure(name, order, deg, hard, steps, switch_above=5.0, back_below=3.5):
    prob.set_fieldsplit_order(order)
    prob.set_fieldsplit(chebyshev_weights(deg), hard_weights=None if hard is None else chebyshev_weights(hard),
                        switch_above=switch_above, back_below=back_below)
    st.step()   # graphs of this variant
    st.step()
    n0, l0 = st.newton_iterations, st.linear_iterations
    t0 = time.perf_counter()
    for _ in range(steps):
        st.step()
    dt = time.perf_counter() - t0
    print(f"  {name:58s} newton/step {(st.newton_iterations - n0) / steps:5.2f}  gmres/step "
          f"{(st.linear_iterations - l0) / steps:6.2f}  ms/step {1e3 * dt / steps:7.3f}", flush=True)


for _ in range(5):
    st.step()
print(f"early window (t = {st.t:.2e} s)", flush=True)
for cfg in CONFIGS:
    measure(*cfg, steps=10)
prob.set_fieldsplit_order("lower")
prob.set_fieldsplit(chebyshev_weights(6), hard_weights=chebyshev_weights(4))
while st.t < late * 5e-12:
    st.step()
print(f"late window (t = {st.t:.2e} s); every variant starts from the same stored state", flush=True)
snap = (prob.get_state(), prob.get_state_old(), st.t, st.dt.time_step, st.dt_old.time_step, list(st.max_error),
        list(st.error))


def restore():
    U, Uold, st.t, st.dt.time_step, st.dt_old.time_step, me, er = snap
    prob.set_state(U, Uold, Uold)
    st.max_error[:], st.error[:] = me, er


LATE = CONFIGS[:3] + [("upper, Chebyshev(8)", "upper", 8, None), ("upper, Chebyshev(10)", "upper", 10, None),
                      ("upper, Chebyshev(12)", "upper", 12, None),
                      ("upper, Chebyshev(6) / (8) when hard", "upper", 6, 8),
                      ("upper, Chebyshev(6) / (10) when hard", "upper", 6, 10)]
for cfg in LATE:
    restore()
    kw = dict(switch_above=3.4, back_below=2.2) if cfg[3] is not None and cfg[3] > cfg[2] else {}
    measure(*cfg, steps=20, **kw)
for lo, hi in ((0.4, 2.0), (0.5, 2.2), (0.35, 2.2)):
    restore()
    prob.set_fieldsplit_order("upper")
    prob.set_fieldsplit(chebyshev_weights(8, lo, hi))
    st.step()
    n0, l0 = st.newton_iterations, st.linear_iterations
    t0 = time.perf_counter()
    for _ in range(20):
        st.step()
    print(f"  upper, Chebyshev(8) on [{lo}, {hi}]: newton/step {(st.newton_iterations - n0) / 20:5.2f} gmres/step "
          f"{(st.linear_iterations - l0) / 20:6.2f} ms/step {50 * (time.perf_counter() - t0):7.3f}", flush=True)
