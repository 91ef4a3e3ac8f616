"""V-cycle variants on the bench case: iterations and wall time per step, early (steps 1-11) and
later in the run (steps 250-260)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from fedm_amd.cases import streamer
msh = streamer.mesh(576, 4.0)
variants = [("V(1,1)", dict(nu=1)), ("V(0,1)", dict(nu=-1)), ("V(0,2)", dict(nu=-2)), ("V(2,2)", dict(nu=2)),
            ("V(3,3)", dict(nu=3)), ("V(4,4)", dict(nu=4))]
if len(sys.argv) > 1:
    variants = [v for v in variants if v[0] in sys.argv[1:]]
for name, mg in variants:
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob)
    st.initialise()      # initial Poisson solve: CG needs the symmetric V(1,1)
    prob.setup_multigrid(**mg)
    try:
        out = []
        for start in (1, 250):
            while st.steps < start:
                st.step()
            prob.get_state()
            n0 = st.linear_iterations
            t0 = time.time()
            for _ in range(10):
                st.step()
            prob.get_state()
            out.append(f"steps {start}-{start + 10}: gmres/step {(st.linear_iterations - n0) / 10} "
                       f"ms/step {(time.time() - t0) * 100:.3f}")
        print(name, " | ".join(out), flush=True)
    except Exception as e:
        print(name, "FAILED", e, flush=True)
    prob.close()
