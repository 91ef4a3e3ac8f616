"""Damping of the V-cycle's Jacobi smoother against Krylov steps, early and late in the run."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fedm_amd.cases import streamer
msh = streamer.mesh(576, 4.0)
for omega in (0.5, 0.6, 0.67, 0.75, 0.85, 0.95):
    prob = streamer.device_problem(msh.coords, msh.cells)
    st = streamer.Stepper(prob)
    st.initialise()
    prob.setup_multigrid(nu=1, omega=omega)
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
        out.append(f"steps {start}-{start + 10}: gmres/step {(st.linear_iterations - n0) / 10} ms/step {(time.time() - t0) * 100:.3f}")
    print(f"omega {omega}", " | ".join(out), flush=True)
    prob.close()
