"""Streamer run on the locally refined unstructured mesh up to the reference's end time
(fedm-streamer.py:67, T_final = 1.4e-8 s): per-interval statistics for profiles/.

usage: python tools/refined_run.py H_FINE [T_FINAL [GROWTH [R_CHANNEL]]]
Writes gpurun_out/refined_run_<h in nm>.json; stops early on a blow-up (recorded as such)."""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from fedm_amd.cases import streamer

h_fine = float(sys.argv[1])
T_final = float(sys.argv[2]) if len(sys.argv) > 2 else 1.4e-8
growth = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
r_ch = float(sys.argv[4]) if len(sys.argv) > 4 else streamer.CHANNEL[1]
every = 100

t0 = time.time()
channel = (0.0, r_ch) + streamer.CHANNEL[2:]
import tempfile
with tempfile.TemporaryDirectory(prefix="fedm_mesh_") as tmp:      # (tens of MB of XML: not into gpurun_out/)
    msh = streamer.refined_mesh(h_fine, growth=growth, channel=channel, xml_path=tmp + "/mesh.xml")
t_mesh = time.time() - t0
prob = streamer.device_problem(msh.coords, msh.cells)
st = streamer.Stepper(prob)
st.initialise()
t_setup = time.time() - t0
sz = prob.sizes()
print(dict(vertices=msh.num_vertices(), cells=msh.num_cells(), dofs=prob.n, hmin=msh.hmin(), hmax=msh.hmax(),
           mesh_s=round(t_mesh, 1), setup_s=round(t_setup, 1), **sz), flush=True)

x = msh.coords
c = msh.cells
a, b, d = x[c[:, 0]], x[c[:, 1]], x[c[:, 2]]
det = (b[:, 0] - a[:, 0]) * (d[:, 1] - a[:, 1]) - (b[:, 1] - a[:, 1]) * (d[:, 0] - a[:, 0])
G = np.stack([np.stack([b[:, 1] - d[:, 1], d[:, 0] - b[:, 0]], 1), np.stack([d[:, 1] - a[:, 1], a[:, 0] - d[:, 0]], 1),
              np.stack([a[:, 1] - b[:, 1], b[:, 0] - a[:, 0]], 1)], 1) / det[:, None, None]
zc = x[c].mean(axis=1)[:, 1]
near_axis = x[c].mean(axis=1)[:, 0] < 2.0 * h_fine


def field_stats(U):
    E = -np.einsum("ca,cad->cd", U[c, 2], G)
    Em = np.linalg.norm(E, axis=1)
    k = np.argmax(np.where(near_axis, Em, 0.0))
    return float(Em[k]), float(zc[k])


hist, t_run = [], time.time()
status = "reached T_final"
n0 = l0 = a0 = s0 = 0
while st.t < T_final * (1 - 1e-9):
    try:
        st.step()
    except Exception as exc:                       # noqa: BLE001 - recorded, run ends
        status = f"stopped: {type(exc).__name__}: {exc}"
        break
    if st.steps % every == 0 or st.t >= T_final * (1 - 1e-9):
        U = prob.get_state()
        rows = st.log_rows()
        Emax, zhead = field_stats(U)
        span = max(st.steps - s0, 1)
        hist.append(dict(step=st.steps, t=st.t, dt=st.dt.time_step, newton_per_step=(st.newton_iterations - n0) / span,
                         gmres_per_step=(st.linear_iterations - l0) / span, attempts=len(rows) - a0,
                         ne_max=float(np.exp(U[:, 1].max())), ni_max=float(np.exp(U[:, 0].max())),
                         E_max_axis=Emax, z_head=zhead, wall=round(time.time() - t_run, 2)))
        n0, l0, a0, s0 = st.newton_iterations, st.linear_iterations, len(rows), st.steps
        print(hist[-1], flush=True)
        if not np.isfinite(hist[-1]["ne_max"]) or hist[-1]["ne_max"] > 1e24 or st.dt.time_step < 2e-15:
            status = "stopped: blow-up (ne_max > 1e24 or dt collapsed)"
            break
rows = st.log_rows()
out = dict(h_fine=h_fine, growth=growth, channel=channel, vertices=msh.num_vertices(), cells=msh.num_cells(), dofs=prob.n,
           sizes=sz, status=status, t_end=st.t, accepted_steps=st.steps, attempts=len(rows),
           rejected=len(rows) - st.steps, wall_s=round(time.time() - t_run, 2),
           steps_per_s=st.steps / max(time.time() - t_run, 1e-9), history=hist)
json.dump(out, open("gpurun_out/refined_run_%d.json" % round(h_fine * 1e9), "w"), indent=1)
print({k: v for k, v in out.items() if k != "history"}, flush=True)
