import sys, time, cProfile, pstats, io
sys.path.insert(0, '.')
from fedm_amd.cases import streamer
msh = streamer.mesh(576, 4.0)
prob = streamer.device_problem(msh.coords, msh.cells)
st = streamer.Stepper(prob)
st.initialise()
for _ in range(5):
    st.step()
prob.get_state()
t0 = time.perf_counter()
for _ in range(50):
    st.step()
prob.get_state()
print("ms/step", (time.perf_counter() - t0) * 20)
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    st.step()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])
