import sys, numpy as np
sys.path.insert(0, '.')
from fedm_amd.cases import glow_discharge as gdc
host = gdc.Case(nx=24, ny=24, device_pipeline=False, T_final=1.0)
dev = gdc.Case(nx=24, ny=24, device_pipeline=True, T_final=1.0)
for it in range(3):
    host.step(); dev.step()
    fh, fd = host.prob.get_gd_fields(), dev.prob.get_gd_fields()
    scale = np.maximum(np.abs(fh).max(axis=1), 1e-300)
    err = np.abs(fh - fd).max(axis=1) / scale
    print(it, [(r, float('%.2e' % e)) for r, e in enumerate(err) if e > 1e-13])
    print('state', np.abs(host.prob.get_state() - dev.prob.get_state()).max())
