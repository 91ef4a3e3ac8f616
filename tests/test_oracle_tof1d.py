"""BASELINE configs[0]: 1-D time of flight on the CPU path (oracle only, no GPU).

The reference ships no golden for this case (parity unpinned, oracle/tof1d.py); the check is
the one the example performs itself -- the relative L2 error of the density against the
analytic pulse (examples/time_of_flight_1D/fedm-tof_1d.py:155-160) -- plus its order in dt."""
import numpy as np

from oracle import tof1d


def test_tof1d_tracks_the_analytic_pulse():
    errs, u, sp2 = tof1d.run(n_cells=1000, n_steps=40, output_every=10)
    assert sp2.ndof == 2001 and len(errs) == 4
    assert all(np.isfinite(e) and e < 5e-3 for _, e in errs)       # 4.1e-3 ... 2.9e-3 (BDF1 start-up)
    assert errs[-1][1] < errs[0][1]
    # the pulse maximum moved with the drift velocity
    xmax = sp2.x[np.argmax(u)]
    assert abs(xmax - (tof1d.X0 + tof1d.W_DRIFT * 40e-11)) < 2 * sp2.h


def test_tof1d_is_second_order_in_time():
    e1 = tof1d.run(n_cells=1000, dt=1e-11, n_steps=40, output_every=40)[0][-1][1]
    e2 = tof1d.run(n_cells=1000, dt=5e-12, n_steps=80, output_every=80)[0][-1][1]
    assert 3.6 < e1 / e2 < 4.4        # BDF2: halving dt quarters the error (2.9e-3 -> 7.3e-4)


def test_tof1d_jacobian_is_the_derivative_of_the_residual():
    sp2 = tof1d.P2Interval(20, 1e-3)
    rng = np.random.default_rng(0)
    u = tof1d.log_density(sp2.x, 1e-10) + rng.normal(0, 0.1, sp2.ndof)
    uo = tof1d.log_density(sp2.x, 0.9e-10)
    uo1 = tof1d.log_density(sp2.x, 0.8e-10)
    f = tof1d.source(sp2.x, 1e-10)
    R, J = tof1d.residual_jacobian(sp2, u, uo, uo1, f, 1e-11, 1.2e-11)
    v = rng.normal(size=sp2.ndof)
    h = 1e-6
    Rp, _ = tof1d.residual_jacobian(sp2, u + h * v, uo, uo1, f, 1e-11, 1.2e-11, jac=False)
    Rm, _ = tof1d.residual_jacobian(sp2, u - h * v, uo, uo1, f, 1e-11, 1.2e-11, jac=False)
    fd = (Rp - Rm) / (2 * h)
    assert np.abs(fd - J @ v).max() / np.abs(fd).max() < 1e-7
