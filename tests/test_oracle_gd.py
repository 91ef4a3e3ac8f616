"""Pin the oracle on the reference's glow-discharge goldens.

Mirrors tests/integrated_tests/glow_discharge/test_glow_discharge.py:48-62 of the
reference: the 6-row `relative error.log` with np.allclose, and every checkpoint snapshot
of the three written species (log number density) with L1, L2 < 1e-5 and Linf < 1e-3.
The oracle actually agrees to ~1e-13, so tighter bounds are asserted as well.
"""
from pathlib import Path

import numpy as np
import pytest

from oracle import gd

ROOT = Path(__file__).resolve().parent.parent
DECK = ROOT / "decks" / "glow_discharge" / "file_input" / "4_particles"


@pytest.fixture(scope="module")
def result():
    return gd.run(DECK)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(golden_dir / "gd_golden.npz")


def test_mesh_is_dolfins_crossed_mesh(result, golden):
    m = result["model"].mesh
    assert np.array_equal(m.cells, golden["cells"])
    assert np.allclose(m.coords, golden["coords"], rtol=0, atol=1e-17)


def test_glow_discharge_relative_error(result, golden_dir):
    import json
    ref = np.array(json.loads((golden_dir / "error_logs.json").read_text())["glow_discharge"])
    log = np.array(result["log"])
    assert log.shape == ref.shape == (6, 3)
    assert np.allclose(log, ref)                    # the reference's assertion
    assert np.allclose(log, ref, rtol=1e-8, atol=0)


@pytest.mark.parametrize("key,comp", [("electrons", 3), ("Ar_plus", 2), ("Ar_star", 1)])
def test_glow_discharge_number_density(result, golden, key, comp):
    # snapshot 0 is the initial condition, snapshot 1 the state interpolated to t = 1e-11 s
    assert np.allclose(golden[key + "_0"], np.log(1e12), rtol=1e-15)
    ref = golden[key + "_1"]
    error = (result["snapshot"][:, comp] - ref) / ref
    assert np.mean(np.abs(error)) < 1e-5 and np.sqrt(np.mean(error ** 2)) < 1e-5
    assert np.max(np.abs(error)) < 1e-3
    assert np.max(np.abs(error)) < 1e-10
