"""Pin the oracle on the reference's time-of-flight golden results.

Mirrors tests/integrated_tests/time_of_flight/test_time_of_flight.py:45-56 of
the reference: same harness constants, same two assertions, same tolerances.
"""
import numpy as np
import pytest

from oracle import tof


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(golden_dir / "tof_golden.npz")


@pytest.fixture(scope="module")
def result():
    return tof.run()


def test_mesh_matches_golden(golden, result):
    mesh = result["mesh"]
    assert np.array_equal(mesh.cells, golden["cells"])
    assert np.allclose(mesh.coords, golden["coords"], rtol=0, atol=1e-18)
    assert result["h_max"] == pytest.approx(float(golden["h_max"]), rel=1e-14)


def test_time_of_flight_relative_error(golden, result):
    assert result["steps"] == 100
    assert np.isclose(result["relative_error"], float(golden["relative_error"]))


def test_time_of_flight_electron_number_density(golden, result):
    ref = golden["n_e"]
    error = (result["n_num"] - ref) / ref
    assert np.mean(np.abs(error)) < 1e-5
    assert np.sqrt(np.mean(error ** 2)) < 1e-5
    assert np.max(np.abs(error)) < 1e-3
