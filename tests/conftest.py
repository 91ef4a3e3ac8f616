import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _script_globals_are_per_test():
    """`parameters` (dolfin's global of the same name) and the output/input folders of `files` are
    process-wide in the reference too, where every script is a process of its own: restore them after
    each test, so that e.g. the time-of-flight script finds the quadrature degree unset as it expects."""
    import copy
    from fedm_amd import forms
    saved = copy.deepcopy(forms.parameters)
    yield
    forms.parameters.clear()
    forms.parameters.update(saved)
