"""Runs a FEDM user script up to its first nonlinear solve against a RECORDING stand-in for the
device, and condenses what the script handed over into numbers (test infrastructure).

Two kinds of script go through it:

* our own drivers under ``examples/`` (everywhere), and
* the reference's scripts ``examples/*/fedm-*.py`` READ FROM /root/reference AT TEST TIME (build
  container only; nothing of their text is stored in this repo): the three ``from fedm... import *``
  lines and ``from dolfin import *`` are pointed at ``fedm_amd``, everything else runs as written.

What is recorded is everything `fedm.functions.Problem` + the first `PETScSNESSolver.solve` would
upload to the GPU: the bytes of ``fedm_model_desc`` / ``fedm_gd_desc``, mesh, facet tags, Dirichlet
rows, the three states, the LMEA nodal field table, the source Expression's device program and its
parameters, the step sizes.  `digest()` turns a record into a small dict of numbers (descriptor
bytes verbatim; per array: shape, sum, sum of squares, eight strided samples) -- what
tests/golden/script_records.json holds for the reference's scripts.
"""
import contextlib
import io
import os
from pathlib import Path

import numpy as np

REFERENCE = Path("/root/reference")
ROOT = Path(__file__).resolve().parent.parent
SCRIPTS = {"streamer": "examples/streamer_discharge/fedm-streamer.py",
           "glow_discharge": "examples/glow_discharge/fedm-gd.py",
           "time_of_flight": "examples/time_of_flight/fedm-tof.py"}


class FirstSolve(BaseException):
    """Raised by the recorder's newton_solve (BaseException: adaptive_solver catches Exception)."""


class RecordingDevice:
    """Stands in for fedm_amd.device.DeviceProblem; `log` is shared with the caller."""
    log = None

    def __init__(self, coords, cells, model, facet_tags=None, dirichlet_dofs=(), dirichlet_vals=(), device=0, **kw):
        self.model = model
        self.nv, self.n_eq = len(coords), model.n_eq
        self.n = self.nv * self.n_eq
        rec = type(self).log
        rec["descriptor"] = np.frombuffer(bytes(model.to_c()), dtype=np.uint8).copy()
        rec["descriptor_type"] = type(model).__name__
        rec["coords"], rec["cells"] = np.array(coords, dtype=float), np.array(cells, dtype=np.int64)
        rec["facet_tags"] = np.zeros(0) if facet_tags is None else np.array(facet_tags, dtype=np.int64)
        rec["dirichlet_dofs"] = np.array(dirichlet_dofs, dtype=np.int64)
        rec["dirichlet_vals"] = np.array(dirichlet_vals, dtype=float)

    def set_state(self, u_new=None, u_old=None, u_old1=None):
        for k, v in (("u_new", u_new), ("u_old", u_old), ("u_old1", u_old1)):
            if v is not None:
                type(self).log[k] = np.array(v, dtype=float).reshape(self.nv, -1)

    def get_state(self):
        return type(self).log["u_new"].copy()

    def get_state_old(self):
        return type(self).log.get("u_old", type(self).log["u_new"]).copy()

    def shift_state(self):
        log = type(self).log
        log["shifts"] = log.get("shifts", 0) + 1
        if "u_old" in log:
            log["u_old1"] = log["u_old"]
        log["u_old"] = log["u_new"]

    def set_step(self, dt, dt_old):
        type(self).log["step"] = np.array([dt, dt_old])

    def set_dirichlet_values(self, vals):
        type(self).log["dirichlet_vals_at_solve"] = np.array(vals, dtype=float)

    def set_gd_fields(self, fields):
        type(self).log["nodal_fields"] = np.array(fields, dtype=float)

    def set_ext_source_program(self, species, ops, consts, n_params):
        type(self).log["source_program_ops"] = np.array(ops, dtype=np.int64).ravel()
        type(self).log["source_program_consts"] = np.array(consts, dtype=float).ravel()

    def eval_ext_source(self, species, params):
        type(self).log["source_parameters"] = np.array(params, dtype=float)

    def set_ext_source(self, species, nodal):
        type(self).log["source_table"] = np.array(nodal, dtype=float)

    def setup_multigrid(self, **kw):
        pass

    def set_fieldsplit(self, *a, **kw):
        pass

    def newton_solve(self, **kw):
        type(self).log["newton_options"] = np.array([kw["rtol"], kw["max_it"]], dtype=float)
        raise FirstSolve


def _fresh_process_globals():
    """What is per process in the reference (every script is its own process): the `files` roles and
    dolfin's `parameters`."""
    from fedm_amd import file_io, forms
    for key in [k for k in file_io.files.__dict__ if k.startswith("_dir_")]:
        del file_io.files.__dict__[key]
    file_io.files.fresh_logs()
    return forms


@contextlib.contextmanager
def _recording(workdir):
    import copy
    import fedm_amd.device as fdev
    forms = _fresh_process_globals()
    saved_params, saved_cls, saved_cwd = copy.deepcopy(forms.parameters), fdev.DeviceProblem, os.getcwd()
    RecordingDevice.log = {}
    fdev.DeviceProblem = RecordingDevice
    os.chdir(workdir)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            yield RecordingDevice.log
    finally:
        os.chdir(saved_cwd)
        fdev.DeviceProblem = saved_cls
        forms.parameters.clear()
        forms.parameters.update(saved_params)
        _fresh_process_globals()


def prepare_workdir(case, workdir):
    """The files a script expects next to it: its deck as ./file_input, the streamer's mesh.xml."""
    workdir = Path(workdir)
    workdir.mkdir(parents=True, exist_ok=True)
    deck = {"streamer": "streamer_discharge", "glow_discharge": "glow_discharge"}.get(case)
    if deck and not (workdir / "file_input").exists():
        os.symlink(ROOT / "decks" / deck / "file_input", workdir / "file_input")
    if case == "streamer" and not (workdir / "mesh.xml").exists():
        from fedm_amd import mesh_io
        from fedm_amd.cases import streamer
        mesh_io.write_dolfin_xml(streamer.refined_mesh(1.5e-4), workdir / "mesh.xml")
    return workdir


def _exec_reference_script(case, workdir):
    """(child process) the reference's script for `case` from /root/reference, imports redirected, until its
    first nonlinear solve; returns the record."""
    path = REFERENCE / SCRIPTS[case]
    text = path.read_text()
    text = text.replace("from dolfin import *", "from fedm_amd.forms import *")
    for module in ("physical_constants", "file_io", "functions"):
        text = text.replace(f"from fedm.{module} import *", f"from fedm_amd.{module} import *")
    workdir = prepare_workdir(case, workdir)
    with _recording(workdir) as log:
        try:
            exec(compile(text, str(path), "exec"), {"__name__": "__main__", "__file__": str(workdir / path.name)})
        except FirstSolve:
            pass
        else:
            raise AssertionError(f"{path} finished without calling the nonlinear solver")
    return dict(log)


def run_reference_script(case, workdir):
    """Run the reference's script for `case` up to its first nonlinear solve and return the record.  The script is
    third-party code: it runs in a CHILD process of its own (this module as a program) with a scrubbed environment,
    its working directory inside the test's temporary directory, and hands back nothing but arrays and strings
    (an .npz without pickles) -- not in the pytest process."""
    import subprocess
    import sys
    workdir = Path(workdir)
    workdir.mkdir(parents=True, exist_ok=True)
    out = workdir / "record.npz"
    env = {k: v for k, v in os.environ.items() if k in ("PATH", "HOME", "LANG", "LC_ALL", "TMPDIR", "LD_LIBRARY_PATH")}
    env["PYTHONPATH"] = os.pathsep.join([str(ROOT), str(Path(__file__).resolve().parent)])
    r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--child", case, str(workdir), str(out)],
                       env=env, cwd=str(workdir), capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise AssertionError(f"the reference's {SCRIPTS[case]} failed on the facade:\n{r.stderr[-3000:]}")
    with np.load(out, allow_pickle=False) as z:
        rec = {}
        for key in z.files:
            rec[key[2:]] = str(z[key]) if key.startswith("s_") else z[key]
    return rec


def run_own_example(case, workdir):
    """The same for examples/<case>.py of this repo, on the same inputs."""
    import importlib.util
    name = {"streamer": "streamer_discharge"}.get(case, case)
    spec = importlib.util.spec_from_file_location(f"own_{name}", ROOT / "examples" / f"{name}.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    workdir = prepare_workdir(case, workdir)
    deck = str(workdir / "file_input")
    calls = {"streamer": lambda: mod.main(mesh_file=workdir / "mesh.xml", input_dir=deck, output_dir=workdir / "output",
                                          quiet=True),
             "glow_discharge": lambda: mod.main(input_dir=deck, output_dir=workdir / "output"),
             "time_of_flight": lambda: mod.main(output_dir=workdir / "output", quiet=True)}
    with _recording(workdir) as log:
        try:
            calls[case]()
        except FirstSolve:
            pass
        else:
            raise AssertionError(f"examples/{name}.py finished without calling the nonlinear solver")
    return dict(log)


def digest(record):
    """A record as plain numbers (JSON-able)."""
    out = {}
    for key, value in sorted(record.items()):
        if isinstance(value, str):
            out[key] = value
        elif key == "descriptor":
            out[key] = value.tolist()
        else:
            a = np.asarray(value, dtype=float)
            flat = a.ravel()
            pick = flat[:: max(1, flat.size // 8)][:8] if flat.size else flat
            out[key] = dict(shape=list(a.shape), sum=float(flat.sum()), sum_sq=float((flat * flat).sum()),
                            samples=[float(x) for x in pick])
    return out


def assert_same_digest(got, want, rtol=1e-12):
    assert sorted(got) == sorted(want), (sorted(got), sorted(want))
    for key, w in want.items():
        g = got[key]
        if isinstance(w, str):
            assert g == w, key
        elif key == "descriptor":
            assert g == w, "descriptor bytes differ"
        else:
            assert g["shape"] == w["shape"], key
            scale = max(np.sqrt(w["sum_sq"]), 1e-300)
            assert abs(g["sum"] - w["sum"]) <= rtol * max(abs(w["sum"]), scale * np.sqrt(max(np.prod(w["shape"]), 1))), key
            assert abs(g["sum_sq"] - w["sum_sq"]) <= rtol * max(w["sum_sq"], 1e-300), key
            np.testing.assert_allclose(g["samples"], w["samples"], rtol=rtol, atol=rtol * scale, err_msg=key)


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 5 and sys.argv[1] == "--child":
        record = _exec_reference_script(sys.argv[2], sys.argv[3])
        np.savez(sys.argv[4], **{("s_" if isinstance(v, str) else "a_") + k: (np.array(v) if isinstance(v, str) else np.asarray(v))
                                  for k, v in record.items()})
