"""FEDM's input-deck readers (``fedm.file_io``) as drop-ins.

Same function names, argument meaning, return shapes and error types as
fedm/file_io.py:123-535, including the format quirks the decks rely on:

* species/reaction matching by substring count (file_io.py:288-289),
* ``"const."`` (with the dot) accepted for transport coefficients (:412) while the
  interpolation step only knows ``"const"`` (functions.py:583),
* a missing mobility file means dependence ``0`` (:346-359, :447-450),
* the regular expressions of :309, :323, :486-487 (they are part of the file format).

``file_output`` and the PVD / XDMF writers live in :mod:`fedm_amd.mesh_io`.
"""
import re
from pathlib import Path

import numpy as np

from .utils import comma_separated, print_rank_0


# ---------------------------------------------------------------------------
# file_io.py:22-117 -- path manager
# ---------------------------------------------------------------------------
def truncate_file(path):
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    open(path, "w").close()


class _Directory:
    """A directory role of the run.  ``must_exist``: assigning a path that is not a directory is
    an error (the deck); otherwise the directory is created on assignment (the output)."""

    def __init__(self, default, must_exist):
        self.default, self.must_exist = default, must_exist

    def __set_name__(self, owner, name):
        self.name, self.slot = name, "_dir_" + name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        return obj.__dict__.get(self.slot) or Path.cwd() / self.default

    def __set__(self, obj, value):
        value = Path(value)
        if self.must_exist:
            if not value.is_dir():
                raise RuntimeError(f"fedm.files.{self.name}: '{value}' is not a directory")
        else:
            if value.resolve() != self.__get__(obj).resolve():
                obj.fresh_logs()             # another output directory: its logs start empty
            value.mkdir(exist_ok=True)
        obj.__dict__[self.slot] = value


class _LogFile:
    """A log in the output directory, emptied the first time it is asked for there."""

    def __init__(self, file_name):
        self.file_name = file_name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        target = obj.output_folder_path / self.file_name
        if self.file_name not in obj._started:
            truncate_file(target)
            obj._started.add(self.file_name)
        return target


class Files:
    """The path roles of a FEDM run (``fedm.file_io.files``): where the decks are read from, where
    results go, and the two logs kept there."""

    file_input = _Directory("file_input", must_exist=True)
    output_folder_path = _Directory("output", must_exist=False)
    error_file = _LogFile("relative error.log")
    model_log = _LogFile("model.log")

    def __init__(self):
        self._started = set()

    def fresh_logs(self):
        self._started.clear()


files = Files()


def output_files(file_type, type_of_output, output_file_names):
    """fedm/file_io.py:148-188: one writer per name under ``<output>/<type_of_output>/<name>/``
    ('pvd': VTU series, 'xdmf': XDMF + HDF5 checkpoints in DOLFIN's layout); the mesh is taken from
    the first Function written."""
    from . import mesh_io
    kinds = {"pvd": mesh_io.PVDFile, "xdmf": mesh_io.XDMFFile}
    if file_type not in kinds:
        raise ValueError(f"fedm.output_files: file type '{file_type}' is not valid. Options are 'pvd' or 'xdmf'.")
    out = files.output_folder_path / type_of_output
    return [kinds[file_type](out / name / f"{name}.{file_type}") for name in output_file_names]


# ---------------------------------------------------------------------------
# Deck files.  Every deck file is text with '#' comments; what differs is how the payload is
# shaped (one number, one expression string, two columns, key/value lines).  One small class
# reads them all; the public functions of fedm.file_io are thin views of it.
# ---------------------------------------------------------------------------
class _Deck:
    """Comment-stripped view of one deck file."""

    def __init__(self, path, encoding=None):
        self.path = Path(path)
        with open(self.path, "r", encoding=encoding) as handle:
            self.lines = list(decomment(handle))

    def first(self, cast=str):
        return cast(self.lines[0])

    def columns(self):
        table = np.array([ln.split()[:2] for ln in self.lines], dtype=float)
        return table[:, 0].tolist(), table[:, 1].tolist()

    def findall(self, pattern):
        return [hit for ln in self.lines for hit in pattern.findall(ln)]


def decomment(lines):
    """The payload of each line: what precedes '#', stripped; empty lines are dropped."""
    stripped = (ln.partition("#")[0].strip() for ln in lines)
    return (ln for ln in stripped if ln)


def no_convert(x):
    return x


def read_single_value(file_name, convert=no_convert):
    return _Deck(file_name, "utf8").first(convert)


def read_single_float(file_name, convert=no_convert):
    return _Deck(file_name, "utf8").first(float)


def read_single_string(file_name):
    return _Deck(file_name, "utf8").first(str)


def read_and_decomment(file_name):
    return _Deck(file_name).lines


def read_two_columns(file_name):
    return _Deck(file_name).columns()


def flatten(input_list):
    return [item for sub in input_list for item in sub]


def flatten_float(input_list):
    return list(map(float, flatten(input_list)))


def _problem(exc, where, text):
    """Exceptions carry the reference's one-line wording: ``fedm.<function>: <text>``."""
    return exc(f"fedm.{where}: {text}")


# ---------------------------------------------------------------------------
# speclist.cfg / reacscheme.cfg (file_io.py:250-343)
# ---------------------------------------------------------------------------
_KFILE = re.compile(r"kfile: ([A-Za-z0-9_]+.[A-Za-z0-9_]+)")                    # part of the format, :309
_UIN = re.compile(r"Uin:\s?([+-]?\d+.\d+[eE]?[-+]?\d+|0|1.0)")                  # :323
_MASS = re.compile(r"Mass\s?=\s?([+-]?\d+.\d+[eE]?[-+]?\d+|0|1.0)")             # :486
_CHARGE = re.compile(r"Z\s+?=\s+?([+-]?\d+)")                                   # :487


def read_speclist(file_path):
    """(n, species names, property file names, names used for transport files)."""
    entries = [ln.replace("file:", "").split() for ln in _Deck(Path(file_path) / "speclist.cfg").lines
               if "file:" in ln]
    names, prop_files = [e[0] for e in entries], [e[1] for e in entries]
    return len(entries), names, prop_files, [f.split(".")[0] for f in prop_files]


def reaction_matrices(path, species):
    """Power, loss and gain matrices (n_reactions x n_species, int) from reacscheme.cfg.
    A species counts by substring occurrences on each side of ' -> ' (the format's rule)."""
    sides = [ln.partition(" Type:")[0].partition(" -> ")[::2] for ln in _Deck(Path(path) / "reacscheme.cfg").lines]
    count = lambda k: np.array([[side[k].rstrip().count(sp) for sp in species] for side in sides],
                               dtype=int).reshape(len(sides), len(species))
    consumed, produced = count(0), count(1)
    balance = consumed - produced
    return consumed, np.maximum(balance, 0), np.maximum(-balance, 0)


def rate_coefficient_file_names(path):
    folder = Path(path)
    return [folder / "rate_coefficients" / name for name in _Deck(folder / "reacscheme.cfg").findall(_KFILE)]


def read_energy_loss(path):
    losses = [float(v) for v in _Deck(Path(path) / "reacscheme.cfg").findall(_UIN)]
    print_rank_0(losses)
    return losses


def read_dependence(file_name):
    target = Path(file_name)
    if not target.is_file():
        raise _problem(FileNotFoundError, "read_dependence", f"file '{target}' not found")
    with open(target, "r", encoding="utf8") as handle:     # the keyword sits in a comment line
        tagged = [ln for ln in handle if "Dependence:" in ln]
    if not tagged:
        raise _problem(RuntimeError, "read_dependence", f"Did not find dependence in file '{target}'")
    return tagged[0].split()[2]


def read_dependences(file_names, zero_if_file_missing=False):
    def one(name):
        try:
            return read_dependence(name)
        except FileNotFoundError:
            if zero_if_file_missing:
                return 0
            raise
    return [one(name) for name in file_names]


# ---------------------------------------------------------------------------
# coefficient files (file_io.py:362-475): how a file is read follows from its dependence tag
# ---------------------------------------------------------------------------
_SCALAR, _TEXT, _TABLE, _NOTHING = "scalar", "text", "table", "nothing"
_RATE_KINDS = {"const": _SCALAR, "fun:Te,Tgas": _TEXT, "fun:Tgas": _TEXT,
               "Umean": _TABLE, "E/N": _TABLE, "ElecDist": _TABLE}
_TRANSPORT_KINDS = {"const": _SCALAR, "const.": _SCALAR, "fun:Te,Tgas": _TEXT, "fun:E": _TEXT,
                    "Umean": _TABLE, "E/N": _TABLE, "Tgas": _TABLE, "Te": _TABLE}


def _payload(file_name, kind):
    """(x, y) as the reference returns them: tables as two lists, everything else as (0.0, value)."""
    if kind == _TABLE:
        return _Deck(file_name).columns()
    if kind == _SCALAR:
        return 0.0, _Deck(file_name, "utf8").first(float)
    if kind == _TEXT:
        return 0.0, _Deck(file_name, "utf8").first(str)
    return 0.0, 0.0


def read_rate_coefficients(rc_file_names, k_dependences):
    if len(rc_file_names) != len(k_dependences):
        raise _problem(ValueError, "read_rate_coefficients",
                       "rc_file_names and k_dependences should be the same length.")
    unknown = [d for d in k_dependences if d not in _RATE_KINDS]
    if unknown:
        raise _problem(ValueError, "read_rate_coefficients",
                       f"The dependence '{unknown[0]}' is not recognised. Options are "
                       f"{comma_separated(list(_RATE_KINDS))}.")
    kxs, kys = [], []
    for name, tag in zip(rc_file_names, k_dependences):
        print_rank_0(name)
        x, y = _payload(name, _RATE_KINDS[tag])
        kxs.append(x)
        kys.append(y)
    return kxs, kys


def read_transport_coefficients(particle_names, transport_type, model):
    """(kxs, kys, dependences).  ``'fun:E'`` entries keep their expression string (the
    reference ``eval``s it later, fedm-streamer.py:237-238; here ``termsum.parse`` does)."""
    folder = files.file_input / model / "transport_coefficients"
    if not folder.is_dir():
        raise _problem(FileNotFoundError, "read_transport_coefficients",
                       f"Transport coeff dir '{folder}' not found.")
    diffusion, mobility = transport_type == "Diffusion", transport_type == "mobility"
    kinds = dict(_TRANSPORT_KINDS)
    if diffusion:
        kinds["ESR"] = _NOTHING            # Einstein relation: computed from the mobility later
    if mobility:
        kinds[0] = _NOTHING                # no mobility file: the species does not drift
    file_names = [folder / f"{particle}{'_ND.dat' if diffusion else '_Nb.dat'}" for particle in particle_names]
    tags = read_dependences(file_names, zero_if_file_missing=mobility)
    for tag in tags:
        if tag not in kinds:
            raise _problem(ValueError, "read_transport_coefficients",
                           f"Dependence '{tag}' not recognised. For the transport type "
                           f"'{transport_type}', the possible options are {comma_separated(list(kinds))}.")
    kxs, kys = [], []
    for name, tag in zip(file_names, tags):
        if tag == 0:
            kxs.append(0)
            kys.append(0)
            continue
        print_rank_0(name)
        x, y = _payload(name, kinds[tag])
        if tag == "fun:Te,Tgas":   # the reference eval()s arbitrary Python here; refused on purpose
            raise _problem(RuntimeError, "read_transport_coefficients", f"ky eval failed, '{y}'")
        kxs.append(x)
        kys.append(y)
    return kxs, kys, tags


# ---------------------------------------------------------------------------
# species property files (file_io.py:478-535)
# ---------------------------------------------------------------------------
def read_particle_properties(file_names, model):
    folder = files.file_input / model / "species"
    masses, charges = [], []
    for name in file_names:
        target = folder / name
        if not target.is_file():
            raise _problem(RuntimeError, "read_particle_properties", f"File '{target}' not found.")
        print_rank_0(target)
        deck = _Deck(target)
        for ln in deck.lines:
            print_rank_0(ln)
        found = {"mass": deck.findall(_MASS), "charge": deck.findall(_CHARGE)}
        for what, hits in found.items():
            if not hits:
                raise _problem(RuntimeError, "read_particle_properties", f"No {what} found in file '{target}'.")
        masses += [float(v) for v in found["mass"]]
        charges += [float(v) for v in found["charge"]]
    return masses, charges


def print_time_step(dt):
    print_rank_0("Time step is dt =", dt)


def print_time(t):
    print_rank_0("t =", t)


def numpy_2d_array_to_str(x):
    rows = str(np.asarray(x)).replace("[", "").replace("]", "").split("\n")
    return "\n".join(row.strip() for row in rows)


# ---------------------------------------------------------------------------
# model log (file_io.py:641-724): one formatter per entry kind
# ---------------------------------------------------------------------------
def _log_properties(gas, model, names, M, charge):
    return "\n\n".join([f"Gas:\t{gas}", f"model:\t{model}", f"Particle names:\n{names}",
                        f"Mass:\n{M}", f"Charge:\n{charge}"]) + "\n"


def _log_conditions(dt_var, U_w, p0, gap_length, N0, Tgas):
    items = [("dt", dt_var, "s"), ("U_w", U_w, "V"), ("p_0", p0, "Torr"), ("d", gap_length, "m"),
             ("N_0", N0, "m^-3"), ("T_gas", Tgas, "K")]
    line = "\t ".join(f"{k} = {v} {u}" + ("," if k != "T_gas" else "") for k, v, u in items)
    return f"Simulation conditions:\n{line}\n"


def _log_matrices(gain, loss, power):
    blocks = [("Gain", gain), ("Loss", loss), ("Power", power)]
    return "\n\n".join(f"{title} matrix:\n{numpy_2d_array_to_str(m)}" for title, m in blocks) + "\n"


def file_output(*args, **kwargs):
    """fedm/file_io.py:527-616, under the name and in the module the scripts import it from
    (``from fedm.file_io import *``); the writers live in :mod:`fedm_amd.mesh_io`."""
    from .mesh_io import file_output as write
    return write(*args, **kwargs)


def mesh_statistics(mesh):
    """fedm/file_io.py:619-631: the mesh as ``<output>/mesh/mesh.pvd`` and its element count and
    extreme edge lengths on the terminal and in ``mesh info.txt``."""
    from . import mesh_io
    from .utils import _rank, mesh_info
    mesh_dir = files.output_folder_path / "mesh"
    info = mesh_info(mesh)
    if _rank() == 0:
        mesh_io.PVDFile(mesh_dir / "mesh.pvd", mesh).write(np.zeros(mesh.num_vertices()), "mesh", 0.0)
        print(info.rstrip())
        (mesh_dir / "mesh info.txt").write_text(info)


_LOG_ENTRIES = {
    "properties": _log_properties,
    "conditions": _log_conditions,
    "matrices": _log_matrices,
    "initial time": lambda t: f"Time:\n{t}",
    "time": lambda t: str(t),
    "mesh": lambda mesh: __import__("fedm_amd.utils", fromlist=["mesh_info"]).mesh_info(mesh),
}


def log(log_type, log_file_name, *args):
    """Append one entry to the model log in the reference's layout."""
    from .utils import _rank
    if _rank() != 0:
        return
    if log_type not in _LOG_ENTRIES:
        raise _problem(ValueError, "log",
                       f"log_type '{log_type}' not recognised. Options are 'properties', 'conditions', "
                       "'matrices', 'initial time', 'time', or 'mesh'")
    with open(log_file_name, "a") as handle:
        handle.write(_LOG_ENTRIES[log_type](*args) + "\n")
        handle.flush()
