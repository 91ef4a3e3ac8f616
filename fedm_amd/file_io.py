"""FEDM's input-deck readers (``fedm.file_io``) as drop-ins.

Same function names, argument meaning, return shapes and error types as
fedm/file_io.py:123-535, including the format quirks the decks rely on:

* species/reaction matching by substring count (file_io.py:288-289),
* ``"const."`` (with the dot) accepted for transport coefficients (:412) while the
  interpolation step only knows ``"const"`` (functions.py:583),
* a missing mobility file means dependence ``0`` (:346-359, :447-450),
* the regular expressions of :309, :323, :486-487 (they are part of the file format).

Result writers (``output_files``, ``file_output``, ``mesh_statistics``) are DOLFIN I/O and
outside the hot path; they are not provided.
"""
import itertools
import re
from pathlib import Path
from textwrap import dedent
from typing import Any, List

import numpy as np

from .utils import comma_separated, print_rank_0


# ---------------------------------------------------------------------------
# file_io.py:22-117 -- path manager
# ---------------------------------------------------------------------------
def truncate_file(path):
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    open(path, "w").close()


class Files:
    """``file_input`` / ``output_folder_path`` / ``error_file`` / ``model_log`` with the
    reference's semantics: assigning ``file_input`` to a missing directory raises
    RuntimeError; log files are truncated on first access per output directory."""

    def __init__(self):
        self._input_dir = Path.cwd() / "file_input"
        self._output_dir = Path.cwd() / "output"
        self._touched = set()

    @property
    def file_input(self):
        return self._input_dir

    @file_input.setter
    def file_input(self, value):
        value = Path(value)
        if not value.is_dir():
            raise RuntimeError(f"fedm.files.file_input: '{value}' is not a directory")
        self._input_dir = value

    @property
    def output_folder_path(self):
        return self._output_dir

    @output_folder_path.setter
    def output_folder_path(self, value):
        value = Path(value)
        if value.resolve() != self._output_dir.resolve():
            self._touched.clear()
        if not value.is_dir():
            value.mkdir()
        self._output_dir = value

    def _log_file(self, name):
        result = self.output_folder_path / name
        if name not in self._touched:
            truncate_file(result)
            self._touched.add(name)
        return result

    @property
    def error_file(self):
        return self._log_file("relative error.log")

    @property
    def model_log(self):
        return self._log_file("model.log")


files = Files()


# ---------------------------------------------------------------------------
# file_io.py:123-247 -- small readers
# ---------------------------------------------------------------------------
def no_convert(x):
    return x


def decomment(lines):
    """Yield the non-empty part of each line before '#'."""
    for line in lines:
        text = line.split("#", 1)[0].strip()
        if text:
            yield text


def read_single_value(file_name, convert=no_convert):
    with open(file_name, "r", encoding="utf8") as f_input:
        return convert(next(decomment(f_input)))


def read_single_float(file_name, convert=no_convert):
    return read_single_value(file_name, convert=float)


def read_single_string(file_name):
    return read_single_value(file_name, convert=str)


def read_and_decomment(file_name):
    with open(file_name, "r") as f_input:
        return list(decomment(f_input))


def read_two_columns(file_name):
    rows = [line.split() for line in read_and_decomment(file_name)]
    return [float(r[0]) for r in rows], [float(r[1]) for r in rows]


def flatten(input_list):
    return list(itertools.chain.from_iterable(input_list))


def flatten_float(input_list):
    return [float(x) for x in flatten(input_list)]


# ---------------------------------------------------------------------------
# file_io.py:250-296
# ---------------------------------------------------------------------------
def read_speclist(file_path):
    """(n, species names, property file names, names used for transport files)."""
    lines = [line.replace("file:", "").split()
             for line in read_and_decomment(Path(file_path) / "speclist.cfg") if "file:" in line]
    names = [line[0] for line in lines]
    prop_files = [line[1] for line in lines]
    tc_names = [line[1].split(".")[0] for line in lines]
    return len(names), names, prop_files, tc_names


def reaction_matrices(path, species):
    """Power, loss and gain matrices (n_reactions x n_species, int) from reacscheme.cfg."""
    reactions = [line.partition(" Type:")[0] for line in read_and_decomment(Path(path) / "reacscheme.cfg")]
    lhs = [r.partition(" -> ")[0].rstrip() for r in reactions]
    rhs = [r.partition(" -> ")[2].rstrip() for r in reactions]
    l_matrix = np.array([[side.count(sp) for sp in species] for side in lhs], dtype=int).reshape(len(reactions), len(species))
    g_matrix = np.array([[side.count(sp) for sp in species] for side in rhs], dtype=int).reshape(len(reactions), len(species))
    power_matrix = l_matrix
    net = l_matrix - g_matrix
    loss_matrix = np.where(net > 0, net, 0)
    gain_matrix = np.where(net < 0, -net, 0)
    return power_matrix, loss_matrix, gain_matrix


# ---------------------------------------------------------------------------
# file_io.py:299-359
# ---------------------------------------------------------------------------
def rate_coefficient_file_names(path):
    regex = re.compile(r"kfile: ([A-Za-z0-9_]+.[A-Za-z0-9_]+)")
    names = flatten([regex.findall(line) for line in read_and_decomment(Path(path) / "reacscheme.cfg")])
    return [Path(path) / "rate_coefficients" / name for name in names]


def read_energy_loss(path):
    regex = re.compile(r"Uin:\s?([+-]?\d+.\d+[eE]?[-+]?\d+|0|1.0)")
    values = flatten_float([regex.findall(line) for line in read_and_decomment(Path(path) / "reacscheme.cfg")])
    print_rank_0(values)
    return values


def read_dependence(file_name):
    file_name = Path(file_name)
    if not file_name.is_file():
        raise FileNotFoundError(f"fedm.read_dependence: file '{file_name}' not found")
    with open(file_name, "r", encoding="utf8") as f_input:
        for line in f_input:
            if "Dependence:" in line:
                return line.split()[2]
    raise RuntimeError(f"fedm.read_dependence: Did not find dependence in file '{file_name}'")


def read_dependences(file_names, zero_if_file_missing=False):
    dependences = []
    for file_name in file_names:
        try:
            dependences.append(read_dependence(file_name))
        except FileNotFoundError:
            if not zero_if_file_missing:
                raise
            dependences.append(0)
    return dependences


# ---------------------------------------------------------------------------
# file_io.py:362-475
# ---------------------------------------------------------------------------
def read_rate_coefficients(rc_file_names, k_dependences):
    if len(rc_file_names) != len(k_dependences):
        raise ValueError(
            "fedm.read_rate_coefficients: rc_file_names and k_dependences should be "
            "the same length."
        )
    float_dependences = ["const"]
    str_dependences = ["fun:Te,Tgas", "fun:Tgas"]
    two_col_dependences = ["Umean", "E/N", "ElecDist"]
    all_dependences = float_dependences + str_dependences + two_col_dependences
    for dependence in k_dependences:
        if dependence not in all_dependences:
            raise ValueError(
                f"fedm.read_rate_coefficients: The dependence '{dependence}' is not "
                f"recognised. Options are {comma_separated(all_dependences)}."
            )
    kxs, kys = [], []
    for dependence, name in zip(k_dependences, rc_file_names):
        print_rank_0(name)
        if dependence in two_col_dependences:
            kx, ky = read_two_columns(name)
        elif dependence in float_dependences:
            kx, ky = 0.0, read_single_float(name)
        else:
            kx, ky = 0.0, read_single_string(name)
        kxs.append(kx)
        kys.append(ky)
    return kxs, kys


def read_transport_coefficients(particle_names, transport_type, model):
    """(kxs, kys, dependences).  ``'fun:E'`` entries keep their expression string (the
    reference ``eval``s it later, fedm-streamer.py:237-238; here ``termsum.parse`` does)."""
    path = files.file_input / model / "transport_coefficients"
    if not path.is_dir():
        raise FileNotFoundError(
            f"fedm.read_transport_coefficients: Transport coeff dir '{path}' not found."
        )
    float_dependences = ["const", "const."]
    str_dependences = ["fun:Te,Tgas", "fun:E"]
    two_col_dependences = ["Umean", "E/N", "Tgas", "Te"]
    all_dependences = float_dependences + str_dependences + two_col_dependences
    if transport_type == "Diffusion":
        all_dependences.append("ESR")
    if transport_type == "mobility":
        all_dependences.append(0)
    suffix = "_ND.dat" if transport_type == "Diffusion" else "_Nb.dat"
    file_names = [path / (particle + suffix) for particle in particle_names]
    k_dependences = read_dependences(file_names, zero_if_file_missing=(transport_type == "mobility"))
    for dependence in k_dependences:
        if dependence not in all_dependences:
            err_msg = dedent(
                f"""\
                fedm.read_transport_coefficients: Dependence '{dependence}' not
                recognised. For the transport type '{transport_type}', the possible
                options are {comma_separated(all_dependences)}.
                """
            )
            raise ValueError(err_msg.rstrip().replace("\n", " "))
    kxs, kys = [], []
    for file_name, dependence in zip(file_names, k_dependences):
        if transport_type == "mobility" and dependence == 0:
            kxs.append(0)
            kys.append(0)
            continue
        print_rank_0(file_name)
        if dependence in two_col_dependences:
            kx, ky = read_two_columns(file_name)
        elif dependence == "ESR":
            kx, ky = 0.0, 0.0
        elif dependence in float_dependences:
            kx, ky = 0.0, read_single_float(file_name)
        else:
            kx, ky = 0.0, read_single_string(file_name)
        if dependence == "fun:Te,Tgas":
            raise RuntimeError(
                f"fedm.read_transport_coefficients: ky eval failed, '{ky}'"
            )  # the reference eval()s arbitrary Python here; not supported on purpose
        kxs.append(kx)
        kys.append(ky)
    return kxs, kys, k_dependences


# ---------------------------------------------------------------------------
# file_io.py:478-535
# ---------------------------------------------------------------------------
def read_particle_properties(file_names, model):
    path = files.file_input / model / "species"
    regex_mass = re.compile(r"Mass\s?=\s?([+-]?\d+.\d+[eE]?[-+]?\d+|0|1.0)")
    regex_charge = re.compile(r"Z\s+?=\s+?([+-]?\d+)")
    masses, charges = [], []
    for name in file_names:
        file_name = path / name
        if not file_name.is_file():
            raise RuntimeError(f"fedm.read_particle_properties: File '{file_name}' not found.")
        print_rank_0(file_name)
        mass_found = charge_found = False
        for line in read_and_decomment(file_name):
            print_rank_0(line)
            mass, charge = regex_mass.findall(line), regex_charge.findall(line)
            if mass:
                mass_found = True
                masses.append(float(mass[0]))
            if charge:
                charge_found = True
                charges.append(float(charge[0]))
        if not mass_found:
            raise RuntimeError(f"fedm.read_particle_properties: No mass found in file '{file_name}'.")
        if not charge_found:
            raise RuntimeError(f"fedm.read_particle_properties: No charge found in file '{file_name}'.")
    return masses, charges


def print_time_step(dt):
    print_rank_0("Time step is dt =", dt)


def print_time(t):
    print_rank_0("t =", t)


def numpy_2d_array_to_str(x):
    no_brackets = str(np.asarray(x)).replace("[", "").replace("]", "")
    return "\n".join(row.strip() for row in no_brackets.split("\n"))


def log(log_type, log_file_name, *args):
    """Model log entries with the reference's layout (file_io.py:641-724)."""
    from .utils import _rank, mesh_info
    if _rank() != 0:
        return
    if log_type == "properties":
        gas, model, names, M, charge = args
        log_str = (f"Gas:\t{gas}\n\nmodel:\t{model}\n\nParticle names:\n{names}\n\n"
                   f"Mass:\n{M}\n\nCharge:\n{charge}\n")
    elif log_type == "conditions":
        dt_var, U_w, p0, gap_length, N0, Tgas = args
        body = "\t ".join([f"dt = {dt_var} s,", f"U_w = {U_w} V,", f"p_0 = {p0} Torr,",
                           f"d = {gap_length} m,", f"N_0 = {N0} m^-3,", f"T_gas = {Tgas} K"])
        log_str = f"Simulation conditions:\n{body}\n"
    elif log_type == "matrices":
        gain, loss, power = args
        log_str = (f"Gain matrix:\n{numpy_2d_array_to_str(gain)}\n\nLoss matrix:\n"
                   f"{numpy_2d_array_to_str(loss)}\n\nPower matrix:\n{numpy_2d_array_to_str(power)}\n")
    elif log_type == "initial time":
        log_str = f"Time:\n{args[0]}"
    elif log_type == "time":
        log_str = str(args[0])
    elif log_type == "mesh":
        log_str = mesh_info(args[0])
    else:
        err_msg = dedent(
            f"""\
            fedm.log: log_type '{log_type}' not recognised. Options are 'properties',
            'conditions', 'matrices', 'initial time', 'time', or 'mesh'
            """
        )
        raise ValueError(err_msg.rstrip().replace("\n", " "))
    with open(log_file_name, "a") as log_file:
        log_file.write(log_str)
        log_file.write("\n")
        log_file.flush()
