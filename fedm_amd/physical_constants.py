"""Physical constants under the names user scripts import from ``fedm.physical_constants``.

The numerical values have to be the reference's digit for digit (fedm/physical_constants.py:5-15):
they enter the weak forms (e/eps0 in the Poisson source, kB*T/e in the Einstein relation) and
the goldens were produced with them.  They are CODATA 2014 / 2018 figures.
"""

_SI = (
    # symbol                value                 unit       what
    ("elementary_charge",   1.6021766208e-19,     "C",       "elementary charge e"),
    ("me",                  9.10938356e-31,       "kg",      "electron rest mass"),
    ("epsilon_0",           8.854187817e-12,      "F/m",     "vacuum permittivity"),
    ("kB",                  1.38064852e-23,       "J/K",     "Boltzmann constant"),
    ("kB_eV",               8.6173303e-5,         "eV/K",    "Boltzmann constant in electron volts"),
    ("speed_of_light",      2.99792458e8,         "m/s",     "speed of light in vacuum"),
    ("h_planck",            6.62607015e-34,       "J s",     "Planck constant"),
    ("mag_perm",            1.25663706212e-6,     "N/A^2",   "vacuum magnetic permeability"),
    ("N_avogadro",          6.02214076e23,        "1/mol",   "Avogadro constant"),
    ("Ry_const",            10973731.568160,      "1/m",     "Rydberg constant"),
    ("M_atomic",            1.66053906660e-27,    "kg",      "atomic mass constant"),
)

units = {name: unit for name, _, unit, _ in _SI}
globals().update({name: value for name, value, _, _ in _SI})
__all__ = [name for name, _, _, _ in _SI]
