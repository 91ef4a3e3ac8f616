"""fedm_amd: the assemble()/solve() hot path of INP-PM/FEDM on MI355X.

Keeps FEDM's Python surface (``fedm_amd.functions``, ``fedm_amd.file_io``,
``fedm_amd.physical_constants``) and runs residual/Jacobian assembly, SpMV,
GMRES and Newton as hand-written HIP kernels (``libfedm_hip.so``).
"""
from . import physical_constants  # noqa: F401

__all__ = ["physical_constants", "functions", "file_io", "device", "termsum", "mesh"]
