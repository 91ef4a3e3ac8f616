"""ctypes binding of libfedm_hip.so (include/fedm_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).
There is no CPU fallback: a missing library or a missing GPU is an error.
"""
import ctypes as C
import os
from pathlib import Path

MAX_SPECIES, MAX_TERMS, MAX_REACTIONS, MAX_TAGS = 4, 6, 8, 8
MAX_QP, MAX_FQP, MAX_EXT_NODES = 32, 8, 10

EQ_TYPES = {"reaction": 0, "diffusion-reaction": 1, "drift-diffusion-reaction": 2}
BC_KINDS = {"zero flux": 0, "Neumann": 1}
DIVERGED = {1: "maximum number of Newton iterations reached",
            2: "NaN or Inf in the residual", 3: "linear solve (GMRES) did not converge"}


class TermSumC(C.Structure):
    _fields_ = [("n_terms", C.c_int32), ("pad_", C.c_int32),
                ("c", C.c_double * MAX_TERMS), ("p", C.c_double * MAX_TERMS),
                ("q", C.c_double * MAX_TERMS), ("r", C.c_double * MAX_TERMS)]


class ModelDesc(C.Structure):
    _fields_ = [
        ("n_species", C.c_int32), ("poisson", C.c_int32), ("axisymmetric", C.c_int32),
        ("n_reactions", C.c_int32),
        ("eq_type", C.c_int32 * MAX_SPECIES), ("Z", C.c_double * MAX_SPECIES),
        ("mu", TermSumC * MAX_SPECIES), ("D", TermSumC * MAX_SPECIES),
        ("has_drift_w", C.c_int32 * MAX_SPECIES), ("drift_w", (C.c_double * 2) * MAX_SPECIES),
        ("k", TermSumC * MAX_REACTIONS),
        ("power", (C.c_int32 * MAX_SPECIES) * MAX_REACTIONS),
        ("net", (C.c_int32 * MAX_SPECIES) * MAX_REACTIONS),
        ("charge_over_eps", C.c_double),
        ("n_tags", C.c_int32),
        ("bc_kind", (C.c_int32 * MAX_SPECIES) * MAX_TAGS),
        ("n_qp", C.c_int32), ("n_fqp", C.c_int32),
        ("qp_x", C.c_double * MAX_QP), ("qp_y", C.c_double * MAX_QP), ("qp_w", C.c_double * MAX_QP),
        ("fqp_t", C.c_double * MAX_FQP), ("fqp_w", C.c_double * MAX_FQP),
        ("ext_nodes", C.c_int32 * MAX_SPECIES),
        ("ext_B", (C.c_double * MAX_EXT_NODES) * MAX_QP),
        ("linear_representation", C.c_int32), ("pad2_", C.c_int32),
    ]


GD_MAX_SPECIES, GD_MAX_REACTIONS = 6, 16


class GdDesc(C.Structure):
    _fields_ = [
        ("n_species", C.c_int32), ("n_reactions", C.c_int32), ("n_tags", C.c_int32),
        ("axisymmetric", C.c_int32), ("N0", C.c_double), ("charge_over_eps", C.c_double),
        ("eq_type", C.c_int32 * GD_MAX_SPECIES), ("grad_diffusion", C.c_int32 * GD_MAX_SPECIES),
        ("is_ion", C.c_int32 * GD_MAX_SPECIES), ("sign", C.c_double * GD_MAX_SPECIES),
        ("vth", C.c_double * GD_MAX_SPECIES), ("vth_e_coef", C.c_double),
        ("power", (C.c_int32 * GD_MAX_SPECIES) * GD_MAX_REACTIONS),
        ("net", (C.c_int32 * GD_MAX_SPECIES) * GD_MAX_REACTIONS),
        ("energy_loss", C.c_double * GD_MAX_REACTIONS),
        ("ref", (C.c_double * GD_MAX_SPECIES) * MAX_TAGS), ("gamma", C.c_double * MAX_TAGS),
        ("we_secondary", C.c_double),
        ("n_qp", C.c_int32), ("n_fqp", C.c_int32),
        ("qp_x", C.c_double * MAX_QP), ("qp_y", C.c_double * MAX_QP), ("qp_w", C.c_double * MAX_QP),
        ("fqp_t", C.c_double * MAX_FQP), ("fqp_w", C.c_double * MAX_FQP),
        ("energy_Ei", C.c_double), ("mean_energy_form", C.c_int32),
    ]


GD_ME_FORMS = {None: 0, "unknown_ratio": 1}


class GdFieldProg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("table", C.c_int32), ("arg", C.c_int32), ("src_row", C.c_int32),
                ("scale", C.c_double)]


GDP = {"keep": 0, "table": 1, "scaled_row": 2, "me_old": 3, "me": 4, "ue_old": 5}
GDP_ARG = {"energy": 0, "redfield": 1}


class MeshDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_int32), ("n_cells", C.c_int32),
                ("coords", C.POINTER(C.c_double)), ("cells", C.POINTER(C.c_int32)),
                ("facet_tags", C.POINTER(C.c_int8)),
                ("n_dirichlet", C.c_int32),
                ("dirichlet_dofs", C.POINTER(C.c_int32)),
                ("dirichlet_vals", C.POINTER(C.c_double)),
                ("n_owned_vertices", C.c_int32),
                ("n_identity_vertices", C.c_int32), ("identity_vertices", C.POINTER(C.c_int32)),
                ("halo_depth", C.c_int32)]


class NewtonOpts(C.Structure):
    _fields_ = [("rtol", C.c_double), ("atol", C.c_double), ("stol", C.c_double),
                ("max_it", C.c_int32), ("ksp_restart", C.c_int32),
                ("ksp_rtol", C.c_double), ("ksp_atol", C.c_double),
                ("ksp_max_it", C.c_int32), ("watch_component", C.c_int32)]


class NewtonReport(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32),
                ("linear_iterations", C.c_int32), ("reason", C.c_int32),
                ("fnorm0", C.c_double), ("fnorm", C.c_double)]


class Csr(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("n_cols", C.c_int32),
                ("indptr", C.POINTER(C.c_int64)), ("indices", C.POINTER(C.c_int32)),
                ("values", C.POINTER(C.c_double))]


ALLREDUCE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int32, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int32, C.c_void_p)

# FEDM_HIP_LIB points at another build of the same C ABI (kernel experiments, system installs)
LIB_PATH = Path(os.environ.get("FEDM_HIP_LIB") or Path(__file__).resolve().parent / "libfedm_hip.so")

_P = C.c_void_p
_D = C.POINTER(C.c_double)
_SIGNATURES = {
    "fedm_last_error": (C.c_char_p, []),
    "fedm_abi_version": (C.c_int, []),
    "fedm_ctx_create": (C.c_int, [C.POINTER(MeshDesc), C.POINTER(ModelDesc), C.c_int, C.POINTER(_P)]),
    "fedm_ctx_create_gd": (C.c_int, [C.POINTER(MeshDesc), C.POINTER(GdDesc), C.c_int, C.POINTER(_P)]),
    "fedm_gd_set_fields": (C.c_int, [_P, _D]),
    "fedm_gd_get_fields": (C.c_int, [_P, _D]),
    "fedm_gd_prep_setup": (C.c_int, [_P, C.POINTER(Csr), C.c_int, C.POINTER(C.c_int32), _D, _D,
                                     C.POINTER(GdFieldProg)]),
    "fedm_gd_prep_step": (C.c_int, [_P]),
    "fedm_gd_update_mean_energy": (C.c_int, [_P]),
    "fedm_get_state_old": (C.c_int, [_P, _D]),
    "fedm_ctx_destroy": (None, [_P]),
    "fedm_set_state": (C.c_int, [_P, _D, _D, _D]),
    "fedm_get_state": (C.c_int, [_P, _D]),
    "fedm_shift_state": (C.c_int, [_P]),
    "fedm_reset_state": (C.c_int, [_P]),
    "fedm_state_snapshot": (C.c_int, [_P]),
    "fedm_state_restore": (C.c_int, [_P]),
    "fedm_set_step": (C.c_int, [_P, C.c_double, C.c_double]),
    "fedm_set_dirichlet_values": (C.c_int, [_P, _D]),
    "fedm_set_ext_source": (C.c_int, [_P, C.c_int, _D]),
    "fedm_ext_source_program": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, _D, C.c_int]),
    "fedm_ext_source_eval": (C.c_int, [_P, C.c_int, _D]),
    "fedm_residual": (C.c_int, [_P, _D, _D]),
    "fedm_jacobian": (C.c_int, [_P]),
    "fedm_jacobian_nnz": (C.c_int64, [_P]),
    "fedm_jacobian_csr": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int32), _D]),
    "fedm_spmv": (C.c_int, [_P, _D, _D]),
    "fedm_newton_solve": (C.c_int, [_P, C.POINTER(NewtonOpts), C.POINTER(NewtonReport)]),
    "fedm_poisson_solve": (C.c_int, [_P, C.c_double, C.c_int, C.POINTER(C.c_int)]),
    "fedm_block_nnz": (C.c_int64, [_P]),
    "fedm_block_csr": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32), _D]),
    "fedm_jacobian_poisson_only": (C.c_int, [_P]),
    "fedm_amg_setup": (C.c_int, [_P, C.c_int, C.POINTER(Csr), C.POINTER(Csr), C.POINTER(Csr), _D,
                                 C.c_int, C.c_double]),
    "fedm_amg_setup_poly": (C.c_int, [_P, C.c_int, C.POINTER(Csr), C.POINTER(Csr), C.POINTER(Csr), _D,
                                      C.c_int, _D, C.c_int]),
    "fedm_amg_clear": (C.c_int, [_P]),
    "fedm_amg_set_global_hierarchy": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(Csr), C.POINTER(Csr),
                                                 C.POINTER(Csr), _D, C.c_int, C.c_double]),
    "fedm_set_fieldsplit": (C.c_int, [_P, C.c_int, _D]),
    "fedm_set_fieldsplit_alternative": (C.c_int, [_P, C.c_int, _D, C.c_double, C.c_double]),
    "fedm_fieldsplit_policy": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "fedm_amg_aggregate": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32)]),
    "fedm_field_error": (C.c_int, [_P, C.c_int, _D]),
    "fedm_time_kernel": (C.c_int, [_P, C.c_int, C.c_int, _D]),
    "fedm_copy_bandwidth": (C.c_int, [C.c_int, C.c_int64, C.c_int, _D]),
    "fedm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "fedm_comm_init_rccl": (C.c_int, [_P, C.c_int] + [C.POINTER(C.c_int32)] * 4
                            + [C.c_void_p, C.c_int, C.c_int]),
    "fedm_comm_init_callbacks": (C.c_int, [_P, C.c_int] + [C.POINTER(C.c_int32)] * 4
                                 + [ALLREDUCE_FN, EXCHANGE_FN, C.c_void_p, C.c_int, C.c_int]),
    "fedm_sync_ghosts": (C.c_int, [_P]),
    "fedm_comm_stats": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "fedm_time_comm": (C.c_int, [_P, C.c_int, C.c_int, _D]),
    "fedm_debug_comm_fault": (C.c_int, [C.c_int, C.POINTER(C.c_int64)]),
    "fedm_debug_comm_roundtrip": (C.c_int, [_P, _D, _D, C.c_int]),
    "fedm_pattern_stats": (C.c_int, [C.POINTER(MeshDesc), C.POINTER(C.c_int64)]),
    "fedm_pattern_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "fedm_fieldsplit_tiles_info": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "fedm_fieldsplit_tiles_stats": (C.c_int, [C.POINTER(MeshDesc), C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "fedm_debug_fieldsplit_apply": (C.c_int, [_P, _D, _D]),
    "fedm_debug_fieldsplit_tiles": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "fedm_debug_species_planes_check": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "fedm_profile": (C.c_int, [_P, C.c_int]),
    "fedm_profile_read": (C.c_int, [_P, C.c_int, _D, C.POINTER(C.c_int64)]),
    "fedm_set_assembly": (C.c_int, [_P, C.c_int]),
    "fedm_set_preconditioner_side": (C.c_int, [_P, C.c_int]),
    "fedm_plane_masks": (C.c_int, [_P, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "fedm_set_fieldsplit_order": (C.c_int, [_P, C.c_int]),
    "fedm_sizes": (C.c_int, [_P] + [C.POINTER(C.c_int64)] * 6),
}

_lib = None


# opcodes of the expression programs (include/fedm_hip.h, FEDM_OP_*)
EXPR_OPS = {"const": 0, "x": 1, "param": 2, "add": 3, "sub": 4, "mul": 5, "div": 6, "pow": 7, "neg": 8,
            "exp": 9, "log": 10, "sqrt": 11, "sin": 12, "cos": 13, "tan": 14, "fabs": 15, "abs": 15,
            "tanh": 16, "atan": 17}
EXPR_MAX_OPS, EXPR_MAX_PARAMS, EXPR_STACK = 256, 16, 24


ABI_VERSION = 5          # include/fedm_hip.h FEDM_ABI_VERSION


def exported_symbols():
    return sorted(_SIGNATURES)


def _share_the_hip_runtime_with_torch():
    """PyTorch-ROCm wheels carry their own libamdhip64.so (soname libamdhip64.so.7, asked for by file name);
    libfedm_hip.so asks for the soname.  Loaded after torch it binds to torch's copy; loaded BEFORE torch it
    would bring /opt/rocm's copy in, torch would then add its own, and the second runtime of the process
    finds no device (measured on the MI355X box: fedm_ctx_create -3 after torch.cuda.set_device).  The
    multi-GPU layer needs torch.distributed in the same process, so where torch is installed it goes first."""
    import importlib.util
    import sys
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401  (importing does not initialise the GPU)


def load():
    """Load libfedm_hip.so and attach signatures.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  fedm_amd has no CPU fallback.")
    _share_the_hip_runtime_with_torch()
    lib = C.CDLL(os.fspath(LIB_PATH))
    lib.fedm_abi_version.restype = C.c_int
    if lib.fedm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} implements ABI version {lib.fedm_abi_version()}, this binding was "
                           f"written against {ABI_VERSION} (include/fedm_hip.h FEDM_ABI_VERSION): rebuild it")
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().fedm_last_error().decode()
