"""ctypes binding of ``liboasisx_hip.so`` (declared in ``include/oasisx_hip.h``).

The HIP library IS the compute path: there is no CPU fallback.  ``load()`` raises
if the shared object is missing, and every wrapper raises ``OasisxHipError`` with the
library's message when a call fails.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OX_LIB_PATH") or os.path.join(_HERE, "liboasisx_hip.so")  # override: tuning builds

KSP_CG, KSP_BCGS, KSP_CG_SINGLE, KSP_BCGS_MERGED, KSP_CG_MERGED = 1, 2, 3, 4, 5
ROW_BLOCK_WAVES, ROW_BLOCK_LDS = 8, 134 * 1024  # OX_ROW_BLOCK_WAVES / OX_ROW_BLOCK_LDS of include/oasisx_hip.h
CONVERGED_RTOL, CONVERGED_ATOL, CONVERGED_ITS = 2, 3, 4
DIVERGED_ITS, DIVERGED_DTOL, DIVERGED_BREAKDOWN, DIVERGED_NANORINF = -3, -4, -5, -9


class OasisxHipError(RuntimeError):
    pass


class ox_sell(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int64),
        ("n_cols", C.c_int64),
        ("n_slices", C.c_int32),
        ("n_dict", C.c_int32),
        ("slice_ptr", C.c_void_p),
        ("cols", C.c_void_p),
        ("vals", C.c_void_p),
        ("cols16", C.c_void_p),
        ("cbase", C.c_void_p),
        ("vcode", C.c_void_p),
        ("vdict", C.c_void_p),
        ("ib_slices", C.c_void_p),
        ("n_interior", C.c_int32),
        ("n_wb_interior", C.c_int32),
        ("ps_ptr", C.c_void_p),
        ("ps_code", C.c_void_p),
        ("ps_base", C.c_void_p),
        ("wb_slices", C.c_void_p),
        ("wb_waves", C.c_void_p),
        ("wb_ptr", C.c_void_p),
        ("wlist", C.c_void_p),
        ("wt_ptr", C.c_void_p),
        ("wcode", C.c_void_p),
        ("wvcode", C.c_void_p),
        ("n_wblocks", C.c_int32),
        ("w_max", C.c_int32),
        ("levels", C.c_int32),
        ("w_cap", C.c_int32),
    ]


class ox_window_info(C.Structure):
    _fields_ = [
        ("n_wblocks", C.c_int32),
        ("w_max", C.c_int32),
        ("n_list", C.c_int64),
        ("n_tiles", C.c_int64),
        ("n_over_16bit", C.c_int64),
        ("wb_slices", C.c_void_p),
        ("wb_waves", C.c_void_p),
        ("wb_ptr", C.c_void_p),
        ("wlist", C.c_void_p),
        ("wt_ptr", C.c_void_p),
        ("wcode", C.c_void_p),
    ]


class ox_cells(C.Structure):
    _fields_ = [
        ("gdim", C.c_int32),
        ("reserved", C.c_int32),
        ("n_cells", C.c_int64),
        ("geom", C.c_void_p),
    ]


class ox_adj(C.Structure):
    _fields_ = [
        ("n_slices", C.c_int32),
        ("nd", C.c_int32),
        ("adj_ptr", C.c_void_p),
        ("adj_cell", C.c_void_p),
        ("adj_loc", C.c_void_p),
    ]


class ox_mesh_info(C.Structure):
    _fields_ = [
        ("gdim", C.c_int32),
        ("tile_bits", C.c_int32),
        ("n_vertices", C.c_int64),
        ("n_cells", C.c_int64),
        ("lo", C.c_double * 3),
        ("span", C.c_double * 3),
        ("coords", C.c_void_p),
        ("cells", C.c_void_p),
        ("cell_perm", C.c_void_p),
        ("cells_struct", ox_cells),
        ("lattice", C.c_int32),
        ("reserved", C.c_int32),
    ]


class ox_pattern_info(C.Structure):
    _fields_ = [
        ("sell", ox_sell),
        ("size", C.c_int64),
        ("nnz", C.c_int64),
        ("n_compressed", C.c_int64),
        ("row_len", C.c_void_p),
        ("n_bins", C.c_int32),
        ("bin_ptr_host", C.c_void_p),
        ("bin_width_host", C.c_void_p),
        ("bin_slices", C.c_void_p),
        ("widths_host", C.c_void_p),
        ("n_row_blocks", C.c_int32),
        ("reserved_rb", C.c_int32),
        ("row_blk_ptr", C.c_void_p),
        ("row_blk_entries", C.c_int64),
    ]


class ox_space_info(C.Structure):
    _fields_ = [
        ("degree", C.c_int32),
        ("nd", C.c_int32),
        ("pw", C.c_int32),
        ("gdim", C.c_int32),
        ("n_dofs", C.c_int64),
        ("n_edges", C.c_int64),
        ("n_pairs", C.c_int64),
        ("cell_dofs", C.c_void_p),
        ("x", C.c_void_p),
        ("rank_initial", C.c_void_p),
        ("edge_keys", C.c_void_p),
        ("adj", ox_adj),
        ("adj_pos", C.c_void_p),
        ("pair_start", C.c_void_p),
        ("pattern", ox_pattern_info),
        ("n_faces", C.c_int64),
        ("face_keys", C.c_void_p),
    ]


class ox_rect_info(C.Structure):
    _fields_ = [
        ("pw", C.c_int32),
        ("pos", C.c_void_p),
        ("pattern", ox_pattern_info),
    ]


class ox_ksp_result(C.Structure):
    _fields_ = [
        ("reason", C.c_int32 * 4),
        ("its", C.c_int32 * 4),
        ("rnorm", C.c_double * 4),
        ("bnorm", C.c_double * 4),
        ("resumed", C.c_int32 * 4),
    ]


class ox_ksp_options(C.Structure):
    _fields_ = [
        ("rtol", C.c_double),
        ("atol", C.c_double),
        ("divtol", C.c_double),
        ("max_it", C.c_int32),
        ("nonzero_guess", C.c_int32),
        ("check_every", C.c_int32),
        ("max_restarts", C.c_int32),
        ("fold_blocks", C.c_int32),
        ("run_ahead", C.c_int32),
        ("ax0", C.c_void_p),
        ("dinv_code", C.c_void_p),
        ("dinv_dict", C.c_void_p),
        ("n_dinv_dict", C.c_int32),
        ("reserved", C.c_int32),
    ]


_P = C.c_void_p
_I = C.c_int
_L = C.c_int64
_D = C.c_double

# name -> (restype, argtypes); the complete export list of include/oasisx_hip.h
SIGNATURES = {
    "ox_version": (_I, []),
    "ox_last_error": (C.c_char_p, []),
    "ox_sell_kv": (_I, []),
    "ox_device_info": (_I, [C.POINTER(_I), C.c_char_p, _I]),
    "ox_mesh_create": (_I, [_P, _L, _P, _L, _I, _I, _I, C.POINTER(_P)]),
    "ox_mesh_view": (_I, [_P, C.POINTER(ox_mesh_info)]),
    "ox_mesh_destroy": (_I, [_P]),
    "ox_space_create": (_I, [_P, _I, _I, C.POINTER(_P)]),
    "ox_space_create_ordered": (_I, [_P, _I, _I, _I, C.POINTER(_P)]),
    "ox_mesh_create_sub": (_I, [_P, _L, _P, _L, _I, _I, C.POINTER(_D), C.POINTER(_D), _I, _I, _L, C.POINTER(_P)]),
    "ox_space_create_part": (_I, [_P, _I, _I, _P, _L, _I, _L, C.POINTER(_P)]),
    "ox_space_view": (_I, [_P, C.POINTER(ox_space_info)]),
    "ox_space_windows": (_I, [_P, C.POINTER(ox_window_info)]),
    "ox_space_windows_split": (_I, [_P, _I, C.POINTER(ox_window_info)]),
    "ox_space_destroy": (_I, [_P]),
    "ox_rect_create": (_I, [_P, _P, C.POINTER(_P)]),
    "ox_rect_view": (_I, [_P, C.POINTER(ox_rect_info)]),
    "ox_rect_destroy": (_I, [_P]),
    "ox_value_dictionary": (_I, [_P, _L, _I, _P, _P, C.POINTER(_I), _P]),
    "ox_pair_stream_size": (_I, [C.POINTER(ox_sell), _P, _P, C.POINTER(_L), _P]),
    "ox_pair_stream_fill": (_I, [C.POINTER(ox_sell), _P, _P, _P, _P, C.POINTER(_L), _P]),
    "ox_malloc": (_I, [C.c_size_t, C.POINTER(_P)]),
    "ox_free": (_I, [_P]),
    "ox_memset": (_I, [_P, _I, C.c_size_t, _P]),
    "ox_synchronize": (_I, [_P]),
    "ox_spmv": (_I, [C.POINTER(ox_sell), _P, _P, _I, _P, _P]),
    "ox_sell_compress_cols": (_I, [C.POINTER(ox_sell), _P, _P, C.POINTER(_L), _P]),
    "ox_spmv_multi": (_I, [_I, _I, C.POINTER(ox_sell), _P, _P, _D, _P, _P, _P]),
    "ox_assemble_rect": (_I, [_I, _I, _I, C.POINTER(ox_cells), C.POINTER(ox_adj), _P, _I, C.POINTER(ox_sell), _P]),
    "ox_axpby": (_I, [_L, _D, _P, _D, _P, _P, _P]),
    "ox_dot": (_I, [_L, _I, _P, _P, C.POINTER(_D), _P, _P]),
    "ox_set_bc": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "ox_scatter_add": (_I, [_P, _P, _P, _L, _I, _I, _D, _P]),
    "ox_zero_rows": (_I, [C.POINTER(ox_sell), _P, _L, _D, _P]),
    "ox_zero_rows_au": (_I, [C.POINTER(ox_sell), _P, _L, _D, _P, _P, _I, _P]),
    "ox_zero_rows_cols": (_I, [C.POINTER(ox_sell), _P, _D, _P]),
    "ox_assemble_matrix": (_I, [_I, _I, C.POINTER(ox_cells), _P, C.POINTER(ox_adj), _P, _I,
                                C.POINTER(ox_sell), _I, C.POINTER(_L), _P, C.POINTER(C.c_int32), _P]),
    "ox_assemble_weights": (_I, [_I, C.POINTER(ox_cells), C.POINTER(ox_adj), _L, _P, _P]),
    "ox_assemble_load_vector": (_I, [C.POINTER(ox_cells), C.POINTER(ox_adj), _L, _I, _I, _P, _P, _P, _P]),
    "ox_assemble_first": (_I, [_I, C.POINTER(ox_cells), _P, C.POINTER(ox_adj), _P, _I,
                               C.POINTER(ox_sell), C.POINTER(ox_sell), C.POINTER(ox_sell), _P, _P, _P, _P, _D,
                               _D, _I, C.POINTER(_L), _P, C.POINTER(C.c_int32), _P]),
    "ox_assemble_first_au": (_I, [_I, C.POINTER(ox_cells), _P, C.POINTER(ox_adj), _P, _I,
                                  C.POINTER(ox_sell), C.POINTER(ox_sell), C.POINTER(ox_sell), _P, _P, _P, _P, _D,
                                  _D, _I, C.POINTER(_L), _P, C.POINTER(C.c_int32), _P, _P]),
    "ox_assemble_matrix_blocks": (_I, [_I, _I, C.POINTER(ox_cells), _P, C.POINTER(ox_adj), _P, _I, C.POINTER(ox_sell), _I, _P,
                                       _L, _P]),
    "ox_assemble_first_blocks": (_I, [_I, C.POINTER(ox_cells), _P, C.POINTER(ox_adj), _P, _I,
                                      C.POINTER(ox_sell), C.POINTER(ox_sell), C.POINTER(ox_sell), _P, _P, _P, _P, _D,
                                      _D, _I, _P, _L, _P, _P]),
    "ox_assemble_grad_vector": (_I, [_I, _I, _I, C.POINTER(ox_cells), _P, C.POINTER(ox_adj), _L, _P,
                                     _P, _D, _P, _P]),
    "ox_assemble_div_vector": (_I, [_I, _I, C.POINTER(ox_cells), _P, C.POINTER(ox_adj), _L, _P, _D,
                                    _P, _P]),
    "ox_dg1_grad_rhs": (_I, [_I, C.POINTER(ox_cells), _P, _P, _P, _P]),
    "ox_dg1_mass": (_I, [_I, C.POINTER(ox_cells), _I, _P, _P, _P]),
    "ox_jacobi_setup": (_I, [C.POINTER(ox_sell), _P, _P]),
    "ox_ksp_work_bytes": (C.c_size_t, [_L, _L, _I, _I]),
    "ox_ksp_solve": (_I, [_I, C.POINTER(ox_sell), _P, _P, _P, _I, _D, _D, _I, _I, _I, _I, _P,
                          C.c_size_t, C.POINTER(ox_ksp_result), _P, _P]),
    "ox_ksp_solve_ax0": (_I, [_I, C.POINTER(ox_sell), _P, _P, _P, _I, _D, _D, _I, _I, _I, _I, _P,
                              C.c_size_t, C.POINTER(ox_ksp_result), _P, _P, _P]),
    "ox_ksp_solve_dc": (_I, [_I, C.POINTER(ox_sell), _P, _P, _P, _I, _D, _D, _I, _I, _I, _I, _P,
                             C.c_size_t, C.POINTER(ox_ksp_result), _P, _P, _P, _P, _P, _I]),
    "ox_remove_mean": (_I, [_L, _L, _P, _P, _D, _P, _P]),
    "ox_window_retile": (_I, [C.POINTER(ox_sell), _P, _P, _I, _P, _P]),
    "ox_ksp_default_fold_blocks": (_I, []),
    "ox_ksp_kernels_per_iteration": (_I, [_I, C.POINTER(ox_sell), _I, _I, _I, _I]),
    "ox_ksp_options_default": (_I, [C.POINTER(ox_ksp_options)]),
    "ox_ksp_solve_opt": (_I, [_I, C.POINTER(ox_sell), _P, _P, _P, _I, C.POINTER(ox_ksp_options), _P, C.c_size_t,
                              C.POINTER(ox_ksp_result), _P, _P]),
    "ox_ksp_work_bytes_for": (C.c_size_t, [C.POINTER(ox_sell), _I, _I]),
    "ox_profile_begin": (_I, [_I, _I]),
    "ox_profile_end": (_I, []),
    "ox_profile_get": (_I, [_I, C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(_D)]),
    "ox_range_push": (_I, [C.c_char_p]),
    "ox_range_pop": (_I, []),
    "ox_comm_unique_id": (_I, [C.c_char_p]),
    "ox_comm_create": (_I, [C.c_char_p, _I, _I, C.POINTER(_P)]),
    "ox_comm_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "ox_comm_destroy": (_I, [_P]),
    "ox_dist_create": (_I, [_P, _I, _I, _I, C.POINTER(C.c_int32), C.POINTER(_L), _P,
                            C.POINTER(_L), _L, _L, C.POINTER(_P)]),
    "ox_p2p_window_bytes": (C.c_size_t, [_I, _L]),
    "ox_p2p_window_create": (_I, [C.c_size_t, C.POINTER(_P), C.c_char_p]),
    "ox_p2p_window_open": (_I, [C.c_char_p, C.POINTER(_P)]),
    "ox_p2p_window_close": (_I, [_P]),
    "ox_p2p_window_free": (_I, [_P]),
    "ox_dist_enable_p2p": (_I, [_P, _P, C.POINTER(_P), C.POINTER(_L), C.POINTER(_L), _D]),
    "ox_dist_p2p_timeout": (_I, [_P, _D]),
    "ox_dist_disable_p2p": (_I, [_P]),
    "ox_dist_set_overlap": (_I, [_P, _I]),
    "ox_dist_set_p2p_release": (_I, [_P, _I]),
    "ox_dist_status": (_I, [_P]),
    "ox_dist_create_custom": (_I, [_I, _I, _I, C.POINTER(C.c_int32), C.POINTER(_L), _P, C.POINTER(_L), _L, _L,
                                   _P, _P, _P, C.POINTER(_P)]),
    "ox_memcpy": (_I, [_P, _P, C.c_size_t, _I, _P]),
    "ox_dist_destroy": (_I, [_P]),
    "ox_halo_forward": (_I, [_P, _P, _I, _P]),
    "ox_allreduce_sum": (_I, [_P, _P, _I, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OasisxHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  oasisx_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().ox_last_error().decode(errors="replace")
        raise OasisxHipError(f"{what} failed ({rc}): {msg}")


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream():
    """hipStream_t of torch's current stream on the current device (0 on CPU)."""
    import torch

    if torch.cuda.is_available():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(0)
