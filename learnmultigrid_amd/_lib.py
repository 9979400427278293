"""ctypes binding of the C ABI declared in include/lmg.h (liblmg_hip.so, gfx950 only).

There is NO fallback: if the shared library is missing or a call returns a negative
status this module raises.  Nothing here (or anywhere in learnmultigrid_amd) imports
the CPU oracle.
"""
import ctypes
import os
import subprocess

# torch FIRST: PyTorch-ROCm ships its own libamdhip64; if liblmg_hip.so were loaded before it,
# the dynamic linker would bind our kernels to /opt/rocm's copy and the process would end up
# with two HIP runtimes (launches on torch's streams then fail with a HIP runtime error).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# LMG_LIB_PATH selects another build of the same ABI (A/B runs of two kernel versions)
LIB_PATH = os.environ.get("LMG_LIB_PATH") or os.path.join(_HERE, "liblmg_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_c = ctypes
_i64, _i32, _f64, _p, _cp = _c.c_int64, _c.c_int32, _c.c_double, _c.c_void_p, _c.c_char_p

# name -> (restype, argtypes).  One row per symbol of include/lmg.h; tests/test_abi.py
# cross-checks this table against the header.
SIGNATURES = {
    "lmg_version": (_c.c_int, []),
    "lmg_status_string": (_cp, [_c.c_int]),
    "lmg_device_count": (_c.c_int, []),
    "lmg_tune_set": (_c.c_int, [_cp, _c.c_int]),
    "lmg_tune_get": (_c.c_int, [_cp]),
    "lmg_partials_count": (_i64, [_i64]),
    "lmg_csr_residual_norm2": (_c.c_int, [_i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "lmg_csr_jacobi": (_c.c_int, [_i64, _i64, _p, _p, _p, _p, _p, _f64, _p, _p]),
    "lmg_csr_spmv": (_c.c_int, [_i64, _i64, _p, _p, _p, _p, _p, _f64, _f64, _p]),
    "lmg_pcsr_tile_rows": (_c.c_int, []),
    "lmg_pcsr_sweep": (_c.c_int, [_c.c_int, _i64, _i64, _i32, _i32, _p, _p, _p, _p, _c.c_int, _p, _c.c_int, _p, _i32,
                                  _p, _p, _p, _f64, _f64, _p, _p, _p]),
    "lmg_rpat_limits": (_c.c_int, [_p, _p]),
    "lmg_rpat_sweep": (_c.c_int, [_c.c_int, _i64, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _f64, _f64, _p, _p, _p]),
    "lmg_rpat_row_hash": (_c.c_int, [_i64, _p, _p, _p, _p, _p]),
    "lmg_rpat_claim": (_c.c_int, [_i64, _p, _p, _p]),
    "lmg_rpat_verify": (_c.c_int, [_i64, _i64, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p]),
    "lmg_rpat_sweep_grid": (_c.c_int, [_c.c_int, _i64, _p, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _f64, _f64, _p, _p, _p]),
    "lmg_rpat_row_hash_grid": (_c.c_int, [_i64, _p, _p, _p, _p, _p, _p]),
    "lmg_rpat_verify_grid": (_c.c_int, [_i64, _i64, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p]),
    "lmg_stencil_limits": (_c.c_int, [_p]),
    "lmg_stencil_sweep": (_c.c_int, [_c.c_int, _i64, _i32, _p, _i32, _p, _p, _c.c_uint32, _p, _p, _p, _f64, _f64, _p, _p, _p]),
    "lmg_stencil_smooth_supported": (_c.c_int, [_c.c_uint32]),
    "lmg_stencil_smooth": (_c.c_int, [_i64, _i32, _p, _i32, _p, _p, _c.c_uint32, _i32, _p, _c.c_int, _p, _p, _f64, _p, _p, _p]),
    "lmg_stencil_smooth_tiled_supported": (_c.c_int, [_c.c_uint32]),
    "lmg_stencil_smooth_tiled": (_c.c_int, [_i64, _i32, _p, _i32, _p, _p, _c.c_uint32, _i32, _p, _c.c_int, _p, _p, _f64, _p, _p, _p]),
    "lmg_stencil_smooth_tiled_prolong": (_c.c_int, [_i64, _i32, _p, _i32, _p, _p, _c.c_uint32, _i32, _p, _c.c_int, _p, _p, _f64, _p,
                                                    _i64, _i32, _p, _p, _i32, _p, _p, _p]),
    "lmg_stencil_smooth_tiled_restrict": (_c.c_int, [_i64, _i32, _p, _i32, _p, _p, _c.c_uint32, _i32, _p, _c.c_int, _p, _p, _f64, _p,
                                                     _i64, _i32, _p, _p, _i32, _p, _p, _p]),
    "lmg_dia_smooth_supported": (_c.c_int, [_c.c_uint32]),
    "lmg_dia_fill": (_c.c_int, [_i64, _i32, _p, _p, _p, _c.c_uint32, _p, _p, _p, _p]),
    "lmg_dia_smooth": (_c.c_int, [_i64, _i32, _c.c_uint32, _p, _c.c_int, _p, _p, _f64, _p, _p, _p]),
    "lmg_stencil_smooth_prolong_supported": (_c.c_int, [_c.c_uint32]),
    "lmg_stencil_smooth_prolong": (_c.c_int, [_i64, _i32, _p, _i32, _p, _p, _c.c_uint32, _i32, _p, _c.c_int, _p, _p, _f64, _p,
                                              _i64, _i32, _p, _p, _i32, _p, _p, _p, _p, _p]),
    "lmg_stencil_smooth_restrict": (_c.c_int, [_i64, _i32, _p, _i32, _p, _p, _c.c_uint32, _i32, _p, _c.c_int, _p, _p, _f64, _p,
                                               _i64, _i32, _p, _p, _i32, _p, _p, _i32, _p, _p]),
    "lmg_sell_sweep": (_c.c_int, [_c.c_int, _i64, _p, _p, _p, _p, _p, _c.c_int, _p, _i32, _p, _p, _p, _f64, _f64, _p, _p, _p]),
    "lmg_sell_slice_info": (_c.c_int, [_i64, _p, _p, _p, _p, _p, _p]),
    "lmg_sell_fill": (_c.c_int, [_i64, _p, _p, _p, _p, _p, _c.c_int, _p, _p, _p]),
    "lmg_pcsr_tile_colrange": (_c.c_int, [_i64, _i32, _p, _p, _p, _p, _p]),
    "lmg_pcsr_encode_cols16": (_c.c_int, [_i64, _i32, _p, _p, _p, _p, _p]),
    "lmg_value_set_collect": (_c.c_int, [_p, _i64, _p, _i32, _p, _p]),
    "lmg_value_set_insert": (_c.c_int, [_i64, _p, _p, _i64, _i32, _p, _p]),
    "lmg_value_encode": (_c.c_int, [_i64, _p, _p, _i32, _c.c_int, _p, _p, _p]),
    "lmg_csr_inverse_diagonal": (_c.c_int, [_i64, _p, _p, _p, _p, _p]),
    "lmg_csr_transpose_max_row": (_c.c_int, []),
    "lmg_csr_transpose_count": (_c.c_int, [_i64, _i64, _p, _p, _p]),
    "lmg_csr_transpose_fill": (_c.c_int, [_i64, _i64, _p, _p, _p, _p, _p, _p, _p, _p]),
    "lmg_csr_gs_rows": (_c.c_int, [_p, _p, _p, _p, _p, _p, _i64, _p]),
    "lmg_csr_gs_schedule": (_c.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _c.c_int, _p]),
    "lmg_csr_gs_schedule_ell": (_c.c_int, [_i64, _p, _p, _p, _p, _p, _p, _p, _i32, _i64, _p, _i64, _c.c_int, _p]),
    "lmg_stencil_gs_supported": (_c.c_int, [_c.c_uint32]),
    "lmg_stencil_gs_work_bytes": (_i64, [_i64, _i32]),
    "lmg_stencil_gs_sweep": (_c.c_int, [_i64, _i32, _p, _i32, _p, _p, _c.c_uint32, _i32, _p, _p, _p, _p, _c.c_int, _p]),
    "lmg_host_gs_levels": (_i64, [_i64, _p, _p, _p]),
    "lmg_host_greedy_colors": (_i64, [_i64, _p, _p, _p]),
    "lmg_axpby": (_c.c_int, [_i64, _f64, _p, _f64, _p, _p]),
    "lmg_vmul": (_c.c_int, [_i64, _f64, _p, _p, _p, _p]),
    "lmg_copy": (_c.c_int, [_i64, _p, _p, _p]),
    "lmg_zero": (_c.c_int, [_i64, _p, _p]),
    "lmg_dot": (_c.c_int, [_i64, _p, _p, _p, _p, _p]),
    "lmg_gather": (_c.c_int, [_i64, _p, _p, _p, _p]),
    "lmg_scatter": (_c.c_int, [_i64, _p, _p, _p, _p]),
    "lmg_dense_gemv": (_c.c_int, [_i64, _i64, _p, _p, _p, _p]),
    "lmg_dense_gemv_blockdiag": (_c.c_int, [_i64, _i64, _p, _p, _p, _p]),
    "lmg_dense_gemv_windows": (_c.c_int, [_i64, _i64, _i64, _p, _p, _i64, _p, _i64, _f64, _p, _i64, _p]),
    "lmg_dense_gemv_windows_off": (_c.c_int, [_i64, _i64, _i64, _p, _p, _p, _p, _i64, _f64, _p, _i64, _p]),
    "lmg_coarse_front": (_c.c_int, [_i64, _i64, _p, _p, _p, _p, _i64, _p, _p]),
    "lmg_coarse_back": (_c.c_int, [_i64, _i64, _i64, _p, _p, _p, _p, _i64, _f64, _p, _p, _c.c_int, _i64, _p]),
    "lmg_coarse_front_gather": (_c.c_int, [_i64, _i64, _p, _p, _p, _p, _i64, _p, _p, _p]),
    "lmg_coarse_back_gather": (_c.c_int, [_i64, _i64, _i64, _p, _p, _p, _p, _i64, _f64, _p, _p, _c.c_int, _i64, _p, _p]),
    "lmg_batched_gemm": (_c.c_int, [_i64, _i64, _i64, _i64, _f64, _p, _i64, _i64, _p, _i64, _i64, _f64, _p, _i64, _i64, _p]),
    "lmg_copy2d": (_c.c_int, [_i64, _i64, _i64, _f64, _p, _i64, _i64, _p, _i64, _i64, _c.c_int, _p]),
    "lmg_pattern_parity_counts": (_c.c_int, [_i64, _i32, _p, _p, _p]),
    "lmg_block_copy": (_c.c_int, [_i64, _i64, _p, _i64, _p, _i64, _p]),
    "lmg_csr_to_dense": (_c.c_int, [_i64, _i64, _p, _p, _p, _p, _p]),
    "lmg_batched_inverse": (_c.c_int, [_i64, _i32, _p, _p, _p, _p]),
    "lmg_spgemm_count": (_c.c_int, [_i64, _p, _p, _p, _p, _p, _p]),
    "lmg_spgemm_symbolic": (_c.c_int, [_i64, _p, _p, _p, _p, _p, _i32, _p, _p]),
    "lmg_spgemm_numeric": (_c.c_int, [_i64, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p]),
    "lmg_spgemm_numeric_record": (_c.c_int, [_i64, _p, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "lmg_spgemm_numeric_replay": (_c.c_int, [_i64, _p, _p, _p, _p, _p, _p, _i32, _p, _p, _p, _p, _p, _p]),
    "lmg_spgemm_long_rows": (_c.c_int, [_c.c_int, _i64, _p, _p, _p, _p, _p, _p, _p, _i64, _i32, _p, _p, _p, _p, _p, _p, _p]),
    "lmg_scan_scratch_count": (_i64, [_i64]),
    "lmg_exclusive_scan_i32": (_c.c_int, [_i64, _p, _p, _p, _p]),
    "lmg_p1_assemble_2d": (_c.c_int, [_i64, _p, _p, _p, _p, _p, _p, _p, _f64, _p, _p, _p, _p, _p, _p]),
    "lmg_l2_coupling_count": (_c.c_int, [_i64, _i64, _p, _p, _p, _p]),
    "lmg_l2_coupling_fill": (_c.c_int, [_c.c_int, _i64, _i64, _p, _p, _p, _p, _p, _p]),
    "lmg_graph_begin": (_c.c_int, [_p]),
    "lmg_graph_end": (_c.c_int, [_p, _c.POINTER(_p)]),
    "lmg_graph_launch": (_c.c_int, [_p, _p]),
    "lmg_graph_destroy": (_c.c_int, [_p]),
}

_lib = None


class LmgError(RuntimeError):
    pass


def build(force=False, quiet=True):
    """Compile the HIP sources for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL if quiet else None)
    return LIB_PATH


def lib():
    """The loaded library; raises LmgError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LmgError(
                "liblmg_hip.so not found at %s -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.lmg_version() < 100:
            raise LmgError("liblmg_hip.so is older than the Python package")
        _lib = L
    return _lib


def check(status, what=""):
    if status < 0:
        msg = lib().lmg_status_string(int(status)).decode()
        raise LmgError("%s failed: %s (status %d)" % (what or "lmg call", msg, status))
    return status
