"""
ctypes binding of libdensematch.so (include/densematch.h).

The product path has NO fallback: if the library is missing or a call fails,
an exception is raised.  (The NumPy oracle under oracle/ is test
infrastructure and is never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdensematch.so")

DM_OK, DM_EINVAL, DM_ENOMEM, DM_EHIP, DM_ESINGULAR = 0, -1, -2, -3, -4
DM_F16, DM_F32, DM_PROJECT_F64 = 0, 1, 0x10

_p = C.c_void_p
_i = C.c_int
_d = C.c_double

# name -> (restype, argtypes); exactly the symbols include/densematch.h declares
SIGNATURES = {
    "dm_create": (_i, [_i, _p, C.POINTER(_p)]),
    "dm_destroy": (_i, [_p]),
    "dm_last_error": (C.c_char_p, [_p]),
    "dm_version": (C.c_char_p, []),
    "dm_workspace_bytes": (C.c_size_t, [_p]),
    "dm_set_option": (_i, [_p, C.c_char_p, _i]),
    "dm_last_requeued_rows": (_i, [_p, C.POINTER(_i)]),
    "dm_profile_kernel": (_i, [_p, C.c_char_p]),
    "dm_profile_read": (_i, [_p, C.POINTER(_i), C.POINTER(_d)]),
    "dm_profile_report": (_i, [_p, C.c_char_p, C.c_size_t]),
    "dm_measure_peak": (_i, [_p, _i, C.POINTER(_d)]),
    "dm_simnn_f16": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "dm_project": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p]),
    "dm_fmap_c00": (_i, [_p, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p]),
    "dm_fmap_solve": (_i, [_p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _d, _d, _p, _p]),
    "dm_fmap_fit": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _d, _d, _p, _p]),
    "dm_fmap_energy_grad": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "dm_lbfgs_state_bytes": (C.c_size_t, [_i, _i, _i]),
    "dm_lbfgs_init": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "dm_lbfgs_advance": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _d, _d, _i, _i, _i]),
    "dm_lbfgs_result": (_i, [_p, _i, _i, _i, _p, _p, _p, _p]),
    "dm_fmap_fit_steps": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p,
                               _d, _d, _i, _i, _i]),
    "dm_fmap_fit_fused_ok": (_i, [_i, _i, _p, _i]),
    "dm_fmap_fit_fused": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _d, _d, _i, _i, _i, _p, _p, _p, _p, _p]),
    "dm_fmap_descr_ops": (_i, [_p, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p]),
    "dm_fm_to_p2p": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "dm_fm_to_p2p_uses_split": (_i, [_p, _i, _i, _i]),
    "dm_knn_query_f64": (_i, [_p, _i, _i, _i, _i, _p, _p, _p]),
    "dm_knn_query_topk_f64": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "dm_mapped_indicator": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p]),
    "dm_p2p_to_fm": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _p]),
    "dm_eigenbasis": (_i, [_p, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "dm_tufted_cover": (_i, [_i, _i, _p, _p, _d, _p, _p, _p, _p]),
    "dm_tufted_cover_batch": (_i, [_i, _p, _p, _p, _p, _d, _p, _p, _p, _p, _i]),
    "dm_laplacian_rows_bytes": (C.c_size_t, [_i, _i, _i]),
    "dm_laplacian_rows": (_i, [_p, _i, _i, _i, _p, _p, _p, _d, _p, _p, C.POINTER(_i)]),
    "dm_laplacian_ell": (_i, [_p, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "dm_precise_map": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p]),
    "dm_linear_sum_assignment": (_i, [_p, _i, _i, _i, _p, _i, _p, _p]),
    "dm_lsa_indicator_ok": (_i, [_p, _i, _i, _i, _i]),
    "dm_lsa_indicator": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _p, _i, _p, _p]),
    "dm_p2p_to_fm_lstsq": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _p]),
    "dm_icp": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p]),
    "dm_zoomout": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p]),
}

# the float64-basis forms (const double* Phi / mass): same argument lists
for _n in ("dm_project", "dm_fmap_fit", "dm_fmap_c00", "dm_fm_to_p2p", "dm_mapped_indicator", "dm_p2p_to_fm", "dm_precise_map", "dm_p2p_to_fm_lstsq", "dm_icp", "dm_lsa_indicator",
           "dm_zoomout"):
    SIGNATURES[_n + "_f64"] = SIGNATURES[_n]

_libs = {}


def load(path=None):
    """Load libdensematch.so (built in-tree by `python -m densematcher_amd._build`).  `path`: another build of the same
    ABI (tools/ load the -DDM_EXPERIMENTS build this way); nothing in the package passes it."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP library has not been built. "
            "Run `python -m densematcher_amd._build` (needs hipcc). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64; import it first so that the
    # library binds to the runtime that owns the tensors' device memory and streams.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib


class DenseMatchError(RuntimeError):
    pass


def raise_for(rc, lib, ctx):
    if rc == DM_OK:
        return
    msg = lib.dm_last_error(ctx)
    msg = msg.decode() if msg else ""
    if rc == DM_EINVAL:
        raise ValueError(msg or "invalid argument")
    if rc == DM_ENOMEM:
        raise MemoryError(msg or "device workspace allocation failed")
    if rc == DM_ESINGULAR:
        raise DenseMatchError(msg or "system not positive definite")
    raise DenseMatchError(msg or f"HIP failure (status {rc})")
