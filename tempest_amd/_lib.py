"""ctypes binding of libtempest_hip.so (C ABI: include/tempest_hip.h).

The product path has no CPU fallback: if the shared library is missing or a call fails this
module raises.  Build with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C tempest_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libtempest_hip.so")

c_i64 = C.c_int64
c_u64 = C.c_uint64
c_u32 = C.c_uint32
c_dbl = C.c_double
c_int = C.c_int
ptr = C.c_void_p

# name -> (restype, argtypes); mirrors include/tempest_hip.h one to one
SIGNATURES = {
    "tph_last_error": (C.c_char_p, []),
    "tph_version": (c_int, []),
    "tph_ctx_create": (c_int, [c_int, c_int, c_i64, ptr, C.POINTER(ptr)]),
    "tph_ctx_destroy": (c_int, [ptr]),
    "tph_set_stream": (c_int, [ptr, ptr]),
    "tph_synchronize": (c_int, [ptr]),
    "tph_warmup": (c_int, [ptr]),
    "tph_set_option": (c_int, [ptr, c_int, c_int]),
    "tph_history_append": (c_int, [ptr, ptr, ptr, ptr, c_i64, c_i64, c_dbl, c_dbl, c_i64]),
    "tph_history_size": (c_i64, [ptr]),
    "tph_history_iterations": (c_int, [ptr]),
    "tph_history_clear": (c_int, [ptr]),
    "tph_history_memory": (c_int, [ptr, ptr]),
    "tph_history_read": (c_int, [ptr, c_int, c_i64, c_i64, ptr]),
    "tph_history_ptr": (c_int, [ptr, c_int, C.POINTER(ptr), C.POINTER(c_i64)]),
    "tph_history_load": (c_int, [ptr, ptr, ptr, ptr, c_i64, c_int, ptr, ptr, ptr, ptr]),
    "tph_reweight_partials": (c_int, [ptr, ptr, c_int, ptr]),
    "tph_reweight_eval": (c_int, [ptr, ptr, c_int, ptr]),
    "tph_bench_reweight_time": (c_int, [ptr, c_dbl, c_int, c_int, ptr]),
    "tph_bench_membw_time": (c_int, [ptr, c_int, c_i64, c_int, ptr]),
    "tph_bench_fp64_time": (c_int, [ptr, c_int, ptr]),
    "tph_student_sums": (c_int, [ptr, ptr, ptr, ptr, c_int, c_i64, ptr, c_int, ptr]),
    "tph_student_weights": (c_int, [ptr, ptr, ptr, ptr, c_int, c_i64, c_dbl, ptr]),
    "tph_bench_mf_normals": (c_int, [ptr, c_u64, c_u64, c_u64, c_int, ptr]),
    "tph_bench_mf_counters": (c_int, [ptr, ptr]),
    "tph_weights": (c_int, [ptr, c_dbl, c_dbl, c_dbl, ptr]),
    "tph_logw": (c_int, [ptr, c_dbl, c_i64, ptr]),
    "tph_sum_sq_max": (c_int, [ptr, ptr, c_i64, ptr]),
    "tph_trim_threshold": (c_int, [ptr, ptr, c_i64, c_dbl, c_int, ptr, ptr]),
    "tph_cdf": (c_int, [ptr, ptr, c_i64, ptr, ptr]),
    "tph_resample_systematic": (c_int, [ptr, ptr, c_i64, c_i64, c_i64, c_i64, c_dbl, c_dbl, ptr]),
    "tph_resample_multinomial": (c_int, [ptr, ptr, c_i64, c_i64, c_u64, c_u32, c_u32, c_i64, ptr]),
    "tph_resample_select": (c_int, [ptr, ptr, c_i64, c_i64, c_int, c_u64, c_u32, c_u32, c_dbl, c_dbl, c_dbl, c_dbl, c_int, ptr]),
    "tph_gather": (c_int, [ptr, ptr, c_i64, ptr, ptr, ptr, c_i64]),
    "tph_multinomial_counts": (c_int, [ptr, ptr, c_i64, ptr, c_int, c_i64, c_u64, c_u32, c_u32, ptr]),
    "tph_prior_draw": (c_int, [ptr, ptr, c_i64, c_i64, c_u64, c_u32, c_i64]),
    "tph_inf_repair": (c_int, [ptr, ptr, ptr, ptr, c_i64, c_i64, c_u64, c_u32, c_i64, ptr]),
    "tph_inf_repair_src": (c_int, [ptr, ptr, ptr, ptr, c_i64, c_i64, c_u64, c_u32, c_i64, ptr, ptr]),
    "tph_propose": (c_int, [ptr, c_int, ptr, ptr, c_i64, c_i64, c_int, ptr, ptr, ptr, ptr, ptr, ptr,
                            c_u64, c_u32, c_i64, ptr, ptr, ptr, ptr, ptr]),
    "tph_accept": (c_int, [ptr, c_int, c_dbl, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, c_i64, c_i64,
                           c_int, ptr, c_u64, c_u32, c_i64, ptr, ptr, ptr, ptr]),
    "tph_adapt": (c_int, [ptr, c_int, ptr, ptr, c_int, c_dbl, c_int, c_int, c_int, ptr, ptr, ptr, c_int, ptr, c_i64]),
    "tph_accept_sums_global": (c_int, [ptr, ptr, c_i64, c_int, ptr, c_int]),
    "tph_posterior_rows": (c_int, [ptr, c_int, ptr, c_i64, ptr, c_dbl, ptr, ptr, ptr]),
    "tph_index_compose": (c_int, [ptr, ptr, ptr, c_i64, ptr]),
    "tph_cluster_counts": (c_int, [ptr, ptr, c_i64, c_int, ptr]),
    "tph_fit_modes": (c_int, [ptr, ptr, ptr, c_i64, c_int, ptr, ptr, ptr, ptr, ptr]),
    "tph_chol_inv": (c_int, [ptr, ptr, c_int, ptr, ptr, ptr]),
    "tph_weighted_moments": (c_int, [ptr, ptr, c_i64, ptr]),
    "tph_weighted_sums": (c_int, [ptr, ptr, c_i64, ptr]),
    "tph_weighted_cov_centered": (c_int, [ptr, ptr, c_i64, ptr, ptr]),
    "tph_weighted_moments_shifted": (c_int, [ptr, ptr, c_i64, ptr, ptr]),
    "tph_compact_indices": (c_int, [ptr, ptr, c_i64, ptr, ptr]),
    "tph_gather_u_affine": (c_int, [ptr, ptr, c_i64, ptr, ptr, ptr, ptr, c_i64, ptr]),
    "tph_affine": (c_int, [ptr, ptr, c_i64, c_i64, ptr, ptr]),
    "tph_x_weighted_sums": (c_int, [ptr, ptr, c_i64, c_i64, ptr, ptr, ptr]),
    "tph_x_weighted_cov": (c_int, [ptr, ptr, c_i64, c_i64, ptr, ptr, ptr]),
    "tph_gmm_estep": (c_int, [ptr, ptr, c_i64, c_i64, ptr, ptr, c_int, c_int, ptr, c_int, c_dbl, ptr, ptr, ptr, ptr, ptr]),
    "tph_gmm_em_state_doubles": (c_i64, [c_int, c_int]),
    "tph_gmm_em_begin": (c_int, [ptr, ptr, c_i64, c_i64, c_int, ptr, ptr]),
    "tph_gmm_em_run": (c_int, [ptr, ptr, c_i64, c_i64, ptr, ptr, c_int, c_int, ptr, ptr, c_dbl, c_dbl, c_int, c_int]),
    "tph_cv_sum": (c_int, [ptr, ptr, c_i64, ptr, ptr, ptr]),
    "tph_volume_variation": (c_int, [ptr, ptr, c_i64, ptr, ptr]),
    "tph_comm_attach": (c_int, [ptr, c_int, c_int, ptr, c_i64, ptr, ptr, ptr]),
    "tph_comm_detach": (c_int, [ptr]),
    "tph_comm_p2p_export": (c_int, [ptr, ptr]),
    "tph_comm_p2p_attach": (c_int, [ptr, ptr, ptr]),
    "tph_comm_p2p_active": (c_int, [ptr]),
    "tph_comm_p2p_status": (c_int, [ptr]),
    "tph_comm_stats": (c_int, [ptr, ptr, c_int]),
    "tph_comm_allreduce_dev": (c_int, [ptr, ptr, c_i64, c_int, c_int]),
    "tph_resample_put_global": (c_int, [ptr, ptr, c_i64, c_i64, ptr, ptr, ptr, c_i64]),
    "tph_trim_threshold_global": (c_int, [ptr, ptr, c_i64, c_dbl, c_int, ptr, ptr]),
    "tph_cdf_global": (c_int, [ptr, ptr, c_i64, ptr, ptr, ptr]),
    "tph_resample_select_global": (c_int, [ptr, ptr, c_i64, c_i64, c_int, c_u64, c_u32, c_u32, c_dbl, c_dbl, ptr]),
    "tph_multinomial_counts_global": (c_int, [ptr, ptr, c_i64, ptr, c_int, c_i64, c_u64, c_u32, c_u32, ptr]),
    "tph_fit_modes_global": (c_int, [ptr, ptr, ptr, c_i64, c_int, ptr, ptr, ptr, ptr, ptr]),
}

# host collectives handed to tph_comm_attach (see include/tempest_hip.h)
ALLREDUCE_FN = C.CFUNCTYPE(c_int, ptr, c_i64, c_i64, c_int, c_int)
ALLGATHER_FN = C.CFUNCTYPE(c_int, ptr, c_i64, c_i64, c_i64, c_int)


class TempestHipError(RuntimeError):
    pass


_lib = None


def load(path=None):
    """Load the shared library and attach prototypes.  Raises if it is missing or incomplete."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    # torch bundles its own HIP/HSA runtime; it must be in the process BEFORE our library is loaded so that both
    # resolve to the same libamdhip64 (loading ROCm's copy first leaves the process with two HSA runtimes and
    # "no ROCm-capable device is detected")
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise TempestHipError(
            f"{p} not found: the HIP extension is not built (run `make -C tempest_amd/csrc`). "
            "tempest_amd has no CPU fallback.")
    lib = C.CDLL(p)
    missing = []
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise TempestHipError(f"{p} lacks symbols declared in include/tempest_hip.h: {missing}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().tph_last_error()
        raise TempestHipError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")
