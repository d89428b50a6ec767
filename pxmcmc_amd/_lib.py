"""
ctypes binding of the C-ABI library (include/pxmcmc_amd.h).  No torch types cross the
boundary: tensors are passed as raw device pointers + sizes, the stream as a void*.

The HIP library is the only compute path: if it is missing or fails to load this module
raises, it never falls back to a CPU implementation.
"""
import ctypes as C
import os

# torch first: it ships its own HIP runtime (torch/lib/libamdhip64.so).  If this library (linked against
# /opt/rocm's libamdhip64.so.7) were loaded before torch, the process would end up with two HIP runtimes and
# the second one sees no device.  With torch loaded first the loader resolves our dependency to the runtime
# torch already brought in, so kernels, streams and device pointers are shared.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PXM_LIB_PATH") or os.path.join(_HERE, "lib", "libpxmcmc_amd.so")  # override: A/B builds

c_i64, c_u64, c_int, c_dbl, c_vp = C.c_int64, C.c_uint64, C.c_int, C.c_double, C.c_void_p

# name -> (restype, argtypes): every symbol include/pxmcmc_amd.h declares
SIGNATURES = {
    "pxm_version": (c_int, []),
    "pxm_noise_bits": (c_int, []),
    "pxm_host_check_address_ranges": (c_i64, [c_int, c_dbl, c_int, c_int, c_int, c_int]),
    "pxm_last_error": (C.c_char_p, []),
    "pxm_device_count": (c_int, []),
    "pxm_capture_begin": (c_int, []),
    "pxm_capture_end": (c_int, []),
    "pxm_deferred_pending": (c_int, []),
    "pxm_tables_trim": (c_int, []),
    "pxm_wav_set_iter_counter": (c_int, [c_vp, c_vp]),
    "pxm_wav_release_iter_counter": (c_int, [c_vp, c_vp]),
    "pxm_wav_iter_counter_add": (c_int, [c_vp, c_u64, c_vp]),
    "pxm_wav_flow_status": (c_int, [c_vp, c_vp]),
    "pxm_wav_flow_enabled": (c_int, [c_vp]),
    "pxm_wav_exact_dft_scales": (c_int, [c_vp]),
    "pxm_wav_status": (c_int, [c_vp, c_int, c_vp]),
    "pxm_sht_status": (c_int, [c_vp, c_int, c_vp]),
    "pxm_wav_profile_enable": (c_int, [c_vp, c_int]),
    "pxm_wav_profile_read": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pxm_wav_profile_read_launches": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "pxm_wav_profile_read_dft": (c_int, [c_vp, c_vp, c_vp, c_vp]),
    "pxm_wav_workspace_nonfinite": (c_i64, [c_vp, c_vp]),
    "pxm_reduce_scratch_doubles": (c_i64, [c_int]),
    "pxm_j_max": (c_int, [c_int, c_dbl]),
    "pxm_wav_bandlimits": (c_int, [c_int, c_dbl, c_int, c_vp, c_int]),
    "pxm_wav_ncoefs": (c_i64, [c_int, c_dbl, c_int, c_vp]),
    "pxm_tiling_axisym": (c_int, [c_int, c_dbl, c_int, c_vp, c_vp]),
    "pxm_mw_ring_weights": (c_int, [c_int, c_vp]),
    "pxm_host_sht_tables": (c_int, [c_int, c_int, c_int, c_vp, c_vp]),
    "pxm_host_rec_table": (c_int, [c_int, c_int, c_int, c_vp]),
    "pxm_host_pfa511_tables": (c_int, [c_vp, c_vp]),
    "pxm_quantile_range": (c_int, [c_vp, c_i64, c_i64, c_i64, c_dbl, c_vp, c_vp]),
    "pxm_sht_uses_recursion": (c_int, [c_vp]),
    "pxm_rec_reduce_selftest": (c_int, [c_vp]),
    "pxm_sht_plan_create": (c_int, [c_int, c_int, c_int, C.c_uint, C.POINTER(c_vp)]),
    "pxm_sht_plan_destroy": (c_int, [c_vp]),
    "pxm_sht_inverse": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "pxm_sht_forward": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "pxm_sht_inverse_adjoint": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "pxm_sht_forward_adjoint": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "pxm_sht_table_bytes": (c_i64, [c_vp, c_int]),
    "pxm_wav_plan_create": (c_int, [c_int, c_dbl, c_int, c_int, C.c_uint, C.POINTER(c_vp)]),
    "pxm_wav_plan_destroy": (c_int, [c_vp]),
    "pxm_wav_synthesis": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "pxm_wav_synthesis_adjoint": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "pxm_wav_analysis": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "pxm_wav_analysis_adjoint": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "pxm_wav_table_bytes": (c_i64, [c_vp, c_int]),
    "pxm_wav_gradg_step": (
        c_int,
        [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_dbl, c_dbl, c_dbl, c_vp, c_int, c_u64, c_u64, c_u64, c_vp, c_int, c_vp],
    ),
    "pxm_wav_image_init": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "pxm_wav_image_step": (
        c_int,
        [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_dbl, c_dbl, c_dbl, c_vp, c_int, c_u64, c_u64, c_u64, c_vp, c_vp, c_int, c_vp],
    ),
    "pxm_wav_ring_set_data": (c_int, [c_vp, c_vp, c_vp]),
    "pxm_wav_ring_init": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "pxm_wav_ring_step": (
        c_int,
        [c_vp, c_vp, c_dbl, c_dbl, c_vp, c_dbl, c_dbl, c_dbl, c_vp, c_int, c_u64, c_u64, c_u64, c_vp, c_int, c_vp],
    ),
    "pxm_wav_ring_preds": (c_int, [c_vp, c_vp, c_int, c_vp]),
    "pxm_wav_wl_attach": (c_int, [c_vp, c_vp, c_vp, c_i64]),
    "pxm_wav_wl_uses_recursion": (c_int, [c_vp]),
    "pxm_wav_wl_forward": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp]),
    "pxm_wav_wl_adjoint": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_vp]),
    "pxm_soft": (c_int, [c_vp, c_vp, c_dbl, c_vp, c_i64, c_int, c_int, c_vp]),
    "pxm_residual_grad": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_int, c_int, c_vp]),
    "pxm_myula_step": (
        c_int,
        [c_vp, c_vp, c_vp, c_dbl, c_vp, c_dbl, c_dbl, c_vp, c_int, c_u64, c_u64, c_u64, c_vp, c_i64, c_int, c_int, c_vp],
    ),
    "pxm_chain_step": (
        c_int,
        [c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_vp, c_int, c_u64, c_u64, c_u64, c_vp, c_i64, c_int, c_int, c_vp],
    ),
    "pxm_myula_step_it": (
        c_int,
        [c_vp, c_vp, c_vp, c_dbl, c_vp, c_dbl, c_dbl, c_vp, c_int, c_u64, c_u64, c_u64, c_vp, c_vp, c_i64, c_int, c_int, c_vp],
    ),
    "pxm_chain_step_it": (
        c_int,
        [c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_vp, c_int, c_u64, c_u64, c_u64, c_vp, c_vp, c_i64, c_int, c_int, c_vp],
    ),
    "pxm_randn": (c_int, [c_vp, c_i64, c_int, c_int, c_u64, c_u64, c_u64, c_vp]),
    "pxm_box_muller": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_vp]),
    "pxm_reduce_l1": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "pxm_reduce_l2": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "pxm_reduce_vdot": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "pxm_logtransition": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "pxm_pxmala_accept": (c_int, [c_vp, c_vp, c_u64, c_u64, c_u64, c_vp, c_vp, c_int, c_dbl, c_i64, c_int, c_vp]),
    "pxm_pxmala_propose": (
        c_int,
        [c_vp, c_vp, c_vp, c_vp, c_dbl, c_vp, c_vp, c_dbl, c_vp, c_int, c_u64, c_u64, c_u64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
         c_i64, c_int, c_int, c_vp],
    ),
    "pxm_pxmala_accept2": (
        c_int,
        [c_vp, c_vp, c_vp, c_vp, c_dbl, c_vp, c_vp, c_vp, c_vp, c_u64, c_u64, c_u64, c_vp, c_vp, c_vp, c_int, c_dbl, c_vp, c_vp,
         c_int, c_int, c_vp],
    ),
    "pxm_pxmala_finish": (
        c_int,
        [c_vp, c_vp, c_vp, c_vp, c_dbl, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_vp, c_dbl, c_dbl, c_vp, c_vp,
         c_vp, c_vp, c_u64, c_u64, c_u64, c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int,
         c_vp],
    ),
    "pxm_select_copy_many": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "pxm_counter_add": (c_int, [c_vp, c_u64, c_vp]),
    "pxm_select_copy": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "pxm_wl_harmonic_mapping": (c_int, [c_vp, c_vp, c_vp, c_i64, c_int, c_vp]),
    "pxm_wl_mask_gather": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_vp]),
    "pxm_wl_mask_scatter": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_int, c_vp]),
    "pxm_csr_matvec": (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_int, c_int, c_vp]),
    "pxm_csr_matvec_batched": (c_int, [c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
}


class PxmError(RuntimeError):
    pass


NOISE_F64 = 16  # PXM_NOISE_F64: OR-ed into the mode / noise_complex / dtype argument of the noise-drawing entry points
STATUS_FLOW_WAIT, STATUS_PAIR_SYNC = 1, 2  # bits of pxm_wav_status / pxm_sht_status


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"pxmcmc_amd: HIP extension not built ({LIB_PATH} missing). "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C pxmcmc_amd/csrc`. "
            "There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def check(rc):
    """Raise if a C-ABI call returned an error code."""
    if rc is not None and rc < 0:
        raise PxmError(lib.pxm_last_error().decode())
    return rc


def require_gpu():
    if lib.pxm_device_count() < 1:
        raise PxmError("pxmcmc_amd: no HIP device visible; the HIP kernels are the only compute path")
