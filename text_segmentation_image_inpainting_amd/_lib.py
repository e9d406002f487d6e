"""ctypes binding of libtsii_hip.so (C ABI: include/tsii_hip.h).

There is no CPU implementation behind this module: if the HIP library is missing or a tensor
is not on a ROCm device, the call fails loudly.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSII_LIBRARY") or os.path.join(_HERE, "libtsii_hip.so")   # TSII_LIBRARY: another BUILD of csrc/ (A/B measurements)
ABI_VERSION = 5           # TSII_ABI_VERSION of include/tsii_hip.h this binding was written against

_p, _i, _l, _f, _z = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t
_GEOM = [_i] * 8  # kh kw sh sw ph pw dh dw

# name -> (restype, argtypes); mirrors include/tsii_hip.h one to one
SIGNATURES = {
    "tsii_version": (_i, []),
    "tsii_last_error": (ctypes.c_char_p, []),
    "tsii_set_gemm_products": (_i, [_i]),
    "tsii_get_gemm_products": (_i, []),
    "tsii_mask_channel_sum": (_i, [_p, _i, _i, _i, _i, _l, _l, _l, _l, _p, _p]),
    "tsii_mask_update": (_i, [_p, _f, _p, _f, _i, _i, _i] + _GEOM + [_i, _i, _f, _i, _p, _p, _p, _p]),
    "tsii_plane_upsample2x": (_i, [_p, _i, _i, _i, _p, _p]),
    "tsii_mul_mask": (_i, [_p, _p, _l, _p, _p]),
    "tsii_pw_ws_bytes": (_z, [_i, _i]),
    "tsii_pw_fwd": (_i, [_p, _l, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _z, _p]),
    "tsii_pw_bwd_dx": (_i, [_p, _l, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p]),
    "tsii_pw_bwd_dw_ws_bytes": (_z, [_l, _i, _i]),
    "tsii_pw_bwd_dw": (_i, [_p, _p, _l, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p, _z, _p]),
    "tsii_dw_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p]),
    "tsii_dw_bwd_dx": (_i, [_p, _p, _p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p]),
    "tsii_dw_bwd_dw_ws_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "tsii_dw_bwd_dw": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p, _z, _p]),
    "tsii_dense_ws_bytes": (_z, [_i, _i, _i, _i]),
    "tsii_dense_fwd": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _z, _p]),
    "tsii_dense_bwd_dx": (_i, [_p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _z, _p]),
    "tsii_head_cat_ok": (_i, [_i, _i, _i, _i, _i, _i]),
    "tsii_head_cat_fwd": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _z, _p]),
    "tsii_head_cat_bwd_dx": (_i, [_p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "tsii_head_cat_bwd_dw": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "tsii_head_cat_low_ok": (_i, [_i, _i, _i, _i, _i, _i]),
    "tsii_head_cat_fwd_low_ok": (_i, [_i, _i, _i, _i, _i, _i]),
    "tsii_head_cat_bwd_low_ok": (_i, [_i, _i, _i, _i, _i, _i]),
    "tsii_head_cat_bwd_low": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "tsii_head_cat_fwd_low": (_i, [_p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "tsii_head_cat_bwd_dw_low": (_i, [_p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "tsii_dense_bwd_dw_ws_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i]),
    "tsii_dense_bwd_dw": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p, _z, _p]),
    "tsii_stem_s2d": (_i, [_p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _p]),
    "tsii_stem_w_fwd": (_i, [_p, _i, _i, _i, _p, _p]),
    "tsii_stem_w_bwd": (_i, [_p, _i, _i, _i, _p, _p]),
    "tsii_bn_ws_bytes": (_z, [_l, _i]),
    "tsii_bn_stats": (_i, [_p, _l, _i, _p, _p, _p, _p, _f, _p, _z, _p]),
    "tsii_bn_act_fwd": (_i, [_p, _l, _i, _p, _p, _p, _p, _f, _i, _f, _p, _p, _p]),
    "tsii_bn_act_bwd": (_i, [_p, _p, _l, _i, _p, _p, _p, _p, _f, _i, _f, _i, _p, _p, _p, _p, _z, _p]),
    "tsii_pw_fwd_up": (_i, [_p, _l, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _z, _p]),
    "tsii_pool2x2_scaled": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "tsii_pw_stat_rows": (_l, [_l]),
    "tsii_pw_fwd_bn": (_i, [_p, _l, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i, _f, _p, _p, _p, _z, _p]),
    "tsii_pw_bwd_dw_bn": (_i, [_p, _p, _l, _i, _i, _p, _p, _p, _i, _p, _p, _p, _i, _f, _p, _p, _p, _z, _p]),
    "tsii_dw_stat_rows": (_l, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "tsii_dw_bwd_stat_rows": (_l, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "tsii_dw_fwd_bn": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _i, _f, _p, _p, _p, _p]),
    "tsii_dw_bwd_dw_bn": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _i, _f, _p, _p, _p, _z, _p]),
    "tsii_dense_stat_rows": (_l, [_i, _i, _i, _i, _i, _i] + _GEOM + [_i, _i]),
    "tsii_dense_fwd_bn": (_i, [_p, _p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p, _z, _p]),
    "tsii_bn_finalize_ws_bytes": (_z, [_l, _i]),
    "tsii_bn_finalize": (_i, [_p, _l, _i, _l, _p, _p, _p, _p, _f, _p, _p, _f, _p, _p, _p, _z, _p]),
    "tsii_dw_bwd_dx_bn": (_i, [_p, _p, _p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p, _p, _p, _f, _i, _f, _p, _p, _p, _p]),
    "tsii_dw_bwd_dxdw_ws_bytes": (_z, [_i, _i, _i, _i] + _GEOM),
    "tsii_dw_bwd_dxdw_bn": (_i, [_p, _p, _p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p, _p, _p, _f, _i, _f, _p, _p, _p, _p, _p, _z, _p]),
    "tsii_dw_bwd_dxdw_fold_ok": (_i, [_i, _i, _i, _i] + _GEOM),
    "tsii_dw_bwd_dxdw_bn2": (_i, [_p, _p, _p, _i, _f, _p, _p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p, _p, _p, _f, _i, _f, _p, _p, _p, _p, _p, _z, _p]),
    "tsii_bn_bwd_reduce_ws_bytes": (_z, [_l, _i]),
    "tsii_bn_bwd_reduce": (_i, [_p, _p, _p, _p, _f, _i, _p, _l, _l, _i, _p, _p, _p, _p, _z, _p]),
    "tsii_bn_bwd_apply": (_i, [_p, _p, _l, _i, _p, _i, _f, _p, _p]),
    "tsii_pw_bwd_dx_bn": (_i, [_p, _l, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _f, _i, _f, _p, _p, _p, _p]),
    "tsii_bn_act_bwd_pre": (_i, [_p, _p, _l, _i, _p, _p, _p, _p, _f, _i, _f, _i, _p, _l, _p, _p, _p, _p, _z, _p]),
    "tsii_bn_act_bwd_pre_pool": (_i, [_p, _p, _l, _i, _p, _p, _p, _p, _f, _i, _f, _i, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "tsii_bn_scale_shift": (_i, [_p, _p, _p, _p, _f, _i, _p, _p, _p]),
    "tsii_act_fwd": (_i, [_p, _l, _i, _f, _p, _p]),
    "tsii_act_bwd": (_i, [_p, _p, _l, _i, _f, _p, _p]),
    "tsii_upcat_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "tsii_upcat_bwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p]),
    "tsii_l1_ws_bytes": (_z, [_l]),
    "tsii_l1_mean_fwd": (_i, [_p, _p, _l, _p, _p, _z, _p]),
    "tsii_l1_mean_bwd": (_i, [_p, _p, _l, _p, _p, _p]),
    "tsii_sgd_nesterov": (_i, [_p, _p, _p, _l, _f, _f, _f, _p]),
    "tsii_add_act_fwd": (_i, [_p, _p, _l, _i, _f, _p, _p]),
    "tsii_copy_channels": (_i, [_p, _l, _i, _i, _p, _i, _i, _p]),
    "tsii_bilinear_up_fwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "tsii_bilinear_up_bwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "tsii_gap_ws_bytes": (_z, [_i, _i, _i]),
    "tsii_gap_fwd": (_i, [_p, _i, _i, _i, _p, _p, _z, _p]),
    "tsii_gap_bwd": (_i, [_p, _i, _i, _i, _p, _p]),
    "tsii_scse_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    "tsii_scse_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "tsii_bce_focal_fwd": (_i, [_p, _p, _l, _f, _f, _f, _p, _p, _z, _p]),
    "tsii_bce_focal_bwd": (_i, [_p, _p, _l, _f, _f, _f, _p, _p, _p]),
    "tsii_pixel_shuffle": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "tsii_compose_fwd": (_i, [_p, _p, _p, _l, _p, _p]),
    "tsii_compose_bwd": (_i, [_p, _p, _l, _p, _p]),
    "tsii_masked_l1_fwd": (_i, [_p, _p, _p, _l, _f, _f, _p, _p, _z, _p]),
    "tsii_masked_l1_bwd": (_i, [_p, _p, _p, _l, _f, _f, _p, _p, _p]),
    "tsii_tv_fwd": (_i, [_p, _i, _i, _i, _i, _p, _p, _z, _p]),
    "tsii_tv_bwd": (_i, [_p, _i, _i, _i, _i, _p, _p, _p]),
    # ---- bf16 activation storage (round 5) ----
    "tsii_bf16_stat_rows": (_l, [_l]),
    "tsii_bf16_pw_ws_bytes": (_z, [_i, _i]),
    "tsii_bf16_pw_fwd": (_i, [_p, _l, _i, _p, _i, _p, _p, _p, _i, _f, _p, _p, _p, _z, _p]),
    "tsii_bf16_pw_bwd_dx": (_i, [_p, _l, _i, _p, _i, _p, _p, _p, _p, _p, _f, _i, _f, _p, _p, _p, _z, _p]),
    "tsii_bf16_pw_bwd_dw_ws_bytes": (_z, [_l, _i, _i]),
    "tsii_bf16_pw_bwd_dw": (_i, [_p, _p, _l, _i, _i, _p, _p, _i, _f, _p, _p, _p, _z, _p]),
    "tsii_bf16_dense_ws_bytes": (_z, [_i, _i, _i, _i]),
    "tsii_bf16_dense_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p, _z, _p]),
    "tsii_bf16_dense_bwd_dx": (_i, [_p, _p, _i, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _z, _p]),
    "tsii_bf16_dense_bwd_dw_ws_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i]),
    "tsii_bf16_dense_bwd_dw": (_i, [_p, _p, _i, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p, _z, _p]),
    "tsii_bf16_dw_stat_rows": (_l, [_i] * 10),
    "tsii_bf16_dw_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _i, _f, _p, _p, _p]),
    "tsii_bf16_dw_bwd_stat_rows": (_l, [_i] * 12),
    "tsii_bf16_dw_bwd_dx": (_i, [_p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _p, _p, _p, _f, _i, _f, _p, _p, _p]),
    "tsii_bf16_dw_bwd_dw_ws_bytes": (_z, [_i] * 10),
    "tsii_bf16_dw_bwd_dw": (_i, [_p, _p, _i, _i, _i, _i] + _GEOM + [_i, _i, _p, _p, _i, _f, _p, _p, _p, _z, _p]),
    "tsii_bf16_avgpool": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "tsii_bf16_bn_stat_rows": (_l, [_l, _i]),
    "tsii_bf16_bn_stats": (_i, [_p, _l, _i, _p, _p]),
    "tsii_bf16_bn_act_fwd": (_i, [_p, _l, _i, _p, _p, _i, _f, _p, _p, _p]),
    "tsii_bf16_bn_ws_bytes": (_z, [_l, _i]),
    "tsii_bf16_bn_act_bwd": (_i, [_p, _p, _l, _i, _p, _p, _p, _p, _f, _i, _f, _i, _p, _l, _p, _p, _p, _p, _z, _p]),
    "tsii_bf16_add_act_fwd": (_i, [_p, _p, _l, _i, _f, _p, _p]),
    "tsii_bf16_act_bwd": (_i, [_p, _p, _l, _i, _f, _p, _p]),
    "tsii_bf16_copy_channels": (_i, [_p, _l, _i, _i, _p, _i, _i, _p]),
    "tsii_bf16_bilinear_up_fwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "tsii_bf16_bilinear_up_bwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "tsii_bf16_stem_s2d": (_i, [_p, _i, _i, _i, _i, _i, _p, _p]),
    "tsii_bf16_from_f32": (_i, [_p, _l, _p, _p]),
    "tsii_bf16_to_f32": (_i, [_p, _l, _p, _p]),
    "tsii_bf16_channel_to_f32": (_i, [_p, _l, _i, _i, _p, _p]),
    "tsii_bf16_channel_from_f32": (_i, [_p, _l, _i, _i, _p, _p]),
}

_LIB = None


def bind(cdll):
    """Attach the header's signatures to a loaded library; fails if a symbol is missing."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)  # AttributeError -> missing export
        fn.restype = res
        fn.argtypes = args
    return cdll


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the MI355X HIP library has not been built "
                "(python -m text_segmentation_image_inpainting_amd.build_ext). There is no CPU fallback.")
        cdll = ctypes.CDLL(LIB_PATH)
        # version first: a stale build lacks the newer symbols, and "rebuild" is a better message than a missing-symbol error
        try:
            cdll.tsii_version.restype = ctypes.c_int
            have = int(cdll.tsii_version())
        except AttributeError:
            have = -1
        if have != ABI_VERSION:
            raise RuntimeError("libtsii_hip.so ABI version mismatch: library %d, binding %d -- rebuild with "
                               "python -m text_segmentation_image_inpainting_amd.build_ext" % (have, ABI_VERSION))
        _LIB = bind(cdll)
    return _LIB


def stream():
    """hipStream_t of torch's current stream (the library enqueues on it, never syncs)."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check_dtype(*tensors, bf16_ok=False):
    """fp32 -- or bf16 for the callers that dispatch to a ``tsii_bf16_*`` kernel themselves (``bf16_ok=True``).  Everything else
    raises: an fp32 kernel handed a bf16 buffer would read / write twice the bytes the buffer holds."""
    for t in tensors:
        if t is None or t.dtype == torch.float32 or (bf16_ok and t.dtype == torch.bfloat16):
            continue
        if t.dtype == torch.bfloat16:
            raise NotImplementedError("text_segmentation_image_inpainting_amd: this op has no bf16 kernel (bf16 activation storage covers the "
                                      "mask-free convolution / BatchNorm / element-wise layers of the Xception segmentation net only); "
                                      "got a bf16 tensor of shape %s" % (tuple(t.shape),))
        raise RuntimeError(f"text_segmentation_image_inpainting_amd: fp32 tensors (or bf16 activation storage, ops.set_activation_storage), got {t.dtype}")


def check_device(*tensors, bf16_ok=False):
    """Every given tensor (None entries are skipped) lives on a ROCm GPU and has a type the calling op has a kernel for."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("text_segmentation_image_inpainting_amd: tensors must live on a ROCm GPU "
                               "(MI355X); this implementation has no CPU path")
    check_dtype(*tensors, bf16_ok=bf16_ok)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# Optional per-entry-point timing with HIP events on the launch stream (bench.py's roofline leg):
# _TIMED maps an entry-point name to a list of (start_event, end_event, scalar_args).
_TIMED = None


def start_timing(names):
    """Record a HIP event pair (on torch's current stream, where the kernels are enqueued) around
    every call of the given entry points until ``stop_timing()``."""
    global _TIMED
    _TIMED = {n: [] for n in names}


def stop_timing():
    """-> {name: [(milliseconds, scalar_args), ...]} ; synchronises the device."""
    global _TIMED
    rec, _TIMED = _TIMED, None
    torch.cuda.synchronize()
    return {n: [(a.elapsed_time(b), args) for a, b, args in lst] for n, lst in (rec or {}).items()}


# Optional census of the entry points called (bench.py: which kernel family -- fp32 or tsii_bf16_* -- a network's step really runs)
_COUNTS = None


def count_calls(on):
    """count_calls(True) starts counting calls per entry point; count_calls(False) stops and returns {name: calls}."""
    global _COUNTS
    rec, _COUNTS = _COUNTS, ({} if on else None)
    return rec or {}


# Arithmetic mode of the matrix products (include/tsii_hip.h: tsii_set_gemm_products).  The library's switch is
# thread-local and autograd runs backward on its own threads, so the mode is kept HERE and handed to the library on
# whichever thread makes a call.  None = the library default (6, or TSII_GEMM_PRODUCTS).
_GEMM_PRODUCTS = None
_GEMM_TOUCHED = False      # set_gemm_products() was called at least once: from then on every call re-applies the selection


def set_gemm_products(products):
    """0 (f32 MFMA), 1 (bf16 operands), 3, 6 (default, fp32 class) or 8; None = library default.  Applies to every
    later call made through this module, on any thread."""
    global _GEMM_PRODUCTS
    if products is not None and products not in (0, 1, 3, 6, 8):
        raise ValueError("gemm products: 0, 1, 3, 6 or 8")
    global _GEMM_TOUCHED
    # _GEMM_TOUCHED stays set for the rest of the process, also after set_gemm_products(None): the library's switch is per THREAD
    # and autograd's worker threads may still hold an earlier selection -- only re-applying "-1" (the process default) on every
    # call brings them back.  (Round 4 tried clearing the flag on None, as a review suggested: every backward pass after a
    # bf16-operand test then ran its GEMMs with bf16 operands on the worker threads -- 11 GPU tests off by 3e-3.)  Code that sets
    # the library's switch directly must therefore call the raw entry points, not call().
    _GEMM_PRODUCTS, _GEMM_TOUCHED = products, True
    lib().tsii_set_gemm_products(-1 if products is None else int(products))


def get_gemm_products():
    if _GEMM_TOUCHED:
        lib().tsii_set_gemm_products(-1 if _GEMM_PRODUCTS is None else int(_GEMM_PRODUCTS))
    return int(lib().tsii_get_gemm_products())


def call(name, *args):
    L = lib()
    if _GEMM_TOUCHED:       # the library's switch is per thread: bring the calling thread (autograd's workers too) in line
        L.tsii_set_gemm_products(-1 if _GEMM_PRODUCTS is None else _GEMM_PRODUCTS)
    if _COUNTS is not None:
        _COUNTS[name] = _COUNTS.get(name, 0) + 1
    timed = _TIMED is not None and name in _TIMED
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = getattr(L, name)(*args)
    if timed:
        e1.record()
        _TIMED[name].append((e0, e1, tuple(a for a in args if isinstance(a, (int, float)))))
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {L.tsii_last_error().decode(errors='replace')}")
