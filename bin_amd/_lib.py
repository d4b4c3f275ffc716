"""ctypes binding of libbinhip.so (include/binhip.h).  The product path has NO fallback: if the
library is missing or a call fails this raises, it never routes to PyTorch/CPU code."""
import ctypes as C
import os

from .build import LIB_PATH

RDN_LAYERS = 66              # bin_stage4's layer count; BinRdnPlan arrays hold RDN_MAX_LAYERS
RDN_MAX_LAYERS, RDN_MAX_CONVS = 192, 7
PLAN_KEEP_ACTS, PLAN_NO_FUSE, PLAN_RDB3, PLAN_FUSED_UPNET, PLAN_FUSED_UPNET_TRAIN = 1, 2, 4, 8, 16
BWD_ACCUMULATE = 1          # BinRdnBwdPlan.reserved flag (BINHIP_BWD_ACCUMULATE)
BWD_SAVED_X3 = 2            # BINHIP_BWD_SAVED_X3
BWD_FUSED_UPNET = 4         # BINHIP_BWD_FUSED_UPNET
EPI_PLANES, EPI_SHUFFLE, EPI_FINAL, EPI_FINAL_SUBPIX = 0, 1, 2, 4
PROF_WGRAD = 16             # BINHIP_PROF_WGRAD
LOSS_CHARBONNIER, LOSS_L1_SUM, LOSS_L2_SUM = 0, 1, 2
RDN_LAYOUT_WORDS, RDN_BWD_LAYOUT_WORDS = 16, 24


class BinConvDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("ksize", C.c_int32),
                ("cin_chunks", C.c_int32), ("cout", C.c_int32), ("cout_pad", C.c_int32),
                ("nterms", C.c_int32), ("epilogue", C.c_int32), ("relu", C.c_int32),
                ("x_cpg", C.c_int32), ("x_group_stride", C.c_int64), ("n_images", C.c_int32),
                ("reserved", C.c_int32), ("status", C.c_void_p)]


LOSS_MAX_TERMS = 24          # BINHIP_LOSS_MAX_TERMS
CONV_HALF_LAST_CHUNK = 1     # BinConvDesc.reserved flag of binhip_conv2d_fwd (BINHIP_CONV_HALF_LAST_CHUNK)


class BinLossTerms(C.Structure):
    _fields_ = [("x", C.c_void_p * LOSS_MAX_TERMS), ("y", C.c_void_p * LOSS_MAX_TERMS), ("n_terms", C.c_int32)]


class BinLossGrads(C.Structure):
    _fields_ = [("out", C.c_void_p * LOSS_MAX_TERMS), ("term_a", C.c_int32 * LOSS_MAX_TERMS),
                ("term_b", C.c_int32 * LOSS_MAX_TERMS), ("sign_a", C.c_float * LOSS_MAX_TERMS),
                ("sign_b", C.c_float * LOSS_MAX_TERMS), ("n_out", C.c_int32)]


class BinRdnShape(C.Structure):
    """(G0, D, C, G) of an RDN sub-network (include/binhip.h); all zero = bin_stage4's (96, 12, 4, 32)."""
    _fields_ = [("G0", C.c_int32), ("D", C.c_int32), ("C", C.c_int32), ("G", C.c_int32)]


class BinRdnPlan(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("n_inputs", C.c_int32),
                ("nterms", C.c_int32), ("reserved", C.c_int32), ("shape", BinRdnShape),
                ("w_hi", C.c_void_p * RDN_MAX_LAYERS), ("w_lo", C.c_void_p * RDN_MAX_LAYERS),
                ("bias", C.c_void_p * RDN_MAX_LAYERS), ("status", C.c_void_p), ("profiler", C.c_void_p)]


class BinRdnBwdPlan(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("n_inputs", C.c_int32),
                ("nterms", C.c_int32), ("reserved", C.c_int32), ("shape", BinRdnShape),
                ("wt_hi", C.c_void_p * RDN_MAX_LAYERS), ("wt_lo", C.c_void_p * RDN_MAX_LAYERS),
                ("zero_bias", C.c_void_p),
                ("dw", C.c_void_p * RDN_MAX_LAYERS), ("db", C.c_void_p * RDN_MAX_LAYERS), ("gin", C.c_void_p * 5),
                ("status", C.c_void_p), ("aux_stream", C.c_void_p), ("profiler", C.c_void_p)]


class BinRelayoutItem(C.Structure):
    _fields_ = [("w", C.c_void_p * (RDN_MAX_CONVS + 1)), ("bias", C.c_void_p), ("w_hi", C.c_void_p), ("w_lo", C.c_void_p),
                ("bias_out", C.c_void_p), ("kind", C.c_int32), ("cout", C.c_int32), ("cin", C.c_int32),
                ("ksize", C.c_int32), ("rows_pad", C.c_int32), ("cin_chunks", C.c_int32), ("cout_block", C.c_int32),
                ("shuffle_or_group", C.c_int32), ("shape", BinRdnShape)]


RELAYOUT_FWD, RELAYOUT_DGRAD, RELAYOUT_RDB_GATHER = 0, 1, 2

_SIGNATURES = {
    "binhip_version": (C.c_int, []),
    "binhip_device_cus": (C.c_int, []),
    "binhip_conv_cout_block": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "binhip_weights_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "binhip_weights_relayout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "binhip_weights_relayout_batch": (C.c_int, [C.POINTER(BinRelayoutItem), C.c_int, C.c_void_p]),
    "binhip_conv2d_fwd": (C.c_int, [C.POINTER(BinConvDesc)] + [C.c_void_p] * 10 +
                          [C.POINTER(C.c_void_p), C.c_void_p]),
    "binhip_nchw_to_planes": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "binhip_planes_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p]),
    "binhip_pixel_unshuffle_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p]),
    "binhip_pack_inputs": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p]),
    "binhip_u8_to_frame": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "binhip_frame_to_u8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                     C.c_void_p]),
    "binhip_convlstm_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                                          C.c_void_p, C.c_void_p]),
    "binhip_lstm_gates_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "binhip_lstm_gates_bwd": (C.c_int, [C.c_void_p] * 4 + [C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "binhip_charbonnier_partials": (C.c_int, [C.c_int64]),
    "binhip_charbonnier_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "binhip_charbonnier_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "binhip_pixel_loss_fwd": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "binhip_pixel_loss_bwd": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "binhip_multi_loss_fwd": (C.c_int, [C.c_int, C.POINTER(BinLossTerms), C.c_int64, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "binhip_multi_loss_bwd": (C.c_int, [C.c_int, C.POINTER(BinLossTerms), C.c_int64, C.c_float, C.c_void_p,
                                        C.POINTER(BinLossGrads), C.c_void_p]),
    "binhip_dgrad_rows_pad": (C.c_int, [C.c_int, C.c_int]),
    "binhip_weights_relayout_dgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "binhip_weights_relayout_rdb_gather": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p]),
    "binhip_conv2d_bwd_data": (C.c_int, [C.POINTER(BinConvDesc)] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "binhip_wgrad_workspace_bytes": (C.c_size_t, [C.c_int] * 6),
    "binhip_conv2d_bwd_weight": (C.c_int, [C.POINTER(BinConvDesc)] + [C.c_void_p] * 6 + [C.c_size_t, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "binhip_grad_scale": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "binhip_nchw_to_planes_scaled": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "binhip_unshuffle_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "binhip_unpack_input_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.POINTER(C.c_void_p), C.c_void_p]),
    "binhip_convlstm_bwd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "binhip_convlstm_bwd": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t] + [C.c_void_p] * 6),
    "binhip_rdn_backward_workspace_bytes": (C.c_size_t, [C.c_int] * 5 + [C.POINTER(BinRdnShape)]),
    "binhip_rdn_backward": (C.c_int, [C.POINTER(BinRdnBwdPlan), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    "binhip_profiler_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "binhip_profiler_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "binhip_profiler_destroy": (None, [C.c_void_p]),
    "binhip_rdb_tail_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 10 +
                            [C.c_int, C.c_void_p, C.c_void_p]),
    "binhip_rdn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(BinRdnShape)]),
    "binhip_rdn_workspace_layout": (C.c_int, [C.c_int] * 5 + [C.POINTER(BinRdnShape), C.POINTER(C.c_int64), C.c_int]),
    "binhip_rdn_backward_workspace_layout": (C.c_int, [C.c_int] * 5 + [C.POINTER(BinRdnShape), C.POINTER(C.c_int64), C.c_int]),
    "binhip_rdn_forward": (C.c_int, [C.POINTER(BinRdnPlan), C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
}

# Entry points that exist only in BINHIP_TUNING side builds (tools/: variant sweeps, ablations) — never in the product
# library, and not declared in include/binhip.h.  Bound when present.
_TUNING_SIGNATURES = {
    "binhip_set_variant": (C.c_int, [C.c_int, C.c_int]),
    "binhip_set_tail_depth": (C.c_int, [C.c_int]),
    "binhip_wgrad_set_debug": (C.c_int, [C.c_int]),
}
STATUS_SATURATED = 1            # BINHIP_STATUS_SATURATED
STATUS_SYNC_TIMEOUT = 2         # BINHIP_STATUS_SYNC_TIMEOUT

_lib = None


def exported_symbols():
    """Names every include/binhip.h entry point must resolve to (checked by the CPU test-suite)."""
    return sorted(_SIGNATURES)


def lib():
    """Load libbinhip.so (once).  Raises RuntimeError with the build hint when it is absent."""
    global _lib
    if _lib is None:
        path = os.environ.get("BIN_AMD_LIB", LIB_PATH)      # developer knob: tools/ experiments load side builds
        if not os.path.exists(path):
            raise RuntimeError(
                f"bin_amd: HIP library {LIB_PATH} not built. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback by design.")
        h = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        for name, (res, args) in _TUNING_SIGNATURES.items():
            if hasattr(h, name):
                fn = getattr(h, name)
                fn.restype = res
                fn.argtypes = args
        # developer knob, tuning side builds only (the product library has no such entry point): pick the experimental
        # weight-gradient kernels / ablations of binhip_wgrad.hip for a whole test or bench run
        dbg = os.environ.get("BIN_AMD_WG_DEBUG")
        if dbg and hasattr(h, "binhip_wgrad_set_debug"):
            h.binhip_wgrad_set_debug(int(dbg, 0))
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        kind = {-1: "bad argument", -2: "unsupported shape", -3: "workspace too small"}.get(rc, f"hipError {rc}")
        raise RuntimeError(f"bin_amd: {what} failed: {kind}")
