"""ctypes binding of the in-tree C-ABI library (include/cra5_amd.h).

There is NO fallback: if `libcra5_amd.so` is missing the import of any compute
entry point raises.  Build it with `python -m cra5_amd.build`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRA5_LIB") or os.path.join(_HERE, "libcra5_amd.so")  # CRA5_LIB: A/B builds

c_int, c_float, c_size_t, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p
P = ctypes.POINTER

# name -> (restype, argtypes); mirrors include/cra5_amd.h one to one
SIGNATURES = {
    "cra5_abi_version": (c_int, []),
    "cra5_rans_encode_with_indexes": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                              P(c_void_p), P(c_size_t)]),
    "cra5_rans_encode_resolved": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, P(c_void_p), P(c_size_t)]),
    "cra5_rans_decode_with_indexes": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_int, c_int, c_void_p,
                                              c_void_p, c_void_p]),
    "cra5_rans_encode_resolved_compact": (c_int, [c_void_p, c_void_p, c_size_t, P(c_void_p), P(c_size_t)]),
    "cra5_rans_resolve_symbols_compact": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "cra5_rans_decode_with_indexes_u8_i16": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_int, c_int,
                                                     c_void_p, c_void_p, c_void_p]),
    "cra5_rans_encode_batch": (c_int, [c_int, P(c_void_p), P(c_void_p), P(c_size_t), P(c_void_p), P(c_int), P(c_int),
                                       P(c_void_p), P(c_void_p), P(c_void_p), P(c_size_t), P(c_int), c_int]),
    "cra5_rans_decode_batch": (c_int, [c_int, P(c_void_p), P(c_size_t), P(c_void_p), P(c_size_t), P(c_void_p), P(c_int),
                                       P(c_int), P(c_void_p), P(c_void_p), P(c_void_p), P(c_int), c_int]),
    "cra5_free": (None, [c_void_p]),
    "cra5_rans_encoder_new": (c_void_p, []),
    "cra5_rans_encoder_push": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_void_p,
                                       c_void_p]),
    "cra5_rans_encoder_flush": (c_int, [c_void_p, P(c_void_p), P(c_size_t)]),
    "cra5_rans_encoder_free": (None, [c_void_p]),
    "cra5_rans_decoder_new": (c_void_p, []),
    "cra5_rans_decoder_set_stream": (c_int, [c_void_p, c_void_p, c_size_t]),
    "cra5_rans_decoder_decode_stream": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_void_p,
                                                c_void_p, c_void_p]),
    "cra5_rans_decoder_free": (None, [c_void_p]),
    "cra5_pmf_to_quantized_cdf": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "cra5_gemm_nt_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int,
                                 c_int, c_int, c_int, c_void_p]),
    "cra5_gemm_nt_split": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                   c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "cra5_split_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "cra5_layernorm_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int,
                                   c_int, c_float, c_int, c_void_p]),
    "cra5_window_attention_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_int, c_int, c_float, c_void_p]),
    "cra5_window_attention_split": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "cra5_gaussian_conditional_compact_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, c_void_p,
                                                      c_void_p, c_size_t, c_void_p]),
    "cra5_unembed_side_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "cra5_gemm_nt_split_unembed": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p,
                                           c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_void_p]),
    "cra5_attention_workspace_bytes": (c_size_t, [c_int, c_int]),
    "cra5_attention_balanced_plan": (c_int, [c_int, c_int, P(c_size_t)]),
    "cra5_window_attention_split_ws": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                               c_int, c_int, c_int, c_float, c_int, c_void_p, c_size_t, c_void_p]),
    "cra5_im2col_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    "cra5_col2im_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "cra5_probe_sums_f32": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "cra5_transpose_f32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "cra5_pixel_shuffle_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "cra5_conv_im2col_f32": (c_int, [c_void_p, c_void_p] + [c_int] * 12 + [c_void_p]),
    "cra5_deconv_col2im_f32": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 12 + [c_void_p]),
    "cra5_unary_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_float, c_void_p]),
    "cra5_gaussian_conditional_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                                              c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "cra5_entropy_bottleneck_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                            c_void_p, c_int, c_int, c_void_p]),
    "cra5_rans_resolve_symbols_i32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_void_p]),
    "cra5_gdn_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "cra5_small_gemm_nt_split": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                         c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int,
                                         c_int, c_void_p]),
    "cra5_hyper_attention_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "cra5_event_create": (c_int, [P(c_void_p)]),
    "cra5_event_record": (c_int, [c_void_p, c_void_p]),
    "cra5_event_elapsed_ms": (c_int, [c_void_p, c_void_p, P(c_float)]),
    "cra5_event_destroy": (c_int, [c_void_p]),
    "cra5_copy_h2d_staged": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_void_p]),
    "cra5_copy_d2h_staged": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_void_p]),
    "cra5_clock_probe": (c_int, [c_void_p, c_int, c_void_p]),
    "cra5_clock_stamp": (c_int, [c_void_p, c_void_p]),
    "cra5_debug_range_counts": (c_int, [P(ctypes.c_uint64), c_int]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is mandatory (no CPU fallback). "
                "Run `python -m cra5_amd.build` (hipcc, gfx950).")
        # torch must be loaded first: it bundles its own HIP runtime (libamdhip64.so.7 in
        # torch/lib); loading ours first would pull /opt/rocm's copy under the same SONAME and
        # torch then fails with hipErrorNoDevice (two runtimes, one process).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the header and the library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class Cra5Error(RuntimeError):
    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


ERR_RANGE = -9     # CRA5_ERR_RANGE: a value does not fit the compact record type
ERR_DESYNC = -10   # CRA5_ERR_DESYNC: the stream decoded to its end without returning the coder to its initial state


class StreamDesyncError(Cra5Error):
    """A rANS stream was decoded to the last symbol, but the coder did not come back to RANS64_L with every word
    consumed: the (tables, indexes) the decoder used are not the encoder's, or the stream is damaged.  The symbols that
    were decoded are NOT the coded ones.  (The reference's decoder has no such check and returns them:
    rans_interface.cpp:215-284.)"""


def check(rc, what):
    if rc == ERR_DESYNC:
        raise StreamDesyncError(
            f"{what}: the stream decoded to its last symbol but the coder did not return to its initial state "
            f"(status {rc}): the CDF indexes / tables are not the encoder's, or the stream is damaged", rc)
    if rc != 0:
        raise Cra5Error(f"{what} failed with status {rc}", rc)
