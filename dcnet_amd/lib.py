"""ctypes binding of libdcnet_hip.so (include/dcnet_hip.h).

There is NO fallback: if the library is missing or a symbol is absent this module raises,
and every wrapper raises on a non-zero return code with the library's own error text.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdcnet_hip.so")

P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> (restype, argtypes).  int-returning entries are status codes unless listed in _VALUE_FUNCS.
SIGNATURES = {
    "dcn_last_error": (c_char_p, []),
    "dcn_version": (I, []),
    "dcn_nchw_to_nhwc": (I, [P, P, I, I, I, I, I, P]),
    "dcn_nhwc_to_nchw": (I, [P, P, I, I, I, I, I, P]),
    "dcn_oihw_to_ohwi": (I, [P, P, I, I, I, I, I, P]),
    "dcn_ohwi_to_oihw": (I, [P, P, I, I, I, I, I, P]),
    "dcn_conv2d_fwd": (I, [P, P, P, I, I, I, I, I, I, I, P, P, I, F, P, I, I, P, I, P, P, P, P, P, I, P]),
    "dcn_absmax": (I, [P, L, I, I, P, P]),
    "dcn_f8_scale": (I, [P, L, I, I, P, P, P]),
    "dcn_conv2d_stats_rows": (I, [I, I, I, I, I, I]),
    "dcn_conv2d_bwd_data": (I, [P, I, P, P, P, I, I, I, I, I, I, I, I, P, P, P, I, P, P]),
    "dcn_conv2d_bwd_data_tap": (I, [P, I, P, P, P, I, I, I, I, I, I, I, I, P, P, P, I, P, P, P, P, P, P, I, F, P, I, P, P]),
    "dcn_conv2d_bwd_data_tap_rows": (I, [I, I, I, I, I, I, I]),
    "dcn_conv2d_pre_supported": (I, [I, I, I, I, I, I, I]),
    "dcn_conv2d_fwd_pre": (I, [P, P, P, I, I, I, I, I, I, I, P, P, I, F, I, P, P, P, P]),
    "dcn_conv2d_bwd_weight_pre_supported": (I, [I, I, I, I, I, I, I]),
    "dcn_conv2d_bwd_weight_pre": (I, [P, I, P, I, P, P, P, I, I, I, I, I, I, I, P, P, I, F, P, P, P]),
    "dcn_bn_act_amax_bound": (I, [P, P, P, I, F, P, P]),
    "dcn_gemm3_supported": (I, [I, I, I, I]),
    "dcn_gemm3_presplit": (I, [P, I, L, P, I, L, I, I, I, P, P]),
    "dcn_gemm3": (I, [P, I, L, I, P, I, L, I, P, I, L, P, L, I, I, I, I, I, P, P, P]),
    "dcn_filter_job_bytes": (I, []),
    "dcn_prepare_filters": (I, [P, I, I, I, P, L, P]),
    "dcn_conv2d_geom_size": (L, [I, I, I, I, I]),
    "dcn_conv2d_geom": (I, [P, I, I, I, I, I, P]),
    "dcn_conv2d_bwd_weight": (I, [P, I, P, I, P, P, P, P, I, I, I, I, I, I, I, P, P, P]),
    "dcn_conv2d_bwd_weight_ws": (L, [I, I, I, I, I, I, I]),
    "dcn_stem_bwd_weight_bn": (I, [P, P, P, I, P, P, P, P, I, F, P, L, I, I, I, I, P, P, P]),
    "dcn_stem_bwd_weight_bn_ws": (L, [I, I, I]),
    "dcn_stem_bwd_weight_bn_b16": (I, [P, P, P, I, P, P, P, P, I, F, P, L, I, I, I, I, P, P, P]),
    "dcn_bn_ws": (L, [I]),
    "dcn_bn_finalize": (I, [P, I, I, L, P, P, F, F, P, P, P, P, P, P, P, P]),
    "dcn_bn_fold": (I, [P, P, P, P, F, I, P, P, P]),
    "dcn_channel_stats": (I, [P, L, I, I, P, P]),
    "dcn_channel_stats_rows": (I, [L]),
    "dcn_scale_act": (I, [P, P, P, I, F, P, P, L, I, I, P, P]),
    "dcn_bn_act_bwd_reduce": (I, [P, P, I, P, P, P, P, I, F, L, I, P, P]),
    "dcn_bn_bwd_sums": (I, [P, I, I, P, P, P]),
    "dcn_bn_act_bwd_apply": (I, [P, P, I, P, P, P, P, I, F, P, L, L, I, P, P, P]),
    "dcn_act_bwd": (I, [P, P, I, F, L, I, P, P]),
    "dcn_coattn_e_size": (L, [I, I]),
    "dcn_coattn_saved_size": (L, [I, I, I]),
    "dcn_coattn_fwd_ws": (L, [I, I, I]),
    "dcn_coattn_fwd": (I, [P, P, I, L, P, P, I, L, P, P, P, P, I, I, I, F, P]),
    "dcn_coattn_bwd_ws": (L, [I, I, I]),
    "dcn_coattn_bwd": (I, [P, P, I, L, P, P, I, L, P, P, I, L, P, P, P, P, P, I, L, I, P, I, I, I, F, P]),
    "dcn_l2norm_score_fwd": (I, [P, I, P, I, P, P, P, P, L, I, I, F, I, P]),
    "dcn_l2norm_score_bwd": (I, [P, I, P, P, I, P, P, P, P, I, P, L, I, I, P]),
    "dcn_rowdot_fwd": (I, [P, I, P, I, P, L, I, I, P]),
    "dcn_rowdot_bwd": (I, [P, I, P, I, P, P, I, P, L, I, I, P]),
    "dcn_phrase_attn_fwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "dcn_phrase_attn_bwd_ws": (L, [I, I, I]),
    "dcn_phrase_attn_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "dcn_colsum": (I, [P, I, I, I, P, P]),
    "dcn_locemb_fwd": (I, [P, P, P, P, P, P, P, F, F, I, L, I, P, P, P, P, P]),
    "dcn_locemb_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, P, P]),
    "dcn_head_obj": (I, [P, P, P, P, P, P, P, P, P, I, P]),
    "dcn_pad_rows": (I, [P, I, P, I, I, I, I, P]),
    "dcn_locbn_fwd": (I, [P, P, P, P, P, P, P, F, F, I, I, L, I, P, P, P, P]),
    "dcn_locbn_bwd": (I, [P, P, P, P, P, P, P, L, P, I, I, L, I, P, P, P, P]),
    "dcn_head_final_fwd": (I, [P, P, P, P, P, P, P, P, I, P]),
    "dcn_head_dloc": (I, [P, P, P, P, P, P, P, P, P, I, P]),
    "dcn_head_fold": (I, [P, P, P, I, I, I, P, P, P]),
    "dcn_head_dlogits": (I, [P, P, P, P, P, P, P, P, P, P, P, P, P, I, P]),
    "dcn_k9_fwd": (I, [P, P, P, I, I, I, I, I, P, P, P, P, P, P]),
    "dcn_k9_bwd": (I, [P, P, P, P, P, I, I, I, I, I, P, P]),
    "dcn_colnorm_fwd": (I, [P, I, I, I, P, P, P]),
    "dcn_colnorm_bwd": (I, [P, P, P, P, I, I, I, I, P, P]),
    "dcn_lagnorm_fwd": (I, [P, I, I, I, P, P, P]),
    "dcn_crossmap": (I, [P, P, P, P, I, I, I, I, P, P, P]),
    "dcn_k14_gather": (I, [P, P, P, P, I, I, I, I, I, P, P, P]),
    "dcn_k14_negscatter": (I, [P, P, P, I, I, P, P]),
    "dcn_k14_dlag": (I, [P, P, P, P, I, I, I, I, P, P]),
    "dcn_build_target": (I, [P, P, I, I, P, P, P]),
    "dcn_target_dense": (I, [P, P, I, I, P, P, P]),
    "dcn_dense_loss_fwd": (I, [P, P, P, P, P, P, I, I, P, P, P, P]),
    "dcn_dense_loss_bwd": (I, [P, P, P, P, P, P, P, P, I, I, P, P, P, P, P]),
    "dcn_contrastive_fwd": (I, [P, P, P, L, I, I, F, P, P, P]),
    "dcn_contrastive_bwd": (I, [P, P, P, L, I, I, F, P, P, P, P, P]),
    "dcn_decode_boxes": (I, [P, P, I, I, P, P, P]),
    "dcn_box_iou": (I, [P, P, I, P, P]),
    "dcn_gemm_nt_batched": (I, [P, I, L, P, I, L, P, I, L, I, I, I, I, P]),
    "dcn_mt_sample_crossmodal_csr": (I, [P, I, I, I, P, P]),
    "dcn_upsample2_nhwc": (I, [P, I, P, I, I, I, I, I, P]),
    "dcn_upsample2_nhwc_bwd": (I, [P, I, P, I, I, I, I, I, I, P]),
    "dcn_copy_slice": (I, [P, I, P, I, L, I, I, P]),
    "dcn_locmod_fwd": (I, [P, P, P, P, P, I, I, I, P]),
    "dcn_locmod_bwd_ws": (L, [I, I]),
    "dcn_locmod_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    "dcn_gemm_nt": (I, [P, I, P, I, P, I, I, I, I, P, I, P, I, I, P]),
    "dcn_gemm_nn": (I, [P, I, P, I, P, I, I, I, I, I, I, P]),
    "dcn_gemm_tn": (I, [P, I, P, I, P, I, I, I, I, I, P]),
    "dcn_lstm_cell_fwd": (I, [P, P, P, P, I, P, P, P, P, I, I, I, P]),
    "dcn_lstm_cell_bwd": (I, [P, I, P, P, P, P, P, I, P, I, P, P, I, I, P]),
    "dcn_bilstm_sync_bytes": (L, []),
    "dcn_bilstm_fwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "dcn_bilstm_bwd": (I, [P, P, P, P, P, P, P, P, I, I, I, P]),
    "dcn_rmsprop_step": (I, [P, P, P, P, I, F, P, F, F, F, P]),
    "dcn_fusion_prefill": (I, [P, P, P, I, P, I, I, I, P]),
    "dcn_fusion_bwd_ws": (L, [I, I]),
    "dcn_fusion_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "dcn_row_lengths": (I, [P, I, I, P, P]),
    "dcn_embedding_fwd": (I, [P, P, P, I, I, I, P]),
    "dcn_embedding_bwd": (I, [P, P, P, I, I, I, P]),
    "dcn_set_tuning": (I, [c_char_p, I]),
    "dcn_stream_create": (P, [I]),
    "dcn_stream_destroy": (I, [P]),
    "dcn_stream_priority_range": (I, [P, P]),
    "dcn_prof_enable": (I, [I]),
    "dcn_prof_collect": (I, [P, P, P, P]),
    "dcn_prof_records": (I, [P, P, P, P, I]),
    "dcn_mt_sample_interframe": (I, [P, P, I, I, I, I, P]),
    "dcn_mt_sample_crossmodal": (I, [P, I, I, I, P]),
    "dcn_quant_job_bytes": (I, []),
    "dcn_quant_rows_e4m3_batched": (I, [P, I, I, L, P]),
    "dcn_mt_sample_step": (I, [P, I, I, I, I, I, P, P, P, P, P]),
    "dcn_device_sample_ws": (L, [I]),
    "dcn_device_sample": (I, [P, I, I, I, I, I, P, P, P, P, P, P]),
    "dcn_conv2d_stats_rows_b16": (I, [I, I, I, I, I, I]),
    "dcn_conv2d_fwd_b16": (I, [P, P, P, I, I, I, I, I, I, I, I, P, P, I, F, P, I, I, P, I, P]),
    "dcn_conv2d_bwd_data_b16": (I, [P, I, P, P, I, I, I, I, I, I, I, I, I, P, P, P, P, P, I, F, P, I, P, P]),
    "dcn_conv2d_bwd_weight_ws_b16": (L, [I, I, I, I, I, I, I]),
    "dcn_conv2d_bwd_weight_b16": (I, [P, I, P, I, P, P, P, P, I, I, I, I, I, I, I, P]),
    "dcn_scale_act_b16": (I, [P, I, P, P, I, F, P, I, P, I, L, I, I, P]),
    "dcn_bn_act_bwd_reduce_rows_b16": (I, [L]),
    "dcn_bn_act_bwd_reduce_b16": (I, [P, I, P, I, I, P, P, P, P, I, F, L, I, P, P]),
    "dcn_bn_act_bwd_apply_b16": (I, [P, I, P, I, I, P, P, P, P, I, F, P, L, L, I, P, P]),
    "dcn_cast_rows": (I, [P, I, I, P, I, I, L, I, I, P]),
    "dcn_upsample2_nhwc_b16": (I, [P, I, P, I, I, I, I, I, P]),
    "dcn_upsample2_nhwc_bwd_b16": (I, [P, I, P, I, I, I, I, I, I, P]),
    "dcn_quant_rows_e4m3": (I, [P, I, L, I, P, I, P, P]),
    "dcn_quant_fusable": (I, [I]),
    "dcn_scale_act_b16_q": (I, [P, P, P, I, F, P, I, P, L, I, P, P, P]),
    "dcn_bn_act_bwd_apply_b16_q": (I, [P, P, I, P, P, P, P, I, F, P, L, L, I, P, P, P, P]),
    "dcn_conv2d_stats_rows_f8": (I, [I, I, I, I, I, I]),
    "dcn_conv2d_fwd_f8": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, P, P, I, F, P, I, I, P, I, P]),
    "dcn_conv2d_bwd_data_f8": (I, [P, P, P, P, P, I, I, I, I, I, I, I, I, I, P, P, P, P, P, I, F, P, I, P, P]),
    "dcn_post_topk": (I, [P, P, P, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P]),
    "dcn_post_fusion": (I, [P, P, P, P, I, I, I, I, P, P, P]),
}
_VALUE_FUNCS = {"dcn_version", "dcn_conv2d_stats_rows", "dcn_conv2d_bwd_data_tap_rows", "dcn_conv2d_pre_supported",
                "dcn_conv2d_bwd_weight_pre_supported", "dcn_gemm3_supported", "dcn_channel_stats_rows", "dcn_filter_job_bytes", "dcn_prof_records",
                "dcn_conv2d_stats_rows_b16", "dcn_bn_act_bwd_reduce_rows_b16", "dcn_conv2d_stats_rows_f8", "dcn_quant_fusable", "dcn_quant_job_bytes"}
ABI_VERSION = 308        # include/dcnet_hip.h DCN_ABI_VERSION this table was written for      # int-returning value functions


class DcnError(RuntimeError):
    pass


class _Lib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise DcnError(
                f"{path} not found: build it with `python -m dcnet_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU or eager fallback.")
        self._dll = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise DcnError(f"{path} does not export {name}; rebuild the library") from e
            fn.restype = res
            fn.argtypes = args
            if res is I and name not in _VALUE_FUNCS:
                setattr(self, name[4:], self._checked(name, fn))
            else:
                setattr(self, name[4:], fn)
        # experiments: DCN_TUNE="key=value,key=value" applies dcn_set_tuning knobs at load time (e.g. to run a test under a knob)
        for kv in [t for t in os.environ.get("DCN_TUNE", "").split(",") if t]:
            k_, v_ = kv.split("=")
            if self._dll.dcn_set_tuning(k_.encode(), int(v_)) != 0:
                raise DcnError(f"DCN_TUNE: unknown knob {k_!r}")
        if self._dll.dcn_version() != ABI_VERSION:
            raise DcnError(f"{path} has ABI version {self._dll.dcn_version()}, this binding was written for {ABI_VERSION}: "
                           "rebuild the library (python -m dcnet_amd.build --force)")

    def _checked(self, name, fn):
        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise DcnError(f"{name} failed ({rc}): {self._dll.dcn_last_error().decode()}")
        call.__name__ = name
        return call


_lib = None


def lib() -> _Lib:
    """The loaded library (loaded on first use; raises DcnError if it cannot be)."""
    global _lib
    if _lib is None:
        _lib = _Lib(LIB_PATH)
    return _lib
