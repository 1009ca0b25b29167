"""ctypes binding of libcti_hip.so (the C ABI declared in include/cti_hip.h).  No torch types cross this boundary:
pointers are `tensor.data_ptr()` integers, the stream is `torch.cuda.current_stream().cuda_stream`."""
import ctypes as C
import os

from . import _build

_i64, _int, _vp, _sz = C.c_int64, C.c_int, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); must list every function include/cti_hip.h declares (tests/test_abi.py checks it)
SIGNATURES = {
    "cti_abi_version": (_int, []),
    "cti_last_error_string": (C.c_char_p, []),
    "cti_set_tuning": (_int, [_int, _i64]),
    "cti_get_tuning": (_i64, [_int]),
    "cti_wn_scale": (_int, [_vp, _vp, _vp, _int, _i64, _vp, _sz, _vp]),
    "cti_wn_scale_workspace_bytes": (_sz, [_int, _i64]),
    "cti_wn_scale_many": (_int, [_vp, _vp, _vp, _vp, _int, _vp, _sz, _vp]),
    "cti_wn_scale_many_workspace_bytes": (_sz, [_vp, _int]),
    "cti_wn_linear_fwd": (_int, [_vp, _i64, _vp, _i64, _vp, _int, _vp, _vp, _i64, _i64, _int, _int, _int, _int, _vp, _sz, _vp]),
    "cti_wn_linear_workspace_bytes": (_sz, [_i64, _int, _int, _int]),
    "cti_zero_row_mask": (_int, [_vp, _i64, _vp, _i64, _int, _vp]),
    "cti_zero_row_mask_bf16": (_int, [_vp, _i64, _vp, _i64, _int, _vp]),
    "cti_teff_scramble": (_int, [_vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_paralind_mbuild_fwd": (_int, [_vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_paralind_mbuild_planes_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _i64, _vp]),
    "cti_paralind_mbuild_f16f6_fwd": (_int, [_vp, _vp, _vp, _vp, _sz, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_paralind_core_fwd": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _sz, _vp]),
    "cti_paralind_core_workspace_bytes": (_sz, [_int, _int, _int, _int, _int, _int]),
    "cti_event_create": (_vp, []),
    "cti_event_destroy": (_int, [_vp]),
    "cti_event_record": (_int, [_vp, _vp]),
    "cti_event_elapsed_ms": (_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "cti_tcnet_forward": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp] + [_int] * 12 + [_vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "cti_tcnet_forward_sm": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp] + [_int] * 12 + [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz]),
    "cti_tcnet_softmax_partials_bytes": (_sz, [_int] * 7),
    "cti_triattention_workspace_bytes": (_sz, [_int] * 11),
    "cti_triattention_forward": (_int, [_vp] * 13 + [_int] * 12 + [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _i64, _int]),
    "cti_triattention_forward_vt16": (_int, [_vp] * 13 + [_int] * 12 + [_vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _i64, _int]),
    "cti_triattention_hoist_ok": (_int, [_int] * 8),
    "cti_tcnet_forward_guard_bytes": (_sz, [_int] * 11),
    "cti_guard_read": (_int, [_vp, _vp, _vp, C.POINTER(C.c_uint32)]),
    "cti_guard_read_ratio": (_int, [_vp, _vp, C.POINTER(C.c_float)]),
    "cti_masked_softmax_tri_from_partials_fwd": (_int, [_vp, _vp, _vp, _sz, _vp, _int, _int, _i64, _int, _vp, _sz, _vp]),
    "cti_tcnet_prepare": (_int, [_vp, _vp, _vp, _vp, _vp] + [_int] * 7 + [_vp, _sz, _vp]),
    "cti_tcnet_prepared_bytes": (_sz, [_int] * 7),
    "cti_tcnet_forward_workspace_bytes": (_sz, [_int] * 11),
    "cti_gemm_nt": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int,
                           _vp, _int, _i64, _vp, _i64, _int, _int, _vp, _sz, _vp]),
    "cti_gemm_nt_workspace_bytes": (_sz, [_i64, _i64, _int, _int]),
    "cti_transpose_f32": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _int, _int, _int, _vp]),
    "cti_sum_batches": (_int, [_vp, _vp, _int, _i64, C.c_float, C.c_float, _vp]),
    "cti_act_bwd": (_int, [_vp, _vp, _vp, _int, _vp, _vp, _i64, _int, _int, _vp, _sz, _vp]),
    "cti_act_bwd_workspace_bytes": (_sz, [_i64, _int]),
    "cti_wn_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _i64, _vp, _sz, _vp]),
    "cti_wn_bwd_workspace_bytes": (_sz, [_int, _i64]),
    "cti_paralind_mbuild_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_paralind_mbuild_bwd_mfma_partials": (_int, [_int, _int]),
    "cti_paralind_mbuild_bwd_mfma": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_paralind_mbuild_bwd_generic": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _sz, _vp]),
    "cti_paralind_mbuild_bwd_generic_workspace_bytes": (_sz, [_int, _int, _int, _int, _int]),
    "cti_masked_softmax_tri_bwd": (_int, [_vp, _vp, _vp, _int, _int, _i64, _int, _vp, _sz, _vp]),
    "cti_softmax_tri_bwd_workspace_bytes": (_sz, [_int, _int, _i64, _int]),
    "cti_masked_softmax_bi_bwd": (_int, [_vp, _vp, _vp, _int, _int, _vp]),
    "cti_tri_pool_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "cti_bi_pool_bwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "cti_bi_logits_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "cti_flat_scale_sumsq": (_int, [_vp, _i64, C.c_float, _vp, _vp]),
    "cti_flat_gather": (_int, [_vp, _int, _vp, _i64, _vp]),
    "cti_adamax_step": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _int, _vp, _vp]),
    "cti_optim_workspace_bytes": (_sz, []),
    "cti_embedding_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i64, _vp]),
    "cti_embedding_fwd_bf16": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _int, _i64, _vp]),
    "cti_embedding_bwd": (_int, [_vp, _vp, _i64, _int, _vp, _i64, _int, _i64, _i64, _vp]),
    "cti_gru_forward": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp, _vp, _vp, _sz, _vp]),
    "cti_gru_forward_x16": (_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp, _vp, _vp, _sz, _vp]),
    "cti_operand_planes_bytes": (_sz, [_i64, _int]),
    "cti_split_operand": (_int, [_vp, _i64, _i64, _int, _vp, _sz, _vp]),
    "cti_gemm_nt_pb": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _int, _int, _int, _int, _vp, _int, _i64, _vp, _i64, _int, _int, _vp, _sz, _vp]),
    "cti_gemm_nt_pb_workspace_bytes": (_sz, [_i64, _i64, _int, _int]),
    "cti_gemm_nt_pb_workspace_bytes2": (_sz, [_i64, _i64, _int, _int, _int, _int, _int]),
    "cti_gemm_bf16_rows": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _int, _i64, _i64, _int, _int, _int, _int, _vp, _int, _i64, _vp, _i64, _int, _vp]),
    "cti_gemm_bf16_rows_sk": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _i64, _vp, _int, _i64, _i64, _int, _int, _int, _int, _vp, _int, _i64, _vp, _i64, _int, _vp, _sz, _vp]),
    "cti_gemm_bf16_rows_sk_workspace_bytes": (_sz, []),
    "cti_gru_forward_workspace_bytes": (_sz, [_int, _int, _int, _int, _int]),
    "cti_gru_backward": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _vp, _sz, _vp]),
    "cti_gru_backward_workspace_bytes": (_sz, [_int, _int, _int, _int]),
    "cti_col_sum": (_int, [_vp, _i64, _int, _vp, C.c_float, C.c_float, _vp, _sz, _vp]),
    "cti_col_sum_workspace_bytes": (_sz, [_i64, _int]),
    "cti_f16f6_planes_bytes": (_sz, [_i64, _int, _i64]),
    "cti_quantize_f16f6": (_int, [_vp, _i64, _i64, _int, _i64, _vp, _sz, _vp]),
    "cti_quantize_f16f6_into": (_int, [_vp, _i64, _i64, _int, _i64, _vp, _sz, _vp]),
    "cti_gemm_nt_f16f6": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _int, _int, _int, _int, _int, _vp, _int, _vp, _int, _vp]),
    "cti_gemm_nt_f16f6_planes": (_int, [_vp, _i64, _vp, _i64, _vp, _sz, _i64, _int, _int, _int, _vp, _int, _vp]),
    "cti_quantize_f16f6_scaled": (_int, [_vp, _i64, _i64, _int, _i64, _vp, _int, _vp, _sz, _vp]),
    "cti_gemm_tn": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _int, _int, _int, _vp, _sz, _vp]),
    "cti_gemm_nn": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _int, _int, _int, _vp, _sz, _vp]),
    "cti_gemm_nn_workspace_bytes": (_sz, [_i64, _int, _int, _int]),
    "cti_gemm_tn_workspace_bytes": (_sz, [_i64, _int, _int, _int]),
    "cti_swish_fwd": (_int, [_vp, _vp, _i64, _vp]),
    "cti_swish_bwd": (_int, [_vp, _vp, _vp, _i64, _vp]),
    "cti_seq_sum": (_int, [_vp, _vp, _int, _int, _int, C.c_float, _vp]),
    "cti_seq_bcast_add": (_int, [_vp, _vp, _vp, _int, _int, _int, _vp]),
    "cti_linear_residual_workspace_bytes": (_sz, [_int] * 4),
    "cti_linear_residual_pb": (_int, [_vp, _i64, _vp, _vp, _int, _vp, _vp, _vp, _vp, C.c_float, _int, _int, _int, _int, _int, _vp, _sz, _vp]),
    "cti_bce_logits_rows_fwd": (_int, [_vp, _vp, _vp, _int, _int, _vp]),
    "cti_bce_logits_bwd": (_int, [_vp, _vp, _vp, C.c_float, _vp, _i64, C.c_float, _vp]),
    "cti_kd_rows_fwd": (_int, [_vp, _vp, _vp, _int, _int, C.c_float, _vp]),
    "cti_kd_rows_bwd": (_int, [_vp, _vp, _vp, C.c_float, _vp, _int, _int, C.c_float, C.c_float, _vp]),
    "cti_dropout_g": (_int, [_vp, _vp, _vp, _i64, C.c_float, C.c_uint64, C.c_uint64, _int, _i64, _vp, _vp]),
    "cti_adamax_step_g": (_int, [_vp, _vp, _vp, _vp, _i64, _vp, C.c_float, _vp, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp]),
    "cti_counter_add": (_int, [_vp, _i64, _vp]),
    "cti_dropout": (_int, [_vp, _vp, _vp, _i64, C.c_float, C.c_uint64, C.c_uint64, _int, _i64, _vp]),
    "cti_paralind_core_bwd": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_paralind_core_planes_fwd": (_int, [_vp, _vp, _i64, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp, _sz, _vp]),
    "cti_paralind_core_planes_workspace_bytes": (_sz, [_int, _int, _int, _int]),
    "cti_paralind_core_bwd_planes": (_int, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_ranknets_drop_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, C.c_float, _int, _vp]),
    "cti_ranknets_drop_fwd_mfma_workspace_bytes": (_sz, [_int, _int, _int]),
    "cti_ranknets_drop_fwd_mfma": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, C.c_float, _int, _int, _vp, _sz, _vp]),
    "cti_ranknets_drop_dw": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, C.c_float, _vp]),
    "cti_ranknets_drop_dx": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _int, C.c_float, _vp]),
    "cti_masked_softmax_tri_fwd": (_int, [_vp, _vp, _vp, _int, _int, _i64, _int, _vp, _sz, _vp]),
    "cti_softmax_tri_workspace_bytes": (_sz, [_int, _int, _i64, _int]),
    "cti_masked_softmax_bi_fwd": (_int, [_vp, _vp, _vp, _int, _int, _int, _int, _vp]),
    "cti_tri_pool_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _int, _int, _int, _int, _vp]),
    "cti_tri_pool_mfma_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_bi_logits_bwd_mfma": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "cti_row_sum": (_int, [_vp, _vp, _i64, _int, _vp]),
    "cti_pool_dw_mfma": (_int, [_vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "cti_bi_pool_fwd": (_int, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _int, _int, _int, _int, _int, _vp]),
    "cti_bi_pool_shift_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _int, _int, _int, _int, _vp]),
    "cti_bi_pool_shift_multi_fwd": (_int, [_vp, _int, _vp, _vp, _vp, _int, _vp, _i64, _i64, _i64, _vp, _i64, _int, _int, _int, _int, _vp]),
    "cti_gemm_pb_partials_count": (_int, [_int, _int, _int]),
    "cti_gemm_pb_partials_workspace_bytes": (_sz, [_int, _int, _int]),
    "cti_gemm_pb_partials": (_int, [_vp, _i64, _vp, _int, _int, _int, _int, _vp, _sz, _vp, _sz, _vp]),
    "cti_bi_pool_shift_vt16_fwd": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp, _int, _int, _int, _int, _vp]),
    "cti_tri_pool_shift_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_tri_pool_shift_vt16_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _int, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_axpby": (_int, [_vp, C.c_float, _vp, C.c_float, _vp, _i64, _vp]),
    "cti_joint_sums": (_int, [_vp, _int, C.c_float, _vp, _int, C.c_float, _vp, C.c_float, _vp, C.c_float, _vp, _int, _int, _vp]),
    "cti_linear_small_n": (_int, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _int, _int, _int, _int, _vp]),
    "cti_rows_equal_prev": (_int, [_vp, _i64, _int, _vp, _vp]),
    "cti_poison_unless_replicated": (_int, [_vp, _int, _int, _vp, _i64, _vp]),
    "cti_bi_logits_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "cti_bi_logits_mfma_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
    "cti_bi_logits_prec_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_bi_logits_prec_vt16_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _int, _vp]),
    "cti_biattention_fwd": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _int, _int, _int, _int, _vp]),
}

PREC_F32, PREC_BF16X3, PREC_BF16, PREC_F16F6 = 0, 1, 2, 3
E_UNSUPPORTED = -4                                   # CTI_E_UNSUPPORTED: shape / mode outside a specialised kernel
ACT_NONE, ACT_RELU = 0, 1
TUNE_GEMM_CFG, TUNE_TRI_CHUNK = 1, 2                # cti_set_tuning keys
TUNE_GUARD_RHO_BF16X3, TUNE_GUARD_RHO_FP32, TUNE_GUARD_POISON_BITS, TUNE_F6_CORE_FREE_CUS, TUNE_GUARD_STRATA, TUNE_GEMM16_SK, TUNE_GRU_PERSISTENT = 3, 4, 5, 6, 7, 8, 9     # f16f6 guard policy (include/cti_hip.h)
GUARD_SATURATED, GUARD_UNDERFLOW, GUARD_NONFINITE = 1, 2, 4     # status bits of the f16f6 range guard (cti_guard_read)

_lib = None


class CtiError(RuntimeError):
    pass


def lib():
    """Load (building first if the in-tree .so is missing or stale and hipcc is present).  There is NO fallback:
    without the library every op of this package raises."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    override = os.environ.get("CTI_HIP_LIB")           # kernel-variant A/B (tools/tune_gemm.py build ...): another build of the SAME library
    if override:
        path = override
    elif _build.stale():
        try:
            _build.build()
        except Exception as e:  # no hipcc on this machine: use the prebuilt copy if there is one
            if not os.path.isfile(path):
                raise CtiError("libcti_hip.so is missing and could not be built: %s" % e)
    try:
        l = C.CDLL(path)
    except OSError as e:
        raise CtiError("cannot load %s: %s (the CTI modules have no non-HIP path)" % (path, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            f = getattr(l, name)
        except AttributeError:
            raise CtiError("%s does not export %s -- rebuild it (python -c 'import __graft_entry__ as g; g.build()')" % (path, name))
        f.restype, f.argtypes = res, args
    if l.cti_abi_version() != 1:
        raise CtiError("libcti_hip.so ABI version %d, expected 1" % l.cti_abi_version())
    _lib = l
    return l


def check(rc, what):
    if rc != 0:
        msg = lib().cti_last_error_string().decode("utf-8", "replace")
        raise CtiError("%s failed (%d): %s" % (what, rc, msg))
