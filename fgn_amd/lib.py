"""ctypes binding of libfgn_hip.so (include/fgn_hip.h).

There is no CPU fallback: if the library is missing or a symbol is absent this module
raises, and every op raises on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FGN_HIP_LIB') or os.path.join(_HERE, 'libfgn_hip.so')     # FGN_HIP_LIB: A/B builds (tools/)

_p = C.c_void_p
_i = C.c_int
_f = C.c_float

# name -> (restype, argtypes); mirrors include/fgn_hip.h one to one
SIGNATURES = {
    'fgn_abi_version': (_i, []),
    'fgn_profile_next_launch': (_i, [_p, _p]),
    'fgn_profile_stamp_words': (_i, []),
    'fgn_profile_stamps': (_i, [_p, _i]),
    'fgn_phase_signal': (_i, [_p, _p]),
    'fgn_phase_wait': (_i, [_p, _i, _i, _p]),
    'fgn_conv2d_workspace_bytes': (C.c_size_t, [_i] * 10),
    'fgn_conv2d_kernel_id': (_i, [_i] * 14),
    'fgn_conv2d_nhwc_f32': (_i, [_p] * 8 + [_i] * 13 + [_p, C.c_size_t, _p]),
    'fgn_winograd_input_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_winograd_t_pad': (_i, [_i]),
    'fgn_winograd_gemm_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_winograd4_input_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_winograd4_output_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_winograd_output_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_winograd4_variant': (_i, [_i, _i, _i]),
    'fgn_winograd_pack_weights_f32': (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    'fgn_winograd4_input2_f32': (_i, [_p, _i, _i, _i, _p, _i, _i, _i, _p, _i, _i, _p]),
    'fgn_winograd4_output2_f32': (_i, [_p, _p, _p, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_nchw3_to_nhwc4_f32': (_i, [_p, _p, _i, _i, _i, _p]),
    'fgn_maxpool3x3s2_nhwc_f32': (_i, [_p, _p, _i, _i, _i, _i, _p]),
    'fgn_group_norm_workspace_bytes': (C.c_size_t, [_i] * 4),
    'fgn_group_norm_nhwc_f32': (_i, [_p] * 6 + [C.c_size_t, _i, _i, _i, _i, _f, _i, _p]),
    'fgn_avgpool2x2_nhwc_f32': (_i, [_p, _p, _i, _i, _i, _i, _p]),
    'fgn_roi_align_nhwc_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _i, _p, _i, _p]),
    'fgn_roi_align2_nhwc_f32': (_i, [_p] * 6 + [_i] * 7 + [_f, _i, _i, _p, _i, _p]),
    'fgn_roi_align_mask_u8': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p]),
    'fgn_support_class_vectors_f32': (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    'fgn_scale_channels_f32': (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    'fgn_support_kmean_f32': (_i, [_p, _p, _i, _i, _i, _i, _p]),
    'fgn_gather_support_vectors_f32': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'fgn_relation_gn_head_f32': (_i, [_p] * 10 + [_i, _i, _i, _i, _i, _f, _p, _p, _p]),
    'fgn_relation_gn_head_scratch_bytes': (C.c_size_t, [_i, _i, _i]),
    'fgn_gemm_small_f32': (_i, [_p, _p, _p] + [_i] * 7 + [_p]),
    'fgn_rpn_merge_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'fgn_rpn_proposals_scratch_bytes': (C.c_size_t, [_i, _i, _i]),
    'fgn_rpn_proposals_zeroed_bytes': (C.c_size_t, [_i]),
    'fgn_rpn_proposals_f32': (_i, [_p] * 9 + [_i, _i, _i, _i, _i, _f, _f, C.POINTER(_f), C.POINTER(_f),
                                            _f, _i, _f, _f, _i, _p]),
    'fgn_conv1x1_dual_nhwc_f32': (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_x3_image_bytes': (C.c_size_t, [_i, _i, _i]),
    'fgn_x3_row_tile': (_i, [C.c_longlong, _i, _i, _i, _i]),
    'fgn_conv1x1_x3_nhwc_f32': (_i, [_p] * 7 + [_i] * 7 + [_p]),
    'fgn_conv1x1_dual_x3_nhwc_f32': (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_winograd_gemm_x3_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_h2_image_bytes': (C.c_size_t, [_i, _i, _i]),
    'fgn_h2_row_tile': (_i, [C.c_longlong, _i, _i, _i, _i]),
    'fgn_conv2d_pair_h2_nhwc_f32': (_i, [_p, _i, _i, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_gemm_h2_f32': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_conv1x1_h2_nhwc_f32': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_conv1x1_dual_h2_nhwc_f32': (_i, [_p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_winograd_gemm_h2_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_gemm_x3_f32': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    'fgn_conv2d_pair_nhwc_f32': (_i, [_p, _p, _i, _i, _i, _p, _p, _i, _i, _i, _p, _p, _p] + [_i] * 8 + [_p]),
    'fgn_det_post_scratch_bytes': (C.c_size_t, [_i, _i]),
    'fgn_det_post_f32': (_i, [_p] * 7 + [_i] + [_p] * 3 + [_i, _i, _i, _f, _f, C.POINTER(_f), C.POINTER(_f), _f, _f, _f, _i, _p]),
    'fgn_mask_logits_f32': (_i, [_p, _p, _f, _p, _p, _p, _p, _i, _i, _i, _p]),
    'fgn_mask_paste_u8': (_i, [_p, _p, _i, _p, _p, _i, _i, _i, _i, _f, _i, _p]),
    'fgn_mask_rle': (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _i, _p]),
    'fgn_dense_rle_scratch_bytes': (C.c_size_t, [_i, _i, _i, _i]),
    'fgn_dense_mask_rle': (_i, [_p, _p, C.c_size_t, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    'fgn_rpn_proposals_large_scratch_bytes': (C.c_size_t, [_i, _i, _i]),
    'fgn_rpn_proposals_large_f32': (_i, [_p] * 7 + [_i, _i, _i, _i, _i, _f, _f, C.POINTER(_f), C.POINTER(_f),
                                                  _f, _i, _f, _f, _i, _p]),
    'fgn_box_assign_scratch_bytes': (C.c_size_t, [_i, _i]),
    'fgn_box_assign_f32': (_i, [_p, _i, _p, _p, _i, _i, _f, _f, _f, _i, _p, _p, _p, _p]),
    'fgn_bbox2delta_f32': (_i, [_p, _p, _p, _i, C.POINTER(_f), C.POINTER(_f), _p]),
    'fgn_bce_logits_sum_f32': (_i, [_p, _p, _p, C.c_longlong, _f, C.c_double, _p, _p]),
    'fgn_smooth_l1_sum_f32': (_i, [_p, _p, _p, C.c_longlong, _f, C.c_double, _p, _p]),
    'fgn_softmax_ce_sum_f32': (_i, [_p, _p, _p, _i, _i, C.c_double, _p, _p]),
    'fgn_bn_train_scratch_bytes': (C.c_size_t, [_i]),
    'fgn_bn_train_f32': (_i, [_p, _i, _i, _p, _p, _f, _f, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    'fgn_bce_logits_grad_f32': (_i, [_p, _p, _p, C.c_longlong, _f, _f, _p, _p]),
    'fgn_smooth_l1_grad_f32': (_i, [_p, _p, _p, C.c_longlong, _f, _f, _p, _p]),
    'fgn_softmax_ce_grad_f32': (_i, [_p, _p, _p, _i, _i, _f, _p, _p]),
    'fgn_relu_backward_f32': (_i, [_p, _p, _p, C.c_longlong, _p]),
    'fgn_colsum_scratch_bytes': (C.c_size_t, [_i]),
    'fgn_colsum_f32': (_i, [_p, C.c_longlong, _i, _p, _p, _i, _p]),
    'fgn_bn_train_backward_scratch_bytes': (C.c_size_t, [_i]),
    'fgn_bn_train_backward_f32': (_i, [_p, _p, _p, _p, _p, _p, _f, _i, _i, _p, _p, _p, _p, _p, _p]),
    'fgn_relation_gn_head_backward_f32': (_i, [_p] * 12 + [_i, _i, _i, _i, _i, _f, _p]),
    'fgn_mask_logits_backward_f32': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p]),
    'fgn_im2col3x3_f32': (_i, [_p, _p, _i, _i, _i, _i, _p]),
    'fgn_gemm_tn_workspace_bytes': (C.c_size_t, [_i, _i, _i]),
    'fgn_gemm_tn_f32': (_i, [_p, _p, _p, _i, _i, _i, _p, _p]),
    'fgn_adagrad_step_f32': (_i, [_p, _p, _p, C.c_longlong, _f, _f, _f, _p]),
    'fgn_adagrad_multi_f32': (_i, [_p, _p, _p, _p, _p, _i, _f, _f, _p]),
}

ABI_VERSION = 29
_lib = None


class FgnHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library (once) and bind every symbol of the header."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; it must be the process's HIP runtime (it owns
    # the device memory and streams we are handed), so import torch BEFORE dlopen: our
    # NEEDED libamdhip64.so.7 then resolves to the copy torch already loaded.  Loading ours
    # first leaves two runtimes in the process and every launch fails with hipErrorNoDevice.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise FgnHipError(
            f'{LIB_PATH} is missing: build it with `python -m fgn_amd.build` '
            '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise FgnHipError(f'libfgn_hip.so does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    v = lib.fgn_abi_version()
    if v != ABI_VERSION:
        raise FgnHipError(f'libfgn_hip.so ABI {v} != expected {ABI_VERSION}; rebuild')
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = {-1: 'unsupported shape', -2: 'bad argument'}.get(rc, f'hipError_t {rc}')
        raise FgnHipError(f'{what} failed: {kind}')
