"""ctypes binding of liborienmask_hip.so (C ABI declared in include/orienmask_hip.h).

The HIP library IS the product: there is no CPU or eager-PyTorch fallback.  If the shared
object is missing or does not export a declared symbol, loading raises immediately.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liborienmask_hip.so")

OM_MAX_SCALES = 3
OM_MAX_ANCHORS = 9
OM_STATUS_SPLIT_RANGE = 1      # include/orienmask_hip.h: bits of the forward's status word
OM_STATUS_SK_TIMEOUT = 2


class LayerInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 64),
                ("cin", ctypes.c_int32), ("cout", ctypes.c_int32), ("cout_pad", ctypes.c_int32),
                ("ksize", ctypes.c_int32), ("stride", ctypes.c_int32),
                ("has_bn", ctypes.c_int32), ("leaky", ctypes.c_int32), ("wino_planes", ctypes.c_int32),
                ("w_off", ctypes.c_int64), ("scale_off", ctypes.c_int64), ("shift_off", ctypes.c_int64),
                ("wino_off", ctypes.c_int64), ("wino_alt_off", ctypes.c_int64), ("w16_off", ctypes.c_int64),
                ("wsplit_off", ctypes.c_int64), ("wsplit_scale_off", ctypes.c_int64),
                ("wsplit_direct_off", ctypes.c_int64), ("wsplit_direct_scale_off", ctypes.c_int64)]


class PostCfg(ctypes.Structure):
    _fields_ = [("num_scales", ctypes.c_int32),
                ("grid_h", ctypes.c_int32 * OM_MAX_SCALES), ("grid_w", ctypes.c_int32 * OM_MAX_SCALES),
                ("image_h", ctypes.c_int32), ("image_w", ctypes.c_int32),
                ("anchors_per_scale", ctypes.c_int32),
                ("anchor_w", ctypes.c_float * OM_MAX_ANCHORS), ("anchor_h", ctypes.c_float * OM_MAX_ANCHORS),
                ("anchor_mask", (ctypes.c_int32 * 3) * OM_MAX_SCALES),
                ("num_classes", ctypes.c_int32),
                ("conf_thresh", ctypes.c_float), ("nms_thresh", ctypes.c_float),
                ("nms_pre", ctypes.c_int32), ("nms_post", ctypes.c_int32),
                ("orien_thresh", ctypes.c_float),
                ("bbox_pix_stride", ctypes.c_int32),
                ("nms_semantics", ctypes.c_int32), ("nms_normalized", ctypes.c_int32),
                ("anchors_of_scale", ctypes.c_int32 * OM_MAX_SCALES)]


class RleImage(ctypes.Structure):
    """include/orienmask_hip.h: om_rle_image"""
    _fields_ = [("mask", ctypes.c_void_p), ("K", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("crop_top", ctypes.c_int32), ("crop_down", ctypes.c_int32), ("crop_left", ctypes.c_int32),
                ("crop_right", ctypes.c_int32), ("hflip", ctypes.c_int32), ("vflip", ctypes.c_int32),
                ("orig_h", ctypes.c_int32), ("orig_w", ctypes.c_int32)]


_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# symbol -> (restype, argtypes); must list every function include/orienmask_hip.h declares
SIGNATURES = {
    "om_version": (_i, []),
    "om_last_error": (ctypes.c_char_p, []),
    "om_model_create": (_i, [ctypes.POINTER(_vp), _i, _i]),
    "om_model_create_variant": (_i, [ctypes.POINTER(_vp), _i, _i, _i]),
    "om_model_destroy": (None, [_vp]),
    "om_model_num_layers": (_i, [_vp]),
    "om_model_layer_info": (_i, [_vp, _i, ctypes.POINTER(LayerInfo)]),
    "om_model_weight_floats": (_sz, [_vp]),
    "om_model_load_weights": (_i, [_vp, _vp, _sz, _i]),
    "om_model_weight_split_words": (_sz, [_vp]),
    "om_model_load_weights_split": (_i, [_vp, _vp, _sz]),
    "om_model_set_precision": (_i, [_vp, _i]),
    "om_model_get_precision": (_i, [_vp]),
    "om_model_set_latency_cells": (_i, [_vp, ctypes.c_longlong]),
    "om_model_set_latency_ksplit": (_i, [_vp, _i]),
    "om_model_attach_postprocess": (_i, [_vp, ctypes.POINTER(PostCfg), _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t]),
    "om_model_set_upsample_on_read": (_i, [_vp, _i]),
    "om_conv2d_split": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "om_conv2d_split_k": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "om_conv2d_winograd24_split": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _sz, _vp, _vp]),
    "om_set_wino14_variant": (_i, [_i]),
    "om_wino14_dual_built": (_i, []),
    "om_get_wino14_variant": (_i, []),
    "om_set_stem_fusion": (_i, [_i, _i]),
    "om_get_stem_fusion": (_i, [_i]),
    "om_get_conv3x3_f16_variant": (_i, []),
    "om_set_conv3x3_f16_variant": (_i, [_i]),
    "om_conv2d_wino14_split": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "om_conv2d_wino14_wide_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "om_conv2d_wino14_wide": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _sz, _vp, _vp]),
    "om_set_wino14_wide": (_i, [_i]),
    "om_get_wino14_wide": (_i, []),
    "om_forward_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "om_forward_status_offset": (_sz, [_vp, _i, _i, _i]),
    "om_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "om_layer_tile": (_i, [_vp, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                           ctypes.POINTER(ctypes.c_int)]),
    "om_model_weight_halfs": (_sz, [_vp]),
    "om_model_load_weights_f16": (_i, [_vp, _vp, _sz]),
    "om_forward_f16_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "om_forward_f16": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "om_layer_tile_f16": (_i, [_vp, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                               ctypes.POINTER(ctypes.c_int)]),
    "om_conv2d_f16": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp]),
    "om_conv2d_stem_f16": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "om_conv2d_winograd_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "om_conv2d_winograd": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "om_layer_output_view": (_i, [_vp, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int),
                                  ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "om_model_keep_activations": (_i, [_vp, _i]),
    "om_conv2d_winograd24_scratch_bytes": (_sz, [_i, _i, _i, _i]),
    "om_conv2d_winograd24": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "om_profile_enable": (_i, [_vp, _i]),
    "om_profile_enable_layers": (_i, [_vp, ctypes.c_char_p, _i]),
    "om_profile_read": (_i, [_vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _i,
                             ctypes.POINTER(ctypes.c_int)]),
    "om_conv2d": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp]),
    "om_conv2d_mode": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    "om_conv2d_stem": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "om_conv2d_stem2_split": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "om_conv2d_stem3_split": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "om_conv2d_stem2_f16": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "om_conv2d_split_gather": (_i, [_i, ctypes.POINTER(_vp), ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i), _i, _i, _i,
                                    _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "om_preprocess": (_i, [_vp, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float),
                           _i, _i, _i, _i, _f, _vp, _vp]),
    "om_pad_nchw": (_i, [_vp, ctypes.c_longlong, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "om_postprocess_workspace_bytes": (_sz, [ctypes.POINTER(PostCfg), _i]),
    "om_postprocess": (_i, [ctypes.POINTER(PostCfg), _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "om_postprocess_detect": (_i, [ctypes.POINTER(PostCfg), _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "om_postprocess_assemble": (_i, [ctypes.POINTER(PostCfg), _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "om_postprocess_candidates": (_i, [ctypes.POINTER(PostCfg), _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "om_postprocess_masks": (_i, [ctypes.POINTER(PostCfg), _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "om_recover_bbox": (_i, [_vp, _i, _i, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), _i, _i, _i, _i,
                             _vp, _vp]),
    "om_recover_masks_rle": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "om_recover_masks_rle_strings": (_i, [_vp, _i, _vp, _i, _vp, _vp, ctypes.c_longlong, _vp, _vp, _vp, _vp]),
    "om_post_kernel_occupancy": (_i, [_i, ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "om_nms_workspace_bytes": (_sz, [_i]),
    "om_nms": (_i, [_vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "om_nms_ex": (_i, [_vp, _i, _f, _i, _vp, _vp, _vp, _sz, _vp]),
    "om_ref_math": (_i, [_vp, ctypes.c_longlong, _i, _i, _vp, _vp]),
}

_lib = None


class OrienMaskHipError(RuntimeError):
    pass


def load():
    """Load the HIP library once; raise loudly if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrienMaskHipError(
            "%s not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C orienmask_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    # torch first: its wheel bundles a HIP runtime, and the process must end up with ONE runtime -- the one whose streams and device
    # pointers the callers hand to this library.  Loaded the other way round (this library before torch) the library binds to
    # /opt/rocm's runtime, torch initialises its own, and every launch fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise OrienMaskHipError("%s does not export %s" % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().om_last_error()
        raise OrienMaskHipError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def require_cuda_tensor(t, name, dtype=None):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise OrienMaskHipError("%s must be a tensor on an MI355X device (got %s); this path has no CPU fallback"
                                % (name, getattr(t, "device", type(t))))
    if dtype is not None and t.dtype != dtype:
        raise OrienMaskHipError("%s must be %s, got %s" % (name, dtype, t.dtype))


def current_stream_ptr(device):
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
