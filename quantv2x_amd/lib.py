"""ctypes binding of ``libqv2x.so`` (C ABI in ``include/qv2x.h``).

The library is the product: there is no fallback.  ``load()`` raises if the shared object is missing or
does not export every symbol the header declares.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqv2x.so")
MAX_GROUPS = 4
ABI_VERSION = 6             # include/qv2x.h: QV2X_ABI_VERSION (the structs below mirror that header)

SYMBOLS = [
    "qv2x_last_error", "qv2x_version", "qv2x_fill_i8", "qv2x_pfn_scatter_i8", "qv2x_pfn_unscatter_i8", "qv2x_conv3x3_i8",
    "qv2x_conv3x3_i8_wide_ok", "qv2x_conv3x3_i8_pack_wide", "qv2x_conv3x3_i8_wide", "qv2x_conv3x3_i8_chain64",
    "qv2x_deconv_i8", "qv2x_deconv_i8_batch", "qv2x_codebook_level_floats", "qv2x_codebook_c2_f32", "qv2x_codebook_encode_f32", "qv2x_codebook_encode_wave_f32", "qv2x_fuse_att_f32", "qv2x_fuse_att_batch_f32", "qv2x_fuse_heads_batch_f32",
    "qv2x_decode_lut_f32", "qv2x_single_heads_lut_f32", "qv2x_table_heads_f32", "qv2x_dequant_i8_f32", "qv2x_heads_f32", "qv2x_decode_heads_f32", "qv2x_heads_pair_f32", "qv2x_voxelize_workspace_bytes", "qv2x_voxelize_f32",
    "qv2x_postprocess_workspace_bytes", "qv2x_postprocess_f32", "qv2x_postprocess_late_workspace_bytes", "qv2x_postprocess_late_f32",
    "qv2x_conv3x3_f32", "qv2x_deconv_f32", "qv2x_pfn_scatter_f32", "qv2x_codebook_encode_f32in",
    "qv2x_pyramid_weighted_fuse_f32", "qv2x_pyramid_weighted_fuse_i8", "qv2x_conv1x1_i8", "qv2x_gconv3x3_i8", "qv2x_conv3x3_i8_res",
    "qv2x_deconv_f32in", "qv2x_codebook_decode_f32", "qv2x_occ_score_i8",
    "qv2x_codebook64_level_floats", "qv2x_codebook64_c2_f32", "qv2x_codebook_encode64_f32", "qv2x_codebook_encode64_f32in",
    "qv2x_add_relu_f32", "qv2x_occ_sigmoid_f32", "qv2x_pyramid_weighted_fuse_f32p", "qv2x_bottleneck_i8",
    "qv2x_comm_unique_id", "qv2x_comm_init", "qv2x_comm_destroy", "qv2x_allgather_codes", "qv2x_pairwise_from_poses_f64", "qv2x_pairwise_from_poses_batch_f64",
    "qv2x_codebook_encode_collapsed_f32", "qv2x_codebook_encode_candidates_i8", "qv2x_codebook_encode_listed_f32", "qv2x_mean_vfe_f32", "qv2x_sp_index_scatter", "qv2x_sp_out_sites_workspace_bytes", "qv2x_sp_out_sites", "qv2x_sp_rulebook", "qv2x_sp_conv_f32in", "qv2x_sp_conv_i8", "qv2x_sp_to_bev_i8",
]
COMM_ID_BYTES = 128


class PfnParams(C.Structure):
    _fields_ = [("w", C.c_float * 640), ("b", C.c_float * 64),
                ("d1", C.c_float), ("z1", C.c_float), ("d2", C.c_float), ("z2", C.c_float),
                ("vox", C.c_float * 3), ("off", C.c_float * 3)]


class ConvDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin_total", C.c_int32),
                ("stride", C.c_int32), ("cout", C.c_int32), ("ngroups", C.c_int32),
                ("group_c0", C.c_int32 * MAX_GROUPS), ("group_c", C.c_int32 * MAX_GROUPS),
                ("group_zx", C.c_int32 * MAX_GROUPS),
                ("out_ctotal", C.c_int32), ("out_c0", C.c_int32), ("relu", C.c_int32),
                ("out_delta", C.c_float), ("out_zp", C.c_float)]


class ChainDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32),
                ("depth", C.c_int32), ("stride0", C.c_int32), ("relu", C.c_int32),
                ("out_delta", C.c_float * 4), ("out_zp", C.c_float * 4)]


class DeconvDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("s", C.c_int32), ("in_zx", C.c_int32), ("in_delta", C.c_float),
                ("out_ctotal", C.c_int32), ("out_c0", C.c_int32), ("relu", C.c_int32),
                ("out_delta", C.c_float), ("out_zp", C.c_float), ("out_h", C.c_int32), ("out_w", C.c_int32)]


class F32ConvDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin_total", C.c_int32), ("cin0", C.c_int32), ("cin", C.c_int32),
                ("stride", C.c_int32), ("cout", C.c_int32), ("out_ctotal", C.c_int32), ("out_c0", C.c_int32), ("relu", C.c_int32)]


class Conv1x1Desc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("stride", C.c_int32),
                ("mode", C.c_int32), ("relu", C.c_int32), ("out_ctotal", C.c_int32), ("out_c0", C.c_int32),
                ("out_delta", C.c_float), ("out_zp", C.c_float), ("res_zx", C.c_int32), ("res_delta", C.c_float)]


class BottleneckDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("planes", C.c_int32), ("width", C.c_int32), ("cg", C.c_int32),
                ("in_zx", C.c_int32), ("in_delta", C.c_float), ("delta1", C.c_float), ("zp1", C.c_float), ("delta2", C.c_float), ("zp2", C.c_float),
                ("out_delta", C.c_float), ("out_zp", C.c_float)]


class GconvDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32), ("cg", C.c_int32), ("stride", C.c_int32),
                ("relu", C.c_int32), ("out_delta", C.c_float), ("out_zp", C.c_float)]


class SpconvDesc(C.Structure):
    _fields_ = [("subm", C.c_int32), ("k", C.c_int32 * 3), ("s", C.c_int32 * 3), ("p", C.c_int32 * 3),
                ("in_shape", C.c_int32 * 3), ("out_shape", C.c_int32 * 3), ("agents", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("cap_in", C.c_int32), ("cap_out", C.c_int32), ("out_delta", C.c_float), ("out_zp", C.c_float)]


class OccDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32), ("aw", C.c_int32), ("corr", C.c_int32),
                ("scale", C.c_float), ("bias", C.c_float), ("out_delta", C.c_float), ("out_zp", C.c_float)]


class EncodeDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("levels", C.c_int32), ("kc", C.c_int32),
                ("in_zx", C.c_int32), ("in_delta", C.c_float), ("segs", C.c_int32)]


class FuseDesc(C.Structure):
    _fields_ = [("agents", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("levels", C.c_int32), ("kc", C.c_int32),
                ("max_cav", C.c_int32), ("ego", C.c_int32),
                ("code_agent_stride", C.c_int64), ("code_level_stride", C.c_int64),
                ("h_metres", C.c_double), ("w_metres", C.c_double), ("discrete_ratio", C.c_double), ("fusion", C.c_int32)]


class PostprocessDesc(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("anchors_per_cell", C.c_int32), ("num_bins", C.c_int32),
                ("score_threshold", C.c_float), ("nms_threshold", C.c_float), ("dir_offset", C.c_float),
                ("range", C.c_float * 6), ("transform", C.c_float * 16), ("max_boxes", C.c_int32),
                ("num_classes", C.c_int32), ("range_xy_only", C.c_int32),
                ("max_extent", C.c_float), ("z_min", C.c_float), ("z_max", C.c_float)]


class Qv2xError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Qv2xError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the deployed path.")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise Qv2xError(f"libqv2x.so does not export {missing}")
    vp = C.c_void_p
    lib.qv2x_last_error.restype = C.c_char_p
    lib.qv2x_version.restype = C.c_int
    if lib.qv2x_version() != ABI_VERSION:
        raise Qv2xError(f"libqv2x.so has ABI {lib.qv2x_version()}, this binding mirrors ABI {ABI_VERSION} (include/qv2x.h): rebuild the library")
    lib.qv2x_fill_i8.argtypes = [vp, C.c_int64, C.c_int, vp]
    lib.qv2x_pfn_scatter_i8.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(PfnParams), vp, C.c_int, C.c_int, C.c_int, vp]
    lib.qv2x_pfn_unscatter_i8.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.qv2x_conv3x3_i8.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_conv3x3_i8_wide.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_conv3x3_i8_chain64.argtypes = [C.POINTER(ChainDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_conv3x3_i8_pack_wide.argtypes = [C.POINTER(ConvDesc), vp, vp, vp]
    lib.qv2x_conv3x3_i8_wide_ok.argtypes = [C.POINTER(ConvDesc)]
    lib.qv2x_deconv_i8.argtypes = [C.POINTER(DeconvDesc), vp, vp, vp, vp, vp]
    lib.qv2x_deconv_i8_batch.argtypes = [C.POINTER(DeconvDesc), C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp]
    lib.qv2x_codebook_level_floats.argtypes = [C.c_int]
    lib.qv2x_codebook_level_floats.restype = C.c_int64
    lib.qv2x_codebook_c2_f32.argtypes = [vp, C.c_int, vp, vp]
    lib.qv2x_codebook_encode_f32.argtypes = [C.POINTER(EncodeDesc), vp, C.POINTER(vp), vp, vp]
    lib.qv2x_codebook_encode_wave_f32.argtypes = [C.POINTER(EncodeDesc), vp, vp, C.POINTER(vp), vp, vp]
    lib.qv2x_fuse_att_f32.argtypes = [C.POINTER(FuseDesc), vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_fuse_att_batch_f32.argtypes = [C.POINTER(FuseDesc), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32), vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_decode_lut_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.qv2x_fuse_heads_batch_f32.argtypes = [C.POINTER(FuseDesc), C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32), vp, vp, vp, vp, vp,
                                              C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_heads_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.qv2x_decode_heads_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.qv2x_table_heads_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_single_heads_lut_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.qv2x_heads_pair_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp,
                                        vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]
    lib.qv2x_dequant_i8_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
    lib.qv2x_codebook_encode_collapsed_f32.argtypes = [C.POINTER(EncodeDesc), vp, vp, vp, vp, vp, vp]
    lib.qv2x_codebook_encode_candidates_i8.argtypes = [C.POINTER(EncodeDesc), vp, vp, vp, vp, C.POINTER(C.c_float), vp, vp, vp, vp]
    lib.qv2x_codebook_encode_listed_f32.argtypes = [C.POINTER(EncodeDesc), vp, C.POINTER(vp), vp, vp, vp, vp]
    lib.qv2x_mean_vfe_f32.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
    lib.qv2x_sp_index_scatter.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]
    lib.qv2x_sp_out_sites_workspace_bytes.argtypes = [C.POINTER(SpconvDesc)]
    lib.qv2x_sp_out_sites_workspace_bytes.restype = C.c_int64
    lib.qv2x_sp_out_sites.argtypes = [C.POINTER(SpconvDesc), vp, vp, vp, vp, vp, vp, C.c_int64, vp]
    lib.qv2x_sp_rulebook.argtypes = [C.POINTER(SpconvDesc), vp, vp, vp, vp, vp]
    lib.qv2x_sp_conv_f32in.argtypes = [C.POINTER(SpconvDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_sp_conv_i8.argtypes = [C.POINTER(SpconvDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_sp_to_bev_i8.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.qv2x_voxelize_workspace_bytes.argtypes = [C.c_int]
    lib.qv2x_voxelize_workspace_bytes.restype = C.c_int64
    lib.qv2x_voxelize_f32.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_int,
                                      vp, C.c_int64, vp, vp, vp, vp, vp]
    lib.qv2x_postprocess_workspace_bytes.argtypes = [C.POINTER(PostprocessDesc)]
    lib.qv2x_postprocess_workspace_bytes.restype = C.c_int64
    lib.qv2x_postprocess_f32.argtypes = [C.POINTER(PostprocessDesc), vp, vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    lib.qv2x_postprocess_late_workspace_bytes.argtypes = [C.POINTER(PostprocessDesc), C.c_int]
    lib.qv2x_postprocess_late_workspace_bytes.restype = C.c_int64
    lib.qv2x_postprocess_late_f32.argtypes = [C.POINTER(PostprocessDesc), C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                              C.POINTER(C.c_float), vp, C.c_int64, vp, vp, vp, vp, vp]
    lib.qv2x_conv3x3_f32.argtypes = [C.POINTER(F32ConvDesc), vp, vp, vp, vp, vp]
    lib.qv2x_deconv_f32.argtypes = [C.POINTER(F32ConvDesc), vp, vp, vp, vp, vp]
    lib.qv2x_pfn_scatter_f32.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                         C.POINTER(C.c_float), vp, C.c_int, C.c_int, C.c_int, vp]
    lib.qv2x_codebook_encode_f32in.argtypes = [C.POINTER(EncodeDesc), vp, C.POINTER(vp), vp, vp]
    lib.qv2x_pyramid_weighted_fuse_f32.argtypes = [C.POINTER(FuseDesc), C.c_int, vp, vp, vp, vp, vp]
    lib.qv2x_pyramid_weighted_fuse_i8.argtypes = [C.POINTER(FuseDesc), C.c_int, vp, C.c_int, C.c_float, vp, vp, vp, vp]
    lib.qv2x_conv1x1_i8.argtypes = [C.POINTER(Conv1x1Desc), vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_gconv3x3_i8.argtypes = [C.POINTER(GconvDesc), vp, vp, vp, vp, vp, vp, vp, vp]
    lib.qv2x_conv3x3_i8_res.argtypes = [C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_float, vp, vp]
    lib.qv2x_deconv_f32in.argtypes = [C.POINTER(DeconvDesc), vp, vp, vp, vp, vp]
    lib.qv2x_codebook_decode_f32.argtypes = [vp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]
    lib.qv2x_occ_score_i8.argtypes = [C.POINTER(OccDesc), vp, vp, vp, vp, vp, vp]
    lib.qv2x_codebook64_level_floats.argtypes = [C.c_int]
    lib.qv2x_codebook64_level_floats.restype = C.c_int64
    lib.qv2x_codebook64_c2_f32.argtypes = [vp, C.c_int, vp, vp]
    lib.qv2x_codebook_encode64_f32.argtypes = [C.POINTER(EncodeDesc), C.c_int, vp, C.POINTER(vp), vp, vp]
    lib.qv2x_codebook_encode64_f32in.argtypes = [C.POINTER(EncodeDesc), C.c_int, vp, C.POINTER(vp), vp, vp]
    lib.qv2x_add_relu_f32.argtypes = [vp, vp, vp, C.c_int64, vp]
    lib.qv2x_occ_sigmoid_f32.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    lib.qv2x_pyramid_weighted_fuse_f32p.argtypes = [C.POINTER(FuseDesc), C.c_int, vp, vp, vp, vp, vp]
    lib.qv2x_bottleneck_i8.argtypes = [C.POINTER(BottleneckDesc), vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp, vp]
    lib.qv2x_comm_unique_id.argtypes = [vp]
    lib.qv2x_comm_init.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.qv2x_comm_destroy.argtypes = [vp]
    lib.qv2x_allgather_codes.argtypes = [vp, vp, vp, C.c_int64, vp]
    lib.qv2x_pairwise_from_poses_f64.argtypes = [vp, C.c_int, C.c_int64, C.c_int64, C.c_int, vp, vp]
    lib.qv2x_pairwise_from_poses_batch_f64.argtypes = [vp, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_int, vp, vp]
    for s in SYMBOLS:
        if s not in ("qv2x_last_error", "qv2x_codebook_level_floats", "qv2x_codebook64_level_floats", "qv2x_voxelize_workspace_bytes", "qv2x_postprocess_workspace_bytes",
                     "qv2x_postprocess_late_workspace_bytes", "qv2x_sp_out_sites_workspace_bytes"):
            getattr(lib, s).restype = C.c_int
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().qv2x_last_error().decode(errors="replace")
        raise Qv2xError(f"{what or 'libqv2x'} failed with {rc}: {msg}")


def ptr(t) -> C.c_void_p:
    """Device (or host) address of a torch tensor / numpy array; None -> NULL."""
    if t is None:
        return C.c_void_p(0)
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)


def current_stream() -> C.c_void_p:
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
