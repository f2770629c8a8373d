"""ctypes binding of the C ABI declared in include/ml3d_hip.h.

``get()`` loads the in-tree ``lib/libml3d_hip.so`` (built by ``__graft_entry__.build()`` with
hipcc for gfx950) and FAILS LOUDLY if it is missing — there is no CPU fallback in this
package.  The raw functions below take integer device addresses so the same signatures
serve the torch front end (``ml3d.ops``) and the ABI tests.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libml3d_hip.so")
MAX_LAYERS = 8
_lib = None

ERRORS = {-1: "invalid argument", -2: "workspace too small", -3: "HIP launch failed", -4: "unsupported configuration"}

ABI_VERSION = 12         # ML3D_ABI_VERSION of include/ml3d_hip.h (checked against the loaded library in get())

# every symbol include/ml3d_hip.h declares (checked by tests/test_abi_symbols.py)
SYMBOLS = [
    "ml3d_abi_version",
    "ml3d_knn_workspace_bytes",
    "ml3d_knn_search",
    "ml3d_radius_workspace_bytes",
    "ml3d_radius_count",
    "ml3d_radius_fill",
    "ml3d_radius_dense_workspace_bytes",
    "ml3d_radius_dense_gather",
    "ml3d_radius_dense_expand",
    "ml3d_ragged_to_dense",
    "ml3d_voxelize_workspace_bytes",
    "ml3d_voxelize_count",
    "ml3d_voxelize_fill",
    "ml3d_subsample_workspace_bytes",
    "ml3d_subsample_count",
    "ml3d_subsample_fill",
    "ml3d_subsample_items_max_points",
    "ml3d_subsample_items_count",
    "ml3d_subsample_items_fill",
    "ml3d_rotate_points",
    "ml3d_kpconv_batch_workspace_bytes", "ml3d_kpconv_batch_host_scratch_bytes", "ml3d_kpconv_batch_build",
    "ml3d_kpconv_workspace_bytes",
    "ml3d_kpconv_rigid", "ml3d_kpconv_deformable", "ml3d_kpconv_weighted", "ml3d_kpconv_weighted_backward",
    "ml3d_linear_workspace_bytes",
    "ml3d_linear",
    "ml3d_gather_pool",
    "ml3d_pillar_features_workspace_bytes",
    "ml3d_pillar_features",
    "ml3d_conv2d_workspace_bytes",
    "ml3d_conv2d_nhwc",
    "ml3d_gemm_pack_bf16x3_bytes",
    "ml3d_gemm_pack_bf16x3",
    "ml3d_conv2d_nhwc_bf16x3",
    "ml3d_linear_bf16x3",
    "ml3d_linear_bf16x3_gathered",
    "ml3d_linear_bf16x3_workspace_bytes",
    "ml3d_kpconv_rigid_bf16x3",
    "ml3d_deconv2d_nhwc_bf16x3",
    "ml3d_deconv2d_nhwc",
    "ml3d_nhwc_to_nchw",
    "ml3d_nms_workspace_bytes",
    "ml3d_nms",
    "ml3d_pp_anchor_scores",
    "ml3d_topk_rows_workspace_bytes",
    "ml3d_topk_rows",
    "ml3d_pp_boxes_workspace_bytes",
    "ml3d_pp_boxes",
    "ml3d_iou_bev",
    "ml3d_iou_3d",
    "ml3d_nearest_to_center_workspace_bytes",
    "ml3d_nearest_to_center",
    "ml3d_nearest_to_center_dev",
    "ml3d_patch_crop",
    "ml3d_patch_recenter",
    "ml3d_vote_update",
    "ml3d_randla_gather_max",
    "ml3d_randla_gather_max_backward",
    "ml3d_randla_attentive_pool",
    "ml3d_randla_attentive_pool_backward",
    "ml3d_argmax_labels",
    "ml3d_randla_pyramid_workspace_bytes",
    "ml3d_randla_knn_pyramid",
    "ml3d_randla_param_layout",
    "ml3d_randla_forward_workspace_bytes",
    "ml3d_randla_forward",
    "ml3d_randla_forward_traced",
    "ml3d_randla_knn_pyramid_traced",
    "ml3d_randla_knn_pyramid_ordered",
    "ml3d_randla_forward_ordered",
    "ml3d_gemm_tn",
    "ml3d_batchnorm_train_workspace_bytes", "ml3d_batchnorm_train_forward", "ml3d_batchnorm_train_backward",
    "ml3d_gather_rows", "ml3d_scatter_add_rows", "ml3d_gather_pool_backward",
    "ml3d_kpconv_deformed_weighted", "ml3d_kpconv_deformed_weighted_backward",
    "ml3d_kpconv_offset_regulariser_blocks", "ml3d_kpconv_offset_regulariser",
    "ml3d_randla_attention_stage", "ml3d_randla_attention_stage_backward_workspace_bytes", "ml3d_randla_attention_stage_backward",
]


class RandlaDesc(C.Structure):
    _fields_ = [
        ("num_layers", C.c_int32),
        ("in_channels", C.c_int32),
        ("dim_features", C.c_int32),
        ("num_classes", C.c_int32),
        ("num_neighbors", C.c_int32),
        ("dim_output", C.c_int32 * MAX_LAYERS),
        ("sub_sampling_ratio", C.c_int32 * MAX_LAYERS),
        ("batch", C.c_int64),
        ("num_points", C.c_int64),
    ]


class Trace(C.Structure):
    pass


Trace._fields_ = [("tag", C.c_int32), ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p), ("next", C.POINTER(Trace))]


def bind(lib):
    """Attach argtypes/restypes to a loaded library object."""
    vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t
    lib.ml3d_abi_version.restype = C.c_int
    lib.ml3d_knn_workspace_bytes.restype = sz
    lib.ml3d_knn_workspace_bytes.argtypes = [i64, i64, i64]
    lib.ml3d_knn_search.restype = C.c_int
    lib.ml3d_knn_search.argtypes = [vp, vp, vp, vp, i64, i64, i64, i32, i32, vp, vp, vp, sz, vp]
    f32 = C.c_float
    lib.ml3d_radius_workspace_bytes.restype = sz
    lib.ml3d_radius_workspace_bytes.argtypes = [i64, i64, i64, i64]
    lib.ml3d_radius_count.restype = C.c_int
    lib.ml3d_radius_count.argtypes = [vp, vp, vp, vp, i64, i64, i64, f32, vp, vp, vp, sz, vp]
    lib.ml3d_radius_fill.restype = C.c_int
    lib.ml3d_radius_fill.argtypes = [vp, vp, vp, vp, i64, i64, i64, f32, vp, i64, i32, i64, C.c_int32, vp, vp, vp, sz, vp, sz, vp]
    lib.ml3d_radius_dense_workspace_bytes.restype = sz
    lib.ml3d_radius_dense_workspace_bytes.argtypes = [i64, i64, i64, i32]
    lib.ml3d_radius_dense_gather.restype = C.c_int
    lib.ml3d_radius_dense_gather.argtypes = [vp, vp, vp, vp, i64, i64, i64, f32, i32, i32, vp, vp, sz, vp]
    lib.ml3d_radius_dense_expand.restype = C.c_int
    lib.ml3d_radius_dense_expand.argtypes = [i64, i64, i64, i32, i64, C.c_int32, vp, vp, sz, vp]
    lib.ml3d_ragged_to_dense.restype = C.c_int
    lib.ml3d_ragged_to_dense.argtypes = [vp, vp, i64, i64, i64, vp, vp, vp]
    lib.ml3d_voxelize_workspace_bytes.restype = sz
    lib.ml3d_voxelize_workspace_bytes.argtypes = [i64, i64]
    lib.ml3d_voxelize_count.restype = C.c_int
    lib.ml3d_voxelize_count.argtypes = [vp, i64, vp, i64, i64, vp, vp, vp, i64, i64, vp, vp, vp, sz, vp]
    lib.ml3d_voxelize_fill.restype = C.c_int
    lib.ml3d_voxelize_fill.argtypes = [i64, i64, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, sz, vp]
    lib.ml3d_subsample_workspace_bytes.restype = sz
    lib.ml3d_subsample_workspace_bytes.argtypes = [i64, i64]
    lib.ml3d_subsample_count.restype = C.c_int
    lib.ml3d_subsample_count.argtypes = [vp, vp, i64, i64, f32, vp, vp, vp, sz, vp]
    lib.ml3d_subsample_fill.restype = C.c_int
    lib.ml3d_subsample_fill.argtypes = [vp, vp, i64, vp, i64, i64, vp, vp, vp, vp, sz, vp]
    lib.ml3d_subsample_items_max_points.restype = i64
    lib.ml3d_subsample_items_max_points.argtypes = []
    lib.ml3d_subsample_items_count.restype = C.c_int
    lib.ml3d_subsample_items_count.argtypes = [vp, vp, i64, i64, f32, i64, vp, vp, vp]
    lib.ml3d_subsample_items_fill.restype = C.c_int
    lib.ml3d_subsample_items_fill.argtypes = [vp, vp, i64, i64, f32, i64, vp, vp, vp]
    lib.ml3d_rotate_points.restype = C.c_int
    lib.ml3d_rotate_points.argtypes = [vp, vp, i64, i64, vp, i32, vp, vp]
    lib.ml3d_kpconv_batch_workspace_bytes.restype = sz
    lib.ml3d_kpconv_batch_workspace_bytes.argtypes = [i64, i64, i32, i32]
    lib.ml3d_kpconv_batch_host_scratch_bytes.restype = sz
    lib.ml3d_kpconv_batch_host_scratch_bytes.argtypes = [i64, i32]
    lib.ml3d_kpconv_batch_build.restype = C.c_int
    lib.ml3d_kpconv_batch_build.argtypes = [vp, vp, i64, i64, vp, vp, vp, sz, vp, vp, vp, sz, vp, sz, vp]
    lib.ml3d_kpconv_workspace_bytes.restype = sz
    lib.ml3d_kpconv_workspace_bytes.argtypes = [i64, i32, i32, i32]
    lib.ml3d_kpconv_rigid.restype = C.c_int
    lib.ml3d_kpconv_rigid.argtypes = [vp, vp, vp, i64, i64, i64, vp, i32, vp, i32, f32, i32, vp, vp, i32, f32, i32, vp,
                                      vp, sz, vp]
    lib.ml3d_kpconv_rigid_bf16x3.restype = C.c_int
    lib.ml3d_kpconv_rigid_bf16x3.argtypes = [vp, vp, vp, i64, i64, i64, vp, i32, vp, i32, f32, i32, vp, vp, vp, i32, f32, i32, vp,
                                             vp, sz, vp]
    lib.ml3d_kpconv_weighted.restype = C.c_int
    lib.ml3d_kpconv_weighted.argtypes = [vp, vp, vp, i64, i64, i64, vp, i32, vp, i32, f32, i32, vp, vp]
    lib.ml3d_kpconv_weighted_backward.restype = C.c_int
    lib.ml3d_kpconv_weighted_backward.argtypes = [vp, vp, vp, i64, i64, i64, i32, vp, i32, f32, i32, vp, vp, vp]
    lib.ml3d_kpconv_deformable.restype = C.c_int
    lib.ml3d_kpconv_deformable.argtypes = [vp, vp, vp, i64, i64, i64, vp, i32, vp, i32, f32, i32, vp, i32, vp, vp, i32, f32,
                                           i32, vp, vp, sz, vp]
    lib.ml3d_linear_workspace_bytes.restype = sz
    lib.ml3d_linear_workspace_bytes.argtypes = [i64, i32, i32]
    lib.ml3d_linear.restype = C.c_int
    lib.ml3d_linear.argtypes = [vp, i64, i32, vp, i64, i64, vp, i64, i32, vp, vp, vp, i64, vp, i64, i64, i32, f32, vp, i64,
                                i64, i32, vp, sz, vp]
    lib.ml3d_gather_pool.restype = C.c_int
    lib.ml3d_gather_pool.argtypes = [vp, i64, i32, vp, i64, i64, i32, vp, vp]
    lib.ml3d_pillar_features_workspace_bytes.restype = sz
    lib.ml3d_pillar_features_workspace_bytes.argtypes = [i64, i32, i32, vp]
    lib.ml3d_pillar_features.restype = C.c_int
    lib.ml3d_pillar_features.argtypes = [vp, i64, i32, vp, vp, vp, vp, i64, i64, i32, f32, f32, f32, f32, i32, i32, i32,
                                         vp, vp, vp, vp, i32, vp, sz, vp]
    lib.ml3d_conv2d_workspace_bytes.restype = sz
    lib.ml3d_conv2d_workspace_bytes.argtypes = [i64, i32, i32, i32, i32, i32, i32]
    lib.ml3d_conv2d_nhwc.restype = C.c_int
    lib.ml3d_conv2d_nhwc.argtypes = [vp, i64, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp, i64, vp, sz, vp]
    lib.ml3d_gemm_pack_bf16x3_bytes.restype = sz
    lib.ml3d_gemm_pack_bf16x3_bytes.argtypes = [i32, i32]
    lib.ml3d_gemm_pack_bf16x3.restype = C.c_int
    lib.ml3d_gemm_pack_bf16x3.argtypes = [vp, i32, i32, vp, sz, vp]
    lib.ml3d_conv2d_nhwc_bf16x3.restype = C.c_int
    lib.ml3d_conv2d_nhwc_bf16x3.argtypes = [vp, i64, i32, i32, i32, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp, i64, vp]
    lib.ml3d_linear_bf16x3.restype = C.c_int
    lib.ml3d_linear_bf16x3.argtypes = [vp, i64, i32, vp, i64, i32, i64, vp, vp, vp, i64, i32, i32, f32, vp, i64, vp, sz, vp]
    lib.ml3d_linear_bf16x3_gathered.restype = C.c_int
    lib.ml3d_linear_bf16x3_gathered.argtypes = [vp, i64, i32, vp, i64, i32, i64, vp, vp, vp, i64, vp, i64, i64, i32, i32, f32, vp, i64, vp, sz, vp]
    lib.ml3d_linear_bf16x3_workspace_bytes.restype = sz
    lib.ml3d_linear_bf16x3_workspace_bytes.argtypes = [i64, i32, i32]
    lib.ml3d_deconv2d_nhwc_bf16x3.restype = C.c_int
    lib.ml3d_deconv2d_nhwc_bf16x3.argtypes = [vp, i64, i32, i32, i32, vp, vp, i32, i32, f32, i32, vp, i64, vp]
    lib.ml3d_deconv2d_nhwc.restype = C.c_int
    lib.ml3d_deconv2d_nhwc.argtypes = [vp, i64, i32, i32, i32, vp, vp, i32, i32, f32, i32, vp, i64, vp, sz, vp]
    lib.ml3d_nhwc_to_nchw.restype = C.c_int
    lib.ml3d_nhwc_to_nchw.argtypes = [vp, i64, i32, i32, i64, i64, vp, vp]
    lib.ml3d_nms_workspace_bytes.restype = sz
    lib.ml3d_nms_workspace_bytes.argtypes = [i64]
    lib.ml3d_nms.restype = C.c_int
    lib.ml3d_nms.argtypes = [vp, vp, i64, f32, vp, vp, vp, sz, vp]
    lib.ml3d_pp_anchor_scores.restype = C.c_int
    lib.ml3d_pp_anchor_scores.argtypes = [vp, vp, i64, i32, i32, i64, vp, vp]
    lib.ml3d_topk_rows_workspace_bytes.restype = sz
    lib.ml3d_topk_rows_workspace_bytes.argtypes = [i64, i64, i64]
    lib.ml3d_topk_rows.restype = C.c_int
    lib.ml3d_topk_rows.argtypes = [vp, i64, i64, i64, vp, vp, vp, sz, vp]
    lib.ml3d_pp_boxes_workspace_bytes.restype = sz
    lib.ml3d_pp_boxes_workspace_bytes.argtypes = [i64, i64, i32]
    lib.ml3d_pp_boxes.restype = C.c_int
    lib.ml3d_pp_boxes.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i64, f32, f32, f32, vp, vp, vp, sz, vp]
    lib.ml3d_iou_bev.restype = C.c_int
    lib.ml3d_iou_bev.argtypes = [vp, vp, i64, i64, vp, vp]
    lib.ml3d_iou_3d.restype = C.c_int
    lib.ml3d_iou_3d.argtypes = [vp, vp, i64, i64, vp, vp]
    lib.ml3d_nearest_to_center_workspace_bytes.restype = sz
    lib.ml3d_nearest_to_center_workspace_bytes.argtypes = [i64]
    lib.ml3d_nearest_to_center.restype = C.c_int
    lib.ml3d_nearest_to_center.argtypes = [vp, i64, vp, i64, vp, vp, vp, sz, vp]
    lib.ml3d_nearest_to_center_dev.restype = C.c_int
    lib.ml3d_nearest_to_center_dev.argtypes = [vp, i64, vp, i64, vp, vp, vp, sz, vp]
    lib.ml3d_patch_crop.restype = C.c_int
    lib.ml3d_patch_crop.argtypes = [vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, sz, vp]
    lib.ml3d_patch_recenter.restype = C.c_int
    lib.ml3d_patch_recenter.argtypes = [vp, i64, i32, vp, i32, f32, f32, vp, vp, sz, vp]
    lib.ml3d_randla_gather_max.restype = C.c_int
    lib.ml3d_randla_gather_max.argtypes = [vp, vp, i64, i64, i64, i32, vp, vp]
    lib.ml3d_randla_gather_max_backward.restype = C.c_int
    lib.ml3d_randla_gather_max_backward.argtypes = [vp, vp, vp, i64, i64, i64, i32, vp, vp]
    lib.ml3d_randla_attentive_pool.restype = C.c_int
    lib.ml3d_randla_attentive_pool.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    lib.ml3d_randla_attentive_pool_backward.restype = C.c_int
    lib.ml3d_randla_attentive_pool_backward.argtypes = [vp, vp, vp, vp, i64, i32, i32, vp, vp, vp]
    lib.ml3d_gemm_tn.restype = C.c_int
    lib.ml3d_gemm_tn.argtypes = [vp, i64, vp, i64, i64, i32, i32, vp, i64, vp, vp]
    lib.ml3d_batchnorm_train_workspace_bytes.restype = sz
    lib.ml3d_batchnorm_train_workspace_bytes.argtypes = [i32]
    lib.ml3d_batchnorm_train_forward.restype = C.c_int
    lib.ml3d_batchnorm_train_forward.argtypes = [vp, i64, i32, vp, vp, f32, i32, f32, vp, vp, vp, vp, vp, sz, vp]
    lib.ml3d_batchnorm_train_backward.restype = C.c_int
    lib.ml3d_batchnorm_train_backward.argtypes = [vp, vp, vp, i64, i32, vp, vp, vp, i32, f32, vp, vp, vp, vp, sz, vp]
    lib.ml3d_gather_rows.restype = C.c_int
    lib.ml3d_gather_rows.argtypes = [vp, i64, i32, vp, i64, i64, vp, vp]
    lib.ml3d_scatter_add_rows.restype = C.c_int
    lib.ml3d_scatter_add_rows.argtypes = [vp, i64, i32, vp, i64, i64, vp, vp]
    lib.ml3d_gather_pool_backward.restype = C.c_int
    lib.ml3d_gather_pool_backward.argtypes = [vp, i64, i32, vp, i64, i64, i32, vp, vp, vp]
    lib.ml3d_kpconv_deformed_weighted.restype = C.c_int
    lib.ml3d_kpconv_deformed_weighted.argtypes = [vp, vp, vp, i64, i64, i64, vp, i32, vp, i32, f32, vp, vp]
    lib.ml3d_kpconv_deformed_weighted_backward.restype = C.c_int
    lib.ml3d_kpconv_deformed_weighted_backward.argtypes = [vp, vp, vp, i64, i64, i64, vp, i32, vp, i32, f32, vp, vp, vp, vp]
    lib.ml3d_kpconv_offset_regulariser_blocks.restype = i64
    lib.ml3d_kpconv_offset_regulariser_blocks.argtypes = [i64]
    lib.ml3d_kpconv_offset_regulariser.restype = C.c_int
    lib.ml3d_kpconv_offset_regulariser.argtypes = [vp, vp, vp, i64, i64, i64, vp, i32, f32, f32, vp, vp, vp, vp, vp]
    lib.ml3d_randla_attention_stage.restype = C.c_int
    lib.ml3d_randla_attention_stage.argtypes = [vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp, vp]
    lib.ml3d_randla_attention_stage_backward.restype = C.c_int
    lib.ml3d_randla_attention_stage_backward_workspace_bytes.restype = sz
    lib.ml3d_randla_attention_stage_backward_workspace_bytes.argtypes = [i64, i64, i32, i32]
    lib.ml3d_randla_attention_stage_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, vp, vp, sz, vp]
    lib.ml3d_vote_update.restype = C.c_int
    lib.ml3d_vote_update.argtypes = [vp, vp, i64, i32, f32, vp, i64, vp]
    lib.ml3d_argmax_labels.restype = C.c_int
    lib.ml3d_argmax_labels.argtypes = [vp, i64, i32, vp, vp]
    lib.ml3d_randla_pyramid_workspace_bytes.restype = sz
    lib.ml3d_randla_pyramid_workspace_bytes.argtypes = [i64, i64, i32, vp]
    lib.ml3d_randla_knn_pyramid.restype = C.c_int
    lib.ml3d_randla_knn_pyramid.argtypes = [vp, i64, i64, i32, vp, i32, vp, vp, vp, sz, vp]
    lib.ml3d_randla_param_layout.restype = C.c_int
    lib.ml3d_randla_param_layout.argtypes = [C.POINTER(RandlaDesc), vp, i32]
    lib.ml3d_randla_forward_workspace_bytes.restype = sz
    lib.ml3d_randla_forward_workspace_bytes.argtypes = [C.POINTER(RandlaDesc)]
    lib.ml3d_randla_forward.restype = C.c_int
    lib.ml3d_randla_forward.argtypes = [C.POINTER(RandlaDesc), vp, vp, vp, vp, vp, vp, vp, sz, vp]
    lib.ml3d_randla_forward_traced.restype = C.c_int
    lib.ml3d_randla_forward_traced.argtypes = [C.POINTER(RandlaDesc), vp, vp, vp, vp, vp, vp, vp, sz, vp, C.POINTER(Trace)]
    lib.ml3d_randla_knn_pyramid_traced.restype = C.c_int
    lib.ml3d_randla_knn_pyramid_traced.argtypes = [vp, i64, i64, i32, vp, i32, vp, vp, vp, sz, vp, C.POINTER(Trace)]
    lib.ml3d_randla_knn_pyramid_ordered.restype = C.c_int
    lib.ml3d_randla_knn_pyramid_ordered.argtypes = [vp, i64, i64, i32, vp, i32, vp, vp, vp, vp, sz, vp, C.POINTER(Trace)]
    lib.ml3d_randla_forward_ordered.restype = C.c_int
    lib.ml3d_randla_forward_ordered.argtypes = [C.POINTER(RandlaDesc), vp, vp, vp, vp, vp, vp, vp, vp, sz, vp,
                                                C.POINTER(Trace)]
    return lib


def get():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "ml3d: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        lib = bind(C.CDLL(LIB_PATH))
        got = int(lib.ml3d_abi_version())
        if got != ABI_VERSION:
            raise RuntimeError("ml3d: %s has ABI version %d, this binding was written against %d (include/ml3d_hip.h) — rebuild "
                               "it with `python -c 'import __graft_entry__ as g; g.build()'`" % (LIB_PATH, got, ABI_VERSION))
        _lib = lib
    return _lib


def require_gpu(device, what):
    """The package's one device gate: every model / batch class calls it before touching the library.  There is no CPU
    execution path behind any of them."""
    import torch
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    if dev.type != "cuda":
        raise RuntimeError("%s needs an MI355X device (got '%s'); there is no CPU fallback" % (what, dev))
    return dev


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s (%d)" % (what, ERRORS.get(rc, "error"), rc))


def make_desc(cfg, batch, num_points):
    d = RandlaDesc()
    L = int(cfg["num_layers"])
    if L > MAX_LAYERS:
        raise RuntimeError("num_layers > %d unsupported" % MAX_LAYERS)
    d.num_layers = L
    d.in_channels = int(cfg["in_channels"])
    d.dim_features = int(cfg["dim_features"])
    d.num_classes = int(cfg["num_classes"])
    d.num_neighbors = int(cfg["num_neighbors"])
    for l in range(L):
        d.dim_output[l] = int(cfg["dim_output"][l])
        d.sub_sampling_ratio[l] = int(cfg["sub_sampling_ratio"][l])
    d.batch = int(batch)
    d.num_points = int(num_points)
    return d


def randla_param_offsets(lib, desc):
    import numpy as np
    off = np.zeros(256, np.int64)
    n = lib.ml3d_randla_param_layout(C.byref(desc), off.ctypes.data, 256)
    if n < 0:
        check(n, "ml3d_randla_param_layout")
    return off[:n + 1].copy()


def ptr_table(ptrs):
    """HOST array of device pointers (kept alive by the caller)."""
    arr = (C.c_void_p * len(ptrs))(*[C.c_void_p(int(p)) for p in ptrs])
    return arr


# ---- ml3d_kpconv_batch_build (include/ml3d_hip.h): the host-side structs ---------------------------------------------------
KPBATCH_MAX_LAYERS = 8
KPBATCH_FALLBACK = 1
E_UNSUPPORTED = -4         # ML3D_E_UNSUPPORTED


class KpBatchDesc(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("cap", C.c_int32), ("has_conv", C.c_int32 * KPBATCH_MAX_LAYERS),
                ("radius", C.c_float * KPBATCH_MAX_LAYERS), ("dl", C.c_float * KPBATCH_MAX_LAYERS),
                ("trace_events", C.c_void_p * 8)]


class KpLayerOut(C.Structure):
    _fields_ = [("n_points", C.c_int64), ("points_offset", C.c_int64), ("conv_offset", C.c_int64), ("conv_cols", C.c_int64),
                ("pool_offset", C.c_int64), ("pool_cols", C.c_int64), ("up_offset", C.c_int64), ("up_cols", C.c_int64)]


class KpBatchOut(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("host_syncs", C.c_int32), ("arena_used", C.c_int64),
                ("layer", KpLayerOut * KPBATCH_MAX_LAYERS)]
