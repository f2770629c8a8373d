"""``ml3d.ops`` — torch front end of libml3d_hip.so (MI355X / gfx950 only).

Mirrors the call surface the reference gets from ``open3d.ml.torch.ops`` /
``open3d.core.nns`` for the inference hot path.  Every function
requires CUDA(HIP) tensors and raises ``RuntimeError`` otherwise: this package has no
CPU path (the CPU oracle lives under /oracle and is test infrastructure only).
All kernels are enqueued on torch's CURRENT stream.
"""
from ._gates import KnnResult, RadiusResult, VoxelizeResult   # noqa: F401
from .search import (knn_search, pyramid_sizes, randla_knn_pyramid, resolve_plans, fixed_radius_search,   # noqa: F401
                     radius_plan_dense, radius_fill_dense, radius_neighbors_dense, ragged_to_dense, _RadiusPlan,
                     _DenseRadiusPlan, kpconv_batch_build)
from .randla import randla_forward, GatherMaxFunction, AttentivePoolFunction   # noqa: F401
from .voxel import (voxelize, subsample_batch, subsample, rotate_points, grid_subsampling_plan,   # noqa: F401
                    batch_grid_subsampling, _SubsamplePlan)
from .kpconv import (kpconv_rigid, kpconv_deformable, linear, gather_pool, kpconv_weighted, kpconv_weighted_backward,   # noqa: F401
                     KPConvFunction)
from .detection import (pillar_features, conv2d_nhwc, pack_bf16x3, linear_bf16x3, deconv2d_nhwc, nhwc_to_nchw, nms, pointpillars_boxes,   # noqa: F401
                        iou_bev, iou_3d, topk_rows)
from .sampler import nearest_to_center, argmax_labels, vote_update, device_patch   # noqa: F401
from .train import (gemm_tn, LinearFunction, BatchNormActFunction, batch_norm_act, GatherRowsFunction, GatherPoolFunction,   # noqa: F401
                    AttentionStageFunction, attention_stage_supported, KPConvDeformedFunction, OffsetRegulariserFunction)
