"""Training-side host modules of the hot-path models: losses and detection targets, on torch tensors of any
device.  The reference keeps them in ml3d/torch/modules/losses/ and ml3d/torch/utils/objdet_helper.py."""
from .losses import CrossEntropyLoss, FocalLoss, SmoothL1Loss, valid_scores_and_labels   # noqa: F401
from .anchor_targets import assign_anchor_targets, encode_boxes, nearest_bev_boxes, pairwise_iou_xyxy   # noqa: F401
