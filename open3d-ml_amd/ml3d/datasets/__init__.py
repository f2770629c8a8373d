"""Data formats either side of the hot path: SemanticKITTI sweep readers / prediction writer and
the raw-sweep front end (0.06 m grid subsample + raw->sub projection) on the GPU ops."""
from .semantickitti import (SemanticKITTIFormat, load_label_kitti, load_pc_kitti, preprocess_sweep)  # noqa: F401
