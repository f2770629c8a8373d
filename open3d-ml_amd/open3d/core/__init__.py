"""``open3d.core`` — what ``ml3d/datasets/utils/dataprocessing.py:99-103`` and ``ml3d/metrics/__init__.py:3`` use."""
import numpy as np

from . import cuda   # noqa: F401
from . import nns    # noqa: F401


class Dtype:
    Float32 = np.float32
    Float64 = np.float64
    Int32 = np.int32
    Int64 = np.int64


float32, float64, int32, int64 = np.float32, np.float64, np.int32, np.int64


class Tensor:
    """Host tensor handle: the reference only round-trips numpy through it (``Tensor.from_numpy(a)`` ... ``.numpy()``)."""

    def __init__(self, array, dtype=None):
        self._a = np.asarray(array, dtype=dtype)

    @staticmethod
    def from_numpy(array):
        return Tensor(array)

    def numpy(self):
        return self._a

    def cpu(self):
        return self

    @property
    def shape(self):
        return self._a.shape

    @property
    def dtype(self):
        return self._a.dtype

    def __len__(self):
        return len(self._a)
