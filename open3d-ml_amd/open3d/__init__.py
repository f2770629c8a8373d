"""``open3d`` — the slice of the Open3D Python package that Open3D-ML's inference hot path imports, served by the
MI355X-native library of this repository.

Open3D-ML has no native code of its own: ``ml3d/datasets/utils/dataprocessing.py:3,6``, ``ml3d/torch/models/
{kpconv,point_pillars}.py`` and ``ml3d/torch/utils/objdet_helper.py:27`` import their primitives from the (un-vendored)
``open3d`` wheel.  With this directory's parent on ``sys.path`` those imports resolve HERE: the functions accept what the
reference hands them (numpy arrays, CPU or GPU ``torch`` tensors), move host data to the current HIP device, call the C ABI
of ``libml3d_hip.so`` (``include/ml3d_hip.h``) through ``ml3d.ops``, and return the reference's types on the caller's
device.  There is no CPU implementation behind them: without an MI355X they raise.

Only the hot-path surface exists (``core.nns``, ``core.Tensor.from_numpy``, ``core.cuda``, ``ml.contrib``,
``ml.torch.ops``, ``ml.torch.layers``, ``_build_config`` and an inert ``visualization.tensorboard_plugin.summary``);
geometry, I/O, visualisation and the other 12 ML ops are out of scope.

``OPEN3D_ML_ROOT`` (as in upstream Open3D): when it names an Open3D-ML checkout, ``open3d.ml`` / ``open3d.ml.torch``
re-export that checkout's ``ml3d`` (utils, datasets, pipelines, dataloaders) unchanged, and the three hot-path model
classes of this repository are registered over the checkout's, so ``scripts/run_pipeline.py`` runs as is.
"""
__version__ = "0.19.0+ml3d.amd.gfx950"

_build_config = {
    "BUILD_GUI": False,
    "BUILD_CUDA_MODULE": True,          # device ops exist (HIP on gfx950); selects the *_cuda contrib names
    "BUILD_PYTORCH_OPS": True,
    "BUILD_TENSORFLOW_OPS": False,
    "BUILD_JUPYTER_EXTENSION": False,
    "BUNDLE_OPEN3D_ML": False,
    "CUDA_VERSION": "",
    "CUDA_GENCODES": "gfx950",
}

from . import core            # noqa: E402,F401
from . import visualization   # noqa: E402,F401
# `open3d.ml` is imported on demand (``import open3d.ml``): it pulls in torch and, with OPEN3D_ML_ROOT, the checkout
