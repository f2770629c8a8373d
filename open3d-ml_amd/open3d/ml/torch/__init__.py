"""``open3d.ml.torch`` — ``ops`` and ``layers`` (native, this repository); with ``OPEN3D_ML_ROOT`` also the checkout's
torch side (``models``, ``modules``, ``pipelines``, ``dataloaders``) exactly as upstream's ``open3d/ml/torch/__init__.py``
re-exports it — with ONE change: the three inference hot-path models (``RandLANet``, ``KPFCNN``, ``PointPillars``) of this
repository are registered over the checkout's in ``ml3d.utils``'s MODEL registry, so
``get_module("model", "RandLANet", "torch")`` (``scripts/run_pipeline.py:129-132``) returns the MI355X-native class.
Set ``ML3D_AMD_KEEP_REFERENCE_MODELS=1`` to leave the registry alone (the reference's PyTorch forwards then run on the
native ops only)."""
import os as _os

from . import layers   # noqa: F401
from . import ops      # noqa: F401
from .. import _checkout

if _checkout():
    from ml3d.torch import dataloaders, models, modules, pipelines   # noqa: F401
    from ml3d.torch.dataloaders import *    # noqa: F401,F403
    from ml3d.utils import Config, get_module   # noqa: F401

    def _register_native_models():
        from ml3d.utils import MODEL
        from ... import _product
        import importlib
        _product.product()
        native = importlib.import_module("ml3d_amd.torch.models")
        for cls in (native.RandLANet, native.KPFCNN, native.PointPillars):
            MODEL._register_module(cls, "torch")
            setattr(models, cls.__name__, cls)

    if _os.environ.get("ML3D_AMD_KEEP_REFERENCE_MODELS", "0") != "1":
        _register_native_models()
