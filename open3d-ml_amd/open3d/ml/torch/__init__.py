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
        import importlib
        import logging
        from ml3d.utils import MODEL
        from ml3d.torch.models.kpconv import KPFCNN as _ReferenceKPFCNN
        from ... import _product
        _product.product()
        native = importlib.import_module("ml3d_amd.torch.models")

        class KPFCNN(native.KPFCNN):
            """The registry entry for ``KPFCNN``: the MI355X-native class -- rigid blocks and, since round 3, the deformable ones
            of ml3d/configs/kpconv_parislille3d.yml:28-32 (``KP_influence: linear``).  A config the native class refuses at
            construction (NotImplementedError: a deformable block with another influence function or an unusual width) FALLS
            BACK to the checkout's own PyTorch ``KPFCNN`` (kpconv.py:1011-1041, 1071-1103), which then runs on PyTorch-ROCm
            with its neighbour searches / subsampling served by the native ops of this shim."""
            _warned = False

            def __new__(cls, *args, **kwargs):
                arch = kwargs.get("architecture", None)
                if arch is not None and any("deformable" in str(b) for b in arch):
                    try:
                        native.KPFCNN(*args, **dict(kwargs, device="cpu"))        # (a dry construction: parameters only)
                    except NotImplementedError as e:
                        if not KPFCNN._warned:
                            logging.getLogger(__name__).warning(
                                "KPFCNN: %s; using the Open3D-ML checkout's PyTorch KPFCNN on the native ops instead", e)
                            KPFCNN._warned = True
                        return _ReferenceKPFCNN(*args, **kwargs)
                return super().__new__(cls)

        KPFCNN.__module__ = native.KPFCNN.__module__
        KPFCNN.__qualname__ = "KPFCNN"
        for cls in (native.RandLANet, KPFCNN, native.PointPillars):
            MODEL._register_module(cls, "torch")
            setattr(models, cls.__name__, cls)

    if _os.environ.get("ML3D_AMD_KEEP_REFERENCE_MODELS", "0") != "1":
        _register_native_models()
