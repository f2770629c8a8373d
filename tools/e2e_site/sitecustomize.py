"""HARNESS ONLY (tools/run_pipeline_e2e.py puts this directory on PYTHONPATH): what must be true of the interpreter BEFORE the
reference's unchanged ``scripts/run_pipeline.py`` starts, without touching that script:

  * third-party packages the reference imports and this image lacks: ``addict`` (tests/stubs) and ``tensorboard``
    (``torch.utils.tensorboard`` is replaced by an inert SummaryWriter when the real one cannot be imported);
  * ML3D_E2E_SEED: python's ``random``, ``numpy.random`` and torch seeded at the ENTRY of the pipeline's ``run_test``
    (run_pipeline.py seeds only its own Generator; the samplers and KPFCNN's test-time augmentation draw from the global
    streams, and the reference's KPFCNN constructor consumes numpy draws for its kernel-point optimisation that the native
    constructor does not -- the two sides must draw the same numbers DURING the test to be comparable);
  * ML3D_E2E_SIDE=reference: ``open3d`` = oracle/ref_shim.py (the reference's PyTorch-CPU models on the oracle's C ops -- the
    "reference PyTorch-CPU path" of north_star), with ``open3d.ml`` re-exporting the checkout like the real wheel does;
    ML3D_E2E_SIDE=native: nothing to do, ``open3d`` is this repository's package (open3d-ml_amd/open3d) via PYTHONPATH;
  * ML3D_E2E_EMU=1 (debugging host glue without a GPU): the host emulation of the HIP library (tests/emu_runtime.py).
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _TensorboardStub(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name != "torch.utils.tensorboard":
            return None
        try:
            import tensorboard  # noqa: F401
            return None
        except Exception:
            return importlib.machinery.ModuleSpec(name, self)

    def create_module(self, spec):
        m = types.ModuleType(spec.name)

        class SummaryWriter:
            def __init__(self, *a, **k):
                pass

            def __getattr__(self, n):
                return lambda *a, **k: None
        m.SummaryWriter = SummaryWriter
        return m

    def exec_module(self, module):
        pass


if os.environ.get("ML3D_E2E_SIDE"):
    sys.meta_path.insert(0, _TensorboardStub())
    for p in (os.path.join(_ROOT, "tests", "stubs"), _ROOT):
        if p not in sys.path:
            sys.path.append(p)
    def _reseed_at_run_test():
        import functools
        import random

        import numpy as _np
        import torch as _torch
        from ml3d.torch import pipelines as _pl
        seed = int(os.environ["ML3D_E2E_SEED"])
        for cls in (_pl.SemanticSegmentation, _pl.ObjectDetection):
            orig = cls.run_test

            def run_test(self, _orig=orig):
                random.seed(seed)
                _np.random.seed(seed)
                _torch.manual_seed(seed)
                return _orig(self)
            cls.run_test = functools.wraps(orig)(run_test)
    if os.environ["ML3D_E2E_SIDE"] == "reference":
        from oracle import ref_shim
        ref_shim.install()
        import ml3d.datasets
        import ml3d.torch
        import ml3d.utils
        _ml = sys.modules["open3d.ml"]
        _ml.utils, _ml.datasets = ml3d.utils, ml3d.datasets
    else:
        import open3d.ml.torch  # noqa: F401   (this repository's package: re-exports the checkout, registers the native models)
        if os.environ.get("ML3D_E2E_EMU") == "1":
            sys.path.insert(0, os.path.join(_ROOT, "tests"))
            import emu_runtime
            emu_runtime.install("ml3d_amd")
    if os.environ.get("ML3D_E2E_SEED"):
        _reseed_at_run_test()
