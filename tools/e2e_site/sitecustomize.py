"""HARNESS ONLY (tools/run_pipeline_e2e.py puts this directory on PYTHONPATH): what must be true of the interpreter BEFORE the
reference's unchanged ``scripts/run_pipeline.py`` starts, without touching that script:

  * third-party packages the reference imports and this image lacks: ``addict`` (tests/stubs) and ``tensorboard``
    (``torch.utils.tensorboard`` is replaced by an inert SummaryWriter when the real one cannot be imported);
  * ML3D_E2E_SEED: python's ``random``, ``numpy.random`` and torch seeded at the ENTRY of the pipeline's ``run_test`` / ``run_train``
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
            for entry in ("run_test", "run_train"):
                orig = getattr(cls, entry)

                def seeded(self, _orig=orig, _entry=entry):
                    random.seed(seed)
                    _np.random.seed(seed)
                    _torch.manual_seed(seed)
                    if _entry == "run_train":
                        # (harness output only: the loss of EVERY step -- the pipeline logs the epoch mean; Adam turns last-bit
                        #  gradient noise of near-zero gradients into +-lr steps, so the first step is the tight comparison)
                        # Dropout masks are drawn per tensor ELEMENT: the reference keeps fc1's activations channel-major
                        # [B, C, N, 1], the native training forward point-major [B, N, C] (and on a GPU the stream differs
                        # anyway), so the same seed gives the two sides different masks.  Off on BOTH sides for this run.
                        for mod in self.model.modules():
                            if isinstance(mod, _torch.nn.Dropout):
                                mod.p = 0.0
                        gl, n = self.model.get_loss, [0]

                        def get_loss(*a, **k):
                            out = gl(*a, **k)
                            print("e2e-step-loss %d %s %.6f" % (n[0], "train" if self.model.training else "valid", float(out[0])), flush=True)
                            n[0] += 1
                            return out
                        self.model.get_loss = get_loss
                    return _orig(self)
                setattr(cls, entry, functools.wraps(orig)(seeded))
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
