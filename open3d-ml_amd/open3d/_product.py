"""Access to this repository's product package from the ``open3d`` shim.

The product package is called ``ml3d`` (it mirrors the reference's interface), and so is the reference's own package,
which must stay importable as ``ml3d`` when ``OPEN3D_ML_ROOT`` points at a checkout (its modules import each other by
that name).  The shim therefore loads the product under the alias ``ml3d_amd`` straight from its directory — the
product only uses relative imports — and never touches ``sys.modules['ml3d']``."""
import importlib.util
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(os.path.dirname(_HERE), "ml3d")


def product():
    m = sys.modules.get("ml3d_amd")
    if m is None:
        spec = importlib.util.spec_from_file_location("ml3d_amd", os.path.join(_PKG, "__init__.py"),
                                                      submodule_search_locations=[_PKG])
        m = importlib.util.module_from_spec(spec)
        sys.modules["ml3d_amd"] = m
        spec.loader.exec_module(m)
    return m


def ops():
    import importlib
    product()
    return importlib.import_module("ml3d_amd.ops")


def device():
    """The HIP device host data is moved to: torch's current device.  Raises without an MI355X (no CPU fallback)."""
    if not torch.cuda.is_available():
        raise RuntimeError("open3d (ml3d_amd shim): the ops behind this call are HIP kernels for MI355X and there is no "
                           "CPU implementation; no GPU is visible")
    return torch.device("cuda", torch.cuda.current_device())


def to_dev(x, dtype=None):
    """numpy array / CPU tensor / GPU tensor -> contiguous tensor on the HIP device (+ where it came from)."""
    if x is None:
        return None, None
    if isinstance(x, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(x))
        src = "numpy"
    elif isinstance(x, torch.Tensor):
        t = x
        src = x.device
    else:
        t = torch.as_tensor(x)
        src = "numpy"
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if not t.is_cuda:
        t = t.to(device())
    return t.contiguous(), src


def back(t, src):
    """tensor on the HIP device -> the caller's kind: numpy for numpy inputs, else a tensor on the input's device."""
    if t is None:
        return None
    if src == "numpy":
        return t.cpu().numpy()
    return t.to(src)
