"""TEST INFRASTRUCTURE ONLY: run the product's PYTHON host side (``ml3d.ops``, the model classes, the ``open3d`` shim) on
CPU tensors against the HOST EMULATION of ``libml3d_hip.so`` (tests/hipemu — the same ``.hip`` sources compiled for the
host), so that host glue (the reference's pipelines driving the native model classes, batchers, vote updates) can be
exercised in this GPU-less container.

The product has no such mode: every op insists on HIP tensors (``ops._gates._need_gpu``), every model on a HIP device
(``_abi.require_gpu``) and ``_abi.get()`` on the hipcc-built library.  ``install()`` monkeypatches exactly those three gates
(plus the handful of ``torch.cuda`` stream / device context calls the host side makes) INSIDE THE TEST PROCESS.  Nothing
under ``open3d-ml_amd/`` imports this module, and it is useless on a GPU box (the -m gpu tests run the real library).
"""
import contextlib
import ctypes as C
import importlib
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class _Stream:
    cuda_stream = 0
    priority = 0

    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass

    def wait_event(self, ev):
        pass

    def wait_stream(self, s):
        pass

    def record_event(self, ev=None):
        return ev or _Event()

    def query(self):
        return True

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Event:
    cuda_event = 0

    def __init__(self, *a, **k):
        pass

    def record(self, stream=None):
        pass

    def synchronize(self):
        pass

    def wait(self, stream=None):
        pass

    def query(self):
        return True

    def elapsed_time(self, other):
        return 0.0


@contextlib.contextmanager
def _null(*a, **k):
    yield


def emu_library_path():
    so = os.environ.get("ML3D_EMU_LIB")
    if so:
        return so
    out = subprocess.check_output([os.path.join(HERE, "hipemu", "build_emu.sh")]).decode().strip().splitlines()
    return out[-1]


def install(product="ml3d"):
    """Patch the product package ``product`` (module name: ``ml3d`` when this repository's package is on the path under its
    own name, ``ml3d_amd`` when the ``open3d`` shim loaded it beside a reference checkout) to run on the emulator."""
    prod = importlib.import_module(product)
    _abi = importlib.import_module(product + "._abi")
    ops = importlib.import_module(product + ".ops")
    _abi._lib = _abi.bind(C.CDLL(emu_library_path()))
    _abi.require_gpu = lambda device, what: (torch.device(device) if not isinstance(device, torch.device) else device)
    gates = importlib.import_module(product + ".ops._gates")      # the one place every op checks / takes its stream from
    gates._need_gpu = lambda *t: None
    gates._stream = lambda: None
    torch.cuda.device = _null
    torch.cuda.stream = _null
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.Stream = _Stream
    torch.cuda.Event = _Event
    torch.cuda.synchronize = lambda *a, **k: None
    torch.Tensor.record_stream = lambda self, s: None
    _pin = torch.Tensor.pin_memory
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    shim = sys.modules.get("open3d._product")
    if shim is not None:
        shim.device = lambda: torch.device("cpu")
    return prod
