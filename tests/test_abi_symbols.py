"""CPU: the HIP library builds for gfx950, loads, and exports every symbol include/ml3d_hip.h declares.
No compute call is made (no GPU in the authoring container)."""
import ctypes as C
import os
import re

import pytest

import __graft_entry__ as ge
from ml3d import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    ge.build()
    return _abi.get()


def _declared():
    src = open(os.path.join(ROOT, "include", "ml3d_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ml3d_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built):
    names = _declared()
    assert len(names) >= 8
    for n in names:
        assert hasattr(built, n), "missing export: " + n
    assert sorted(_abi.SYMBOLS) == names, "ml3d/_abi.py SYMBOLS out of sync with include/ml3d_hip.h"


def test_abi_version_and_sizes(built):
    hdr = open(os.path.join(ROOT, "include", "ml3d_hip.h")).read()
    ver = int(re.search(r"#define\s+ML3D_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert built.ml3d_abi_version() == ver == _abi.ABI_VERSION
    assert built.ml3d_knn_workspace_bytes(1000, 1000, 1) > 1000 * 16
    desc = _abi.make_desc(dict(num_layers=4, in_channels=3, dim_features=8, num_classes=19, num_neighbors=16,
                               dim_output=[16, 64, 128, 256], sub_sampling_ratio=[4, 4, 4, 4]), 2, 45056)
    off = _abi.randla_param_offsets(built, desc)
    assert len(off) == 2 + 18 * 4 + 2 + 8 + 6 + 1
    assert 1200000 < off[-1] < 1242307 + 512      # conv/linear weights + biases; BatchNorm is folded away
    assert built.ml3d_randla_forward_workspace_bytes(C.byref(desc)) > 2 * 45056 * 19 * 4
    assert C.sizeof(_abi.RandlaDesc) == 104


def test_invalid_arguments_are_rejected_without_a_gpu(built):
    # argument validation happens before any HIP call
    assert built.ml3d_knn_search(None, None, None, None, 0, 0, 0, 16, 0, None, None, None, 0, None) == -1
    bad = _abi.make_desc(dict(num_layers=1, in_channels=3, dim_features=8, num_classes=19, num_neighbors=16,
                              dim_output=[15], sub_sampling_ratio=[4]), 1, 64)
    assert built.ml3d_randla_forward_workspace_bytes(C.byref(bad)) == 0


def test_product_ops_refuse_cpu_tensors(built):
    import torch
    from ml3d import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.knn_search(torch.zeros(10, 3), torch.zeros(10, 3), 4)
    from ml3d.torch.models.randlanet import RandLANet
    m = RandLANet(device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m({"coords": [torch.zeros(1, 64, 3)], "features": torch.zeros(1, 64, 3)})


def test_stale_library_is_refused(built, monkeypatch):
    """a library built from another revision of the header must not be called (its arguments would be misread)"""
    monkeypatch.setattr(_abi, "_lib", None)
    monkeypatch.setattr(_abi, "ABI_VERSION", _abi.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        _abi.get()
    monkeypatch.setattr(_abi, "ABI_VERSION", _abi.ABI_VERSION - 1)
    assert _abi.get() is not None
