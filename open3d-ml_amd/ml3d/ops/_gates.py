"""The package's gates and small shared helpers: every op checks its tensors here, takes torch's current stream here and
allocates its workspaces here -- ONE place (the test-only emulator harness, tests/emu_runtime.py, patches exactly these)."""
import ctypes as C
from collections import namedtuple

import numpy as np
import torch

KnnResult = namedtuple("KnnResult", ["neighbors_index", "neighbors_distance"])
RadiusResult = namedtuple("RadiusResult", ["neighbors_index", "neighbors_row_splits", "neighbors_distance"])
VoxelizeResult = namedtuple("VoxelizeResult", ["voxel_coords", "voxel_point_indices", "voxel_point_row_splits",
                                               "voxel_batch_splits"])


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise RuntimeError("ml3d.ops: HIP kernels need tensors on an MI355X device (got %s); "
                               "there is no CPU fallback" % (getattr(t, "device", type(t)),))


def _ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


# (conv / pool / upsample searches, subsampling and the two rotations of a layer): a small cache keyed by (lengths, device,
# stream) turns ~40 tiny host-to-device copies per batch into ~6.  (The stream is part of the key: a cached tensor is only
# handed to work enqueued on the stream its upload was ordered on.)
_SPLITS_CACHE = {}


def _splits_of_lengths(lengths, dev):
    if torch.is_tensor(lengths):
        lengths = lengths.tolist()
    key = (tuple(int(v) for v in lengths), str(dev), torch.cuda.current_stream(dev).cuda_stream)
    hit = _SPLITS_CACHE.get(key)
    if hit is None:
        host = np.zeros(len(key[0]) + 1, np.int64)
        np.cumsum(np.asarray(key[0], np.int64), out=host[1:])
        hit = (torch.from_numpy(host).to(dev), int(host[-1]))
        if len(_SPLITS_CACHE) >= 64:
            _SPLITS_CACHE.clear()
        _SPLITS_CACHE[key] = hit
    return hit


def _splits(rs, n, dev):
    if rs is None:
        return torch.tensor([0, int(n)], dtype=torch.int64, device=dev)
    return rs.to(device=dev, dtype=torch.int64).contiguous()
