"""Frame-parallel inference across the GPUs of one node.

Clouds are independent units (no cross-frame state in ``forward``): one process per GPU (``torchrun``), rank r of
W owns a full weight replica and a contiguous block of the clouds; the ONLY data-path collective is the gather of
the predicted labels to rank 0 (``torch.distributed.gather`` — RCCL over xGMI with the ``nccl`` backend, ``gloo``
in the CPU tests).  The reference does the same for detection boxes with ``gather_object``
(ml3d/torch/pipelines/object_detection.py:222-233) and refuses DDP for segmentation (base_pipeline.py:44-47).
"""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend, device=None):
    """Join the process group described by the torchrun environment (rendezvous on 127.0.0.1 by default)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if dist.is_initialized():
        return
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)


def shard_range(n_items, rank, world):
    """Contiguous block [begin, end) of ``n_items`` clouds owned by ``rank``; blocks differ by at most one item."""
    base, rem = divmod(int(n_items), int(world))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_predictions(labels, dst=0, out=None, async_op=False):
    """labels: this rank's [B_local, N] tensor (same shape on every rank).  Returns the list of every rank's
    tensor on ``dst`` (rank order = cloud order for ``shard_range`` blocks of equal size), None elsewhere.
    A no-op list of one tensor when no process group is initialised.  ``out``: preallocated receive list on
    ``dst``.  ``async_op=True`` returns (list, work): the collective runs on the backend's own stream so the next
    step's kernels overlap it; call ``work.wait()`` before reusing ``labels`` / reading the list."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return ([labels], None) if async_op else [labels]
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == dst and out is None:
        out = [torch.empty_like(labels) for _ in range(world)]
    work = dist.gather(labels, out if rank == dst else None, dst=dst, async_op=async_op)
    res = out if rank == dst else None
    return (res, work) if async_op else res


def compact_labels(scores):
    """argmax over classes as the smallest integer type that holds it (uint8 for <= 256 classes): what travels."""
    lab = torch.argmax(scores, dim=-1)
    return lab.to(torch.uint8 if scores.shape[-1] <= 256 else torch.int32)
