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


# the multi-rank code paths also at world size 1 (bench.py: the RCCL calls of the N > 1 path on a one-GPU box)
FORCE_GROUP = os.environ.get("ML3D_DIST_FORCE_GROUP") == "1"


def _single():
    return not dist.is_initialized() or (dist.get_world_size() == 1 and not FORCE_GROUP)


def init(backend, device=None):
    """Join the process group described by the torchrun environment (rendezvous on 127.0.0.1 by default)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if FORCE_GROUP:                                    # a lone process joining a group of one (no torchrun around it)
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
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
    if _single():
        return ([labels], None) if async_op else [labels]
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == dst and out is None:
        out = [torch.empty_like(labels) for _ in range(world)]
    work = dist.gather(labels, out if rank == dst else None, dst=dst, async_op=async_op)
    res = out if rank == dst else None
    return (res, work) if async_op else res


def gather_ragged(values, dst=0):
    """The ragged case of the prediction gather (KPConv batches hold a different number of points on every
    rank, PointPillars a different number of boxes): ``values`` is this rank's 1-D tensor of ANY length.  Sizes first (one tiny
    all_gather), then a gather of the tensors padded to the longest; ``dst`` gets the list of every rank's tensor trimmed back
    to its own length, the other ranks None.  The reference does the same for detection boxes with ``gather_object``
    (ml3d/torch/pipelines/object_detection.py:222-233); tensors avoid the pickling."""
    if _single():
        return [values]
    rank, world = dist.get_rank(), dist.get_world_size()
    n = torch.tensor([values.numel()], dtype=torch.int64, device=values.device)
    sizes = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(t.item()) for t in sizes]
    longest = max(sizes)
    padded = values.reshape(-1)
    if padded.numel() < longest:
        padded = torch.cat([padded, padded.new_zeros(longest - padded.numel())])
    out = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded.contiguous(), out, dst=dst)
    return [t[:k] for t, k in zip(out, sizes)] if rank == dst else None


def label_checksum(labels):
    """int64 [3] on the labels' device: (sum, position-weighted sum, xor-fold of 8-byte words) of a label tensor -- what a rank
    publishes about its own predictions so that the receiver of a gather can verify the bytes it was handed (bench.py, N > 1)."""
    flat = labels.reshape(-1).to(torch.int64)
    n = flat.numel()
    w = (torch.arange(n, device=flat.device, dtype=torch.int64) % 251) + 1
    raw = labels.reshape(-1).contiguous().view(torch.uint8)
    pad = (-raw.numel()) % 8
    if pad:
        raw = torch.cat([raw, torch.zeros(pad, dtype=torch.uint8, device=raw.device)])
    words = raw.view(torch.int64)
    x = words
    while x.numel() > 1:                      # xor-fold by halves (no bitwise reduction op in torch)
        if x.numel() % 2:
            x = torch.cat([x, torch.zeros(1, dtype=torch.int64, device=x.device)])
        x = x[: x.numel() // 2] ^ x[x.numel() // 2:]
    return torch.stack([flat.sum(), (flat * w).sum(), x.reshape(-1)[0] if x.numel() else torch.zeros((), dtype=torch.int64, device=flat.device)])


def compact_labels(scores):
    """argmax over classes as the smallest integer type that holds it (uint8 for <= 256 classes): what travels."""
    if scores.is_cuda and scores.dtype == torch.float32 and scores.is_contiguous() and scores.shape[-1] <= 256:
        from . import ops
        return ops.argmax_labels(scores)
    lab = torch.argmax(scores, dim=-1)
    return lab.to(torch.uint8 if scores.shape[-1] <= 256 else torch.int32)


class PredictionGather:
    """The N > 1 data path of the inference loop: argmax of a step's scores into one of ``depth`` label buffers, then an
    ASYNCHRONOUS gather of that buffer to ``dst`` so the collective of step i overlaps the kernels of step i + 1.  A
    buffer is reused only after its previous gather has completed (``work.wait()``).  Works on any backend (``nccl`` =
    RCCL over xGMI on the GPUs, ``gloo`` in the CPU tests); with no process group it degenerates to the local argmax."""

    def __init__(self, batch, num_points, num_classes, device, depth=2, dst=0):
        self.dst = dst
        self.depth = int(depth)
        dt = torch.uint8 if int(num_classes) <= 256 else torch.int32
        self.labels = [torch.empty((batch, num_points), dtype=dt, device=device) for _ in range(self.depth)]
        on = not _single()
        self.on = on
        self.world = dist.get_world_size() if on else 1
        self.rank = dist.get_rank() if on else 0
        self.recv = [[torch.empty_like(self.labels[0]) for _ in range(self.world)] if (on and self.rank == dst) else None
                     for _ in range(self.depth)]
        self.pending = [None] * self.depth
        self.step = 0

    def push(self, scores):
        """scores [B, N, C] of this rank's frames -> slot index of the label buffer / receive list used for this step."""
        i = self.step % self.depth
        self.step += 1
        if self.pending[i] is not None:
            self.pending[i].wait()
            self.pending[i] = None
        if scores.is_cuda and self.labels[i].dtype == torch.uint8 and scores.dtype == torch.float32 and scores.is_contiguous():
            from . import ops
            ops.argmax_labels(scores, out=self.labels[i])       # one HIP kernel: 19 floats in, 1 byte out per point
        else:
            self.labels[i].copy_(torch.argmax(scores, dim=2))
        if self.on:
            _, self.pending[i] = gather_predictions(self.labels[i], dst=self.dst, out=self.recv[i], async_op=True)
        return i

    def gathered(self, slot):
        """On ``dst``: the list of every rank's label tensor of that slot's last step (call after ``drain`` or after the
        slot's work has been waited for); elsewhere / single process: this rank's labels."""
        if self.on and self.rank == self.dst:
            return self.recv[slot]
        return [self.labels[slot]]

    def drain(self):
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None


# ---- host placement of the ranks (one process per GPU on a multi-socket node) ---------------------------------------------
def _parse_cpulist(text):
    """'0-15,128-143' -> [0..15, 128..143] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(pci_bus_id, sysfs="/sys"):
    """NUMA node of the PCI function ``dddd:bb:dd.f`` (``/sys/bus/pci/devices/<id>/numa_node``); -1 when the platform does not
    say (single-socket hosts, containers without sysfs)."""
    try:
        with open(os.path.join(sysfs, "bus", "pci", "devices", pci_bus_id.lower(), "numa_node")) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def plan_rank_cpus(local_rank, gpu_nodes, allowed, sysfs="/sys", max_threads=16):
    """Which CPUs rank ``local_rank`` of ``len(gpu_nodes)`` local ranks should run on.

    ``gpu_nodes[i]`` = NUMA node of local rank i's GPU (-1 = unknown), ``allowed`` = the CPUs this job may use
    (``os.sched_getaffinity(0)``).  Ranks whose GPUs hang off the same node share that node's allowed CPUs in equal contiguous
    slices (rank order); ranks with an unknown node split ``allowed`` evenly among ALL ranks instead -- either way no two ranks
    of a node share a core, and the host threads of a rank (collate, the ~100 small launches and read-backs of a KPConv batch
    build) stay next to their GPU's memory controller and PCIe root.  Returns (sorted CPU list, numa node, host thread count =
    min(max_threads, CPUs)).  The reference spawns its ranks without placement (scripts/run_pipeline.py:195-216)."""
    allowed = sorted(allowed)
    n = len(gpu_nodes)
    node = gpu_nodes[local_rank]
    pool, peers = None, None
    if node >= 0:
        try:
            with open(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")) as f:
                on_node = set(_parse_cpulist(f.read()))
            pool = [c for c in allowed if c in on_node]
            peers = [i for i in range(n) if gpu_nodes[i] == node]
        except (OSError, ValueError):
            pool = None
    if not pool or len(pool) < len(peers or [0]):
        node, pool, peers = -1, allowed, list(range(n))
    k, m = peers.index(local_rank), len(peers)
    per = max(1, len(pool) // m)
    mine = pool[k * per:(k + 1) * per] or pool[-1:]
    return mine, node, max(1, min(int(max_threads), len(mine)))


def bind_rank(local_rank, local_world, device_pci_ids=None, sysfs="/sys", max_threads=16):
    """Pin this process to its slice (``plan_rank_cpus``) with ``os.sched_setaffinity`` and return
    {"numa_node", "cpus" (cpulist string), "host_threads"} -- the rank -> CPU map bench.py prints in ``ranks_seen``.
    ``device_pci_ids``: PCI bus id of every LOCAL rank's GPU in rank order (None: unknown, e.g. the CPU dry run)."""
    allowed = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else set(range(os.cpu_count() or 1))
    nodes = [gpu_numa_node(p, sysfs) if p else -1 for p in (device_pci_ids or [None] * local_world)]
    cpus, node, threads = plan_rank_cpus(local_rank, nodes, allowed, sysfs, max_threads)
    if hasattr(os, "sched_setaffinity") and local_world > 1:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    # compress to the kernel's cpulist form
    runs, start, prev = [], None, None
    for c in cpus:
        if start is None:
            start = prev = c
        elif c == prev + 1:
            prev = c
        else:
            runs.append((start, prev)); start = prev = c
    if start is not None:
        runs.append((start, prev))
    return {"numa_node": node, "cpus": ",".join("%d" % a if a == b else "%d-%d" % (a, b) for a, b in runs),
            "host_threads": threads, "bound": local_world > 1}
