"""CPU: the N > 1 path — frame sharding + gather of predictions to rank 0 — with world_size 2 over gloo
(what bench.py runs over RCCL with one process per GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, n_points, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ml3d import dist as mdist
    mdist.init("gloo")
    b, e = mdist.shard_range(n_frames, rank, world)
    # a stand-in "prediction": label = f(frame id, point id); every rank only computes its own frames
    frames = torch.arange(b, e).view(-1, 1)
    labels = ((frames * 7 + torch.arange(n_points).view(1, -1) * 3) % 19).to(torch.int32)
    got = mdist.gather_predictions(labels, dst=0)
    # async variant with a preallocated receive list (what bench.py overlaps with the next step)
    u8 = labels.to(torch.uint8)
    pre = [torch.empty_like(u8) for _ in range(world)] if rank == 0 else None
    res, work = mdist.gather_predictions(u8, dst=0, out=pre, async_op=True)
    work.wait()
    if rank == 0:
        assert torch.equal(torch.cat(res, 0), torch.cat(got, 0).to(torch.uint8))
    if rank == 0:
        np.save(out_path, torch.cat(got, 0).numpy())
    else:
        assert got is None
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_shard_ranges_cover_everything_once():
    from ml3d.dist import shard_range
    for n in (0, 1, 7, 8, 64, 1001):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_gather_labels_in_frame_order(tmp_path):
    world, n_frames, n_points = 2, 6, 257
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_frames, n_points, out), nprocs=world, join=True)
    got = np.load(out)
    ref = (np.arange(n_frames)[:, None] * 7 + np.arange(n_points)[None, :] * 3) % 19
    assert got.shape == (n_frames, n_points) and np.array_equal(got, ref)


def test_single_process_gather_is_identity():
    from ml3d.dist import gather_predictions
    x = torch.arange(12).view(3, 4)
    assert gather_predictions(x)[0] is x
