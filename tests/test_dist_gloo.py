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


def _gather_worker(rank, world, port, steps, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ml3d import dist as mdist
    mdist.init("gloo")
    B, N, C = 3, 129, 19
    g = mdist.PredictionGather(B, N, C, "cpu")
    got = {}
    for step in range(steps):
        # every rank's scores of this step are a pure function of (rank, step): rank 0 can rebuild what it must receive
        gen = torch.Generator().manual_seed(1000 * rank + step)
        scores = torch.rand((B, N, C), generator=gen)
        slot = g.push(scores)
        assert slot == step % 2
        if step >= 1:
            # the buffer of the PREVIOUS step is complete once its work has been waited for; bench.py reads results only
            # after drain(), here we check the double buffering step by step
            prev = (step - 1) % 2
            if g.pending[prev] is not None:
                g.pending[prev].wait()
            if rank == 0:
                got[step - 1] = torch.cat([t.clone() for t in g.gathered(prev)], 0)
    g.drain()
    if rank == 0:
        got[steps - 1] = torch.cat([t.clone() for t in g.gathered((steps - 1) % 2)], 0)
        np.save(out_path, torch.stack([got[s] for s in range(steps)]).numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_prediction_gather_pipeline_two_ranks(tmp_path):
    """bench.py's N > 1 step logic (argmax -> uint8 labels -> async gather, two buffers) with real tensors over gloo."""
    world, steps = 2, 5
    out = str(tmp_path / "g.npy")
    mp.spawn(_gather_worker, args=(world, _free_port(), steps, out), nprocs=world, join=True)
    got = np.load(out)
    assert got.shape == (steps, world * 3, 129) and got.dtype == np.uint8
    for step in range(steps):
        for r in range(world):
            gen = torch.Generator().manual_seed(1000 * r + step)
            want = torch.rand((3, 129, 19), generator=gen).argmax(2).numpy()
            assert np.array_equal(got[step, 3 * r:3 * r + 3], want)


def test_prediction_gather_single_process():
    from ml3d.dist import PredictionGather
    g = PredictionGather(2, 5, 300, "cpu")          # > 256 classes: int32 labels
    s = torch.rand((2, 5, 300))
    slot = g.push(s)
    g.drain()
    assert g.gathered(slot)[0].dtype == torch.int32 and torch.equal(g.gathered(slot)[0].long(), s.argmax(2))


def _ragged_worker(rank, world, port, out_path):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ml3d import dist as mdist
    mdist.init("gloo")
    for step in range(3):
        n = 1000 + 37 * rank + 5 * step + (0 if rank else 400)        # a different length on every rank and step
        vals = ((torch.arange(n) * (rank + 2) + step) % 8).to(torch.uint8)
        got = mdist.gather_ragged(vals, dst=0)
        if rank == 0:
            assert len(got) == world
            for r, t in enumerate(got):
                m = 1000 + 37 * r + 5 * step + (0 if r else 400)
                assert t.shape == (m,) and torch.equal(t, ((torch.arange(m) * (r + 2) + step) % 8).to(torch.uint8))
        else:
            assert got is None
    if rank == 0:
        np.save(out_path, np.array([1]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_ragged_gather_of_per_rank_label_vectors(tmp_path):
    """KPConv / PointPillars at N > 1: every rank's batch has its own number of points / boxes (the ragged
    case): sizes, then a padded gather, trimmed on rank 0."""
    world = 2
    out = str(tmp_path / "ok.npy")
    mp.spawn(_ragged_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert os.path.exists(out)


def test_ragged_gather_single_process_is_identity():
    from ml3d.dist import gather_ragged
    x = torch.arange(7)
    assert gather_ragged(x)[0] is x
