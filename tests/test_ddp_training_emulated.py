"""CPU: data-parallel TRAINING of the native model classes -- ``torch.nn.parallel.DistributedDataParallel`` around RandLANet, KPFCNN
and PointPillars in ``.train()`` mode, world size 2 over gloo, the HIP library emulated (tests/emu_runtime.py) -- the multi-GPU form
of SURVEY.md §8 f4 (the reference wraps its detection models the same way, object_detection.py:300-310; one process per GPU
over RCCL on the real thing).  Each rank runs its own batch through forward + loss + backward; DDP's all-reduce must leave both
ranks with the SAME gradients, equal to the mean of the two single-process gradients (to 2e-4 of each tensor's largest entry),
for every parameter."""
import os
import socket
import subprocess
import sys

import pytest

import emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not emu.available(), reason="clang++ for the host emulator not found")

_WORKER = r'''
import os, sys
ROOT, rank, port, which, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import torch.distributed as dist
import emu_runtime
emu_runtime.install("ml3d")
os.environ.update(RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
dist.init_process_group("gloo", rank=rank, world_size=2)
import synth_data, synth_weights
torch.manual_seed(0)


def randla(r):
    from ml3d.torch.models import RandLANet
    from oracle import randlanet_ref as R
    cfg = dict(num_neighbors=16, num_layers=2, num_points=512, num_classes=6, sub_sampling_ratio=[4, 4], in_channels=3,
               dim_features=8, dim_output=[16, 32], ignored_label_inds=[0])
    m = RandLANet(**cfg, device="cpu")
    m.load_state_dict(R.make_state_dict(cfg, 9))
    m.train()
    m.fc1[2].eval()                     # (dropout masks differ per process by design)
    pts = np.stack([synth_data.semantickitti_patch(40 + 2 * r + b, 512) for b in range(2)])
    labels = torch.from_numpy(np.random.default_rng(r).integers(0, 7, (2, 512)))
    loss_obj = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()

    def step(model, raw):
        t = torch.from_numpy(pts)
        logits = model({"coords": [t], "features": t.clone()})
        return raw.get_loss(loss_obj, logits, {"data": {"labels": labels}}, "cpu")[0]
    return m, step


def pillars(r):
    from ml3d.torch.models import PointPillars
    from oracle import pointpillars_ref as P
    from oracle.gen_golden_loss import loss_inputs
    from oracle.gen_golden_train import PP_LOSS_CFG
    cfg = synth_weights.POINTPILLARS_SMALL_CFG
    m = PointPillars(device="cpu", loss=PP_LOSS_CFG, **cfg)
    m.load_state_dict(P.make_state_dict(cfg, 21))
    m.train()
    clouds = [P.crop_for_cfg(synth_data.kitti_sweep(90 + 2 * r + i), cfg)[:, :3].copy() for i in range(2)]
    _, boxes, labels = loss_inputs(cfg, 30 + r, (3, 5))

    class In:
        point = [torch.from_numpy(c) for c in clouds]
        bboxes = boxes
    In.labels = labels

    def step(model, raw):
        return sum(raw.get_loss(model(In), In).values())
    return m, step


def fail_fast(exc_type, exc, tb):           # (a rank that raises must not leave its peer waiting in a collective)
    import traceback
    traceback.print_exception(exc_type, exc, tb)
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(1)


sys.excepthook = fail_fast
def kpconv(r):
    from ml3d.torch.dataloaders import kpconv_input_features
    from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
    from oracle import kpconv_ref as K
    from oracle.gen_golden_train import TRAIN_CFG
    cfg = dict(TRAIN_CFG, first_features_dim=32)
    m = KPFCNN(**cfg, device="cpu")
    m.load_state_dict(K.make_state_dict(cfg, 77))
    m.train()
    spheres = [synth_data.toronto3d_sphere(63 + 2 * r + i, 350) for i in range(2)]
    rng = np.random.default_rng(5 + r)
    cols = np.concatenate([np.concatenate([s_, rng.random((len(s_), 3), dtype=np.float32)], 1) for s_ in spheres])
    pts = np.concatenate(spheres)
    np.random.seed(31 + r)
    batch = KPConvBatch(pts, [len(s_) for s_ in spheres], cfg, features=kpconv_input_features(pts, cols, cfg["in_features_dim"]).astype(np.float32),
                        device="cpu")
    batch.labels = torch.from_numpy(rng.integers(0, 9, len(pts)).astype(np.int64))
    loss_obj = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()

    def step(model, raw):
        return raw.get_loss(loss_obj, model(batch), {"data": batch}, "cpu")[0]
    return m, step


make = {"randlanet": randla, "pointpillars": pillars, "kpconv": kpconv}[which]
# single-process gradients of BOTH ranks' batches (what the all-reduce must average)
singles = []
for r in range(2):
    m, step = make(r)
    step(m, m).backward()
    singles.append({k: v.grad.clone() for k, v in m.named_parameters() if v.grad is not None})
m, step = make(rank)
ddp = torch.nn.parallel.DistributedDataParallel(m)
step(ddp, m).backward()
worst = 0.0
for k, v in m.named_parameters():
    if v.grad is None:
        assert k not in singles[0], k
        continue
    want = 0.5 * (singles[0][k] + singles[1][k])
    # (relative to the tensor's largest entry, floored: the biases in front of a BatchNorm have a ZERO gradient, 1e-8 of noise)
    worst = max(worst, float((v.grad - want).abs().max()) / max(1e-3, float(want.abs().max())))
    other = [torch.empty_like(v.grad) for _ in range(2)]
    dist.all_gather(other, v.grad.contiguous())
    assert torch.equal(other[0], other[1]), k            # both ranks hold the same reduced gradient
assert worst <= 2e-4, worst
if rank == 0:
    open(out, "w").write("ok %d parameters, worst relative deviation from the mean of the single-process gradients %.2g" % (len(singles[0]), worst))
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("which", ["randlanet", "kpconv", "pointpillars"])
def test_ddp_over_gloo_averages_the_gradients_of_the_native_training_forwards(tmp_path, which):
    emu.lib()
    port, out = str(_free_port()), str(tmp_path / "ok.txt")
    procs = [subprocess.Popen([sys.executable, "-c", _WORKER, ROOT, str(r), port, which, out], cwd="/tmp",
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=600)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    assert open(out).read().startswith("ok")
