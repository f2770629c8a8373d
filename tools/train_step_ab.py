"""GPU box: one training step (forward + loss + backward) of the two segmentation models at their YAML sizes, ML3D_TRAIN_OPS=hip
(csrc/train.hip: Linear / BatchNorm / gathers / fused attention stages, both passes) against =torch (those modules on torch's autograd):
ms per step (median of 5 after 2 warm-up steps) and the peak of torch's allocator during a step.
usage: python tools/train_step_ab.py [randlanet|kpconv] [batch] [paths, e.g. hip or torch,hip]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth_data
import synth_weights as W

which = sys.argv[1] if len(sys.argv) > 1 else "randlanet"
dev = torch.device("cuda:0")
loss_obj = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()
if which == "randlanet":
    from ml3d.torch.models import RandLANet
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4                 # randlanet_semantickitti.yml:38 batch_size 4
    cfg = dict(W.RANDLANET_SEMANTICKITTI_CFG)
    m = RandLANet(**cfg, device=dev)
    m.to(dev)
    pts = torch.from_numpy(np.stack([synth_data.semantickitti_patch(i, cfg["num_points"]) for i in range(B)])).to(dev)
    labels = torch.randint(1, cfg["num_classes"], (B, cfg["num_points"]))
    nbr, itp = m.neighbor_pyramid(pts)
    inputs = {"coords": [pts], "features": pts.clone(), "neighbor_indices": nbr, "interp_idx": itp}

    def step():
        logits = m(inputs)
        loss, _, _ = m.get_loss(loss_obj, logits, {"data": {"labels": labels}}, dev)
        loss.backward()
        return float(loss)
    what = "RandLANet SemanticKITTI, %d x %d points" % (B, cfg["num_points"])
else:
    from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cfg = dict(W.TORONTO3D_CFG)
    m = KPFCNN(**cfg, device=dev)
    m.load_state_dict(W.kpconv_state_dict(cfg, 2024))
    m.to(dev)
    spheres = [synth_data.toronto3d_sphere(i) for i in range(B)]
    np.random.seed(0)
    batch = KPConvBatch(torch.from_numpy(np.concatenate(spheres)).to(dev), [len(s) for s in spheres], cfg, device=dev)
    batch.labels = torch.randint(1, 9, (sum(len(s) for s in spheres),))

    def step():
        logits = m(batch)
        loss, _, _ = m.get_loss(loss_obj, logits, {"data": batch}, dev)
        loss.backward()
        return float(loss)
    what = "KPFCNN Toronto3D, %d spheres, %d points" % (B, sum(len(s) for s in spheres))
m.train()
print(what)
paths = tuple(sys.argv[3].split(",")) if len(sys.argv) > 3 else ("torch", "hip", "torch", "hip")
for path in paths:
    os.environ["ML3D_TRAIN_OPS"] = path
    times = []
    for it in range(7):
        m.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        if it == 2:
            torch.cuda.reset_peak_memory_stats()
        t0 = time.perf_counter()
        loss = step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    print("ML3D_TRAIN_OPS=%-5s  step %.1f ms (median of 5; %s)  peak allocated %.2f GB  loss %.5f"
          % (path, float(np.median(times[2:])), " ".join("%.1f" % t for t in times[2:]), torch.cuda.max_memory_allocated() / 2 ** 30, loss), flush=True)
