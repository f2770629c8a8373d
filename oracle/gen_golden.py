"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference through
oracle/ref_shim.py) in this container.  Run from the repo root:  python -m oracle.gen_golden

The reference's RandLANet module (ml3d/torch/models/randlanet.py) executes its own PyTorch-CPU
forward on seeded inputs; neighbour indices come from the oracle's knn_search wired into the
reference's DataProcessing.knn_search call path (open3d is not installable).
The GPU box has no /root/reference: tests only read the .npz files written here.
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ops as oops  # noqa: E402
from oracle import randlanet_ref as R  # noqa: E402
from oracle import ref_shim  # noqa: E402
import synth_data  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

KITTI_CFG = dict(num_neighbors=16, num_layers=4, num_classes=19, sub_sampling_ratio=[4, 4, 4, 4],
                 in_channels=3, dim_features=8, dim_output=[16, 64, 128, 256])
SMALL_CFG = dict(num_neighbors=16, num_layers=3, num_classes=8, sub_sampling_ratio=[4, 4, 2],
                 in_channels=6, dim_features=8, dim_output=[16, 32, 64])


def reference_logits(rl, cfg, sd, pts, feats):
    """Run the reference's own transform-style neighbour loop + forward."""
    from ml3d.datasets.utils import DataProcessing  # the reference's, resolved through the shim
    B = pts.shape[0]
    model = rl.RandLANet(**cfg, num_points=pts.shape[1], ignored_label_inds=[0], grid_size=0.06)
    model.load_state_dict(sd)
    model.eval()
    model.device = torch.device("cpu")
    coords, nbrs, pools, ups = [], [], [], []
    pcs = [pts[b] for b in range(B)]
    for i in range(cfg["num_layers"]):      # mirrors randlanet.py:218-229 with the reference's own helper
        nb = [DataProcessing.knn_search(pc, pc, cfg["num_neighbors"]) for pc in pcs]
        n_sub = pcs[0].shape[0] // cfg["sub_sampling_ratio"][i]
        subs = [pc[:n_sub] for pc in pcs]
        up = [DataProcessing.knn_search(s, pc, 1) for s, pc in zip(subs, pcs)]
        coords.append(torch.from_numpy(np.stack(pcs)))
        nbrs.append(torch.from_numpy(np.stack(nb).astype(np.int64)))
        pools.append(torch.from_numpy(np.stack([x[:n_sub] for x in nb]).astype(np.int64)))
        ups.append(torch.from_numpy(np.stack(up).astype(np.int64)))
        pcs = subs
    inputs = {"coords": coords, "neighbor_indices": nbrs, "sub_idx": pools, "interp_idx": ups,
              "features": torch.from_numpy(feats)}
    with torch.no_grad():
        out = model(inputs)
    return out.numpy(), inputs


def main():
    os.makedirs(OUT, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(tempfile.mkdtemp())     # the reference may write caches into the CWD
    rl = ref_shim.reference_modules()
    torch.set_num_threads(max(1, os.cpu_count() or 1))

    # 1. small heterogeneous config, batch 2, extra feature channels
    rng = np.random.default_rng(42)
    pts = synth_data.uniform_cloud(42, 2 * 1024).reshape(2, 1024, 3)
    feats = np.concatenate([pts, rng.random((2, 1024, 3), dtype=np.float32)], 2)
    sd = R.make_state_dict(SMALL_CFG, 101)
    logits, inp = reference_logits(rl, SMALL_CFG, sd, pts, feats)
    mine = R.forward(sd, SMALL_CFG, inp).numpy()
    assert np.abs(mine - logits).max() <= 1e-6, "oracle restatement deviates from the reference module"
    np.savez_compressed(os.path.join(OUT, "randlanet_small.npz"), points=pts, features=feats, logits=logits,
                        weights_seed=101, nbr0=inp["neighbor_indices"][0].numpy().astype(np.int32),
                        interp0=inp["interp_idx"][0].numpy().astype(np.int32))

    # 2. SemanticKITTI widths on a 4096-point lidar-shaped patch
    pts = synth_data.semantickitti_patch(7, 4096)[None]
    sd = R.make_state_dict(KITTI_CFG, 2024)
    logits, inp = reference_logits(rl, KITTI_CFG, sd, pts, pts.copy())
    mine = R.forward(sd, KITTI_CFG, inp).numpy()
    assert np.abs(mine - logits).max() <= 1e-6
    np.savez_compressed(os.path.join(OUT, "randlanet_kitti4096.npz"), points=pts, logits=logits, weights_seed=2024,
                        nbr0=inp["neighbor_indices"][0].numpy().astype(np.int32),
                        nbr3=inp["neighbor_indices"][3].numpy().astype(np.int32),
                        interp0=inp["interp_idx"][0].numpy().astype(np.int32))

    # 3. full-size frame (45056 points): logits of every 64th point + checksums
    pts = synth_data.semantickitti_patch(0, 45056)[None]
    logits, inp = reference_logits(rl, KITTI_CFG, sd, pts, pts.copy())
    mine = R.forward(sd, KITTI_CFG, inp).numpy()
    assert np.abs(mine - logits).max() <= 1e-6
    nb0 = inp["neighbor_indices"][0].numpy()
    np.savez_compressed(os.path.join(OUT, "randlanet_kitti45056.npz"), frame_id=0, weights_seed=2024,
                        logits_every64=logits[:, ::64], argmax=logits.argmax(-1).astype(np.int8),
                        points_sum=pts.astype(np.float64).sum(), nbr0_rowsum=nb0.sum(-1).astype(np.int64)[0, ::16],
                        nbr0_checksum=np.int64((nb0.astype(np.int64) * (np.arange(16) + 1)).sum()))
    os.chdir(cwd)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
