"""Synthetic dataset DIRECTORIES in the on-disk layouts the reference's dataset classes read, so that the reference's own
``scripts/run_pipeline.py --split test`` can be driven end to end without the real datasets (no network here):

  * Semantic3D (ml3d/datasets/semantic3d.py:76-90, 219-250): ``<dir>/<name>.txt`` rows ``x y z intensity r g b``; a file
    WITHOUT a sibling ``.labels`` is a test cloud, one with it a training cloud.
  * KITTI (ml3d/datasets/kitti.py:58-73, 94-178, 268-290): ``{training,testing}/velodyne/%06d.bin`` (float32 x y z i),
    ``calib/%06d.txt`` (P0..P3, R0_rect, Tr_velo_to_cam), ``label_2/%06d.txt``.

Bench / test data only; pure numpy on synth_data.py's seeded generators."""
import os

import numpy as np

import synth_data


def write_semantic3d(root, n_test=1, n_train=0, half=9.0, density=0.35, seed=40, val_names=()):
    """-> list of written cloud names.  A ``2 half`` m urban tile per cloud (~120 k points at the defaults).  ``val_names``: the
    LAST labelled clouds get these file names (entries of the YAML's ``val_files``: the validation split, semantic3d.py:92-99)."""
    os.makedirs(root, exist_ok=True)
    names = []
    for i in range(n_test + n_train):
        d = synth_data.toronto3d_tile(seed + i, half=half, density=density)
        pts, rgb = d["point"], d["feat"]
        inten = (np.abs(pts[:, 2]) * 40.0 % 255.0).astype(np.float32)
        name = "synth%02d_xyz_intensity_rgb" % i
        j = i - (n_test + n_train - len(val_names))
        if 0 <= j < len(val_names) and i >= n_test:
            name = val_names[j]
        rows = np.concatenate([pts, inten[:, None], np.round(rgb)], 1)
        np.savetxt(os.path.join(root, name + ".txt"), rows, fmt="%.3f %.3f %.3f %d %d %d %d")
        if i >= n_test:
            np.savetxt(os.path.join(root, name + ".labels"), d["label"].astype(np.int32), fmt="%d")
        names.append(name)
    return names


# KITTI's usual rig: camera x = -lidar y, camera y = -lidar z, camera z = lidar x (+ the mounting offsets)
_TR_VELO_TO_CAM = np.array([[0.0, -1.0, 0.0, 0.0], [0.0, 0.0, -1.0, -0.08], [1.0, 0.0, 0.0, -0.27]], np.float32)
_P2 = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]], np.float32)


def _calib_text():
    def row(name, m):
        return name + ": " + " ".join("%.6e" % v for v in np.asarray(m).reshape(-1))
    return "\n".join([row("P0", _P2), row("P1", _P2), row("P2", _P2), row("P3", _P2), row("R0_rect", np.eye(3)),
                      row("Tr_velo_to_cam", _TR_VELO_TO_CAM), row("Tr_imu_to_velo", np.eye(4)[:3])]) + "\n"


def _label_lines(rng, n):
    """n ground-truth objects in KITTI's camera-frame text format (kitti.py:104-131 reads fields 8..14 + the class)."""
    lines = []
    for _ in range(n):
        cls, (h, w, l) = [("Car", (1.5, 1.6, 3.9)), ("Pedestrian", (1.7, 0.6, 0.8)), ("Cyclist", (1.7, 0.6, 1.76))][int(rng.integers(0, 3))]
        x_l, y_l = rng.uniform(8, 45), rng.uniform(-8, 8)                     # lidar frame, in front of the car
        cam = _TR_VELO_TO_CAM @ np.array([x_l, y_l, -1.73, 1.0])              # bottom centre in the camera frame
        lines.append("%s 0.00 0 0.00 0.00 0.00 50.00 50.00 %.2f %.2f %.2f %.2f %.2f %.2f %.2f"
                     % (cls, h, w, l, cam[0], cam[1], cam[2], rng.uniform(-1.5, 1.5)))
    return "\n".join(lines) + "\n"


def write_kitti(root, n_test=2, n_train=2, seed=60):
    """``training`` holds n_train sweeps with labels (indices 0.. — the validation split is idx >= cfg.val_split, so pass
    ``--dataset.val_split 0`` to validate on them), ``testing`` n_test sweeps without."""
    rng = np.random.default_rng(seed)
    for split, n in (("training", n_train), ("testing", n_test)):
        for sub in ("velodyne", "calib", "label_2"):
            os.makedirs(os.path.join(root, split, sub), exist_ok=True)
        for i in range(n):
            sweep = synth_data.kitti_sweep(seed + (0 if split == "training" else 100) + i)
            sweep.astype(np.float32).tofile(os.path.join(root, split, "velodyne", "%06d.bin" % i))
            open(os.path.join(root, split, "calib", "%06d.txt" % i), "w").write(_calib_text())
            if split == "training":
                open(os.path.join(root, split, "label_2", "%06d.txt" % i), "w").write(_label_lines(rng, 6))
    return root
