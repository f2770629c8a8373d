"""SemanticKITTI on-disk formats and the raw-sweep front end of the RandLA-Net path.

Mirrors, by behaviour:
  * ``DataProcessing.load_pc_kitti`` / ``load_label_kitti``    (ml3d/datasets/utils/dataprocessing.py:69-85)
  * ``SemanticKITTI.__init__`` label look-up tables, ``save_test_result`` and ``is_tested``
                                                               (ml3d/datasets/semantickitti.py:76-100, 142-192)
  * ``SemanticKITTISplit.get_data`` / ``get_attr``             (ml3d/datasets/semantickitti.py:268-300)
  * ``RandLANet.preprocess``                                   (ml3d/torch/models/randlanet.py:118-152)

File formats: ``velodyne/NNNNNN.bin`` = float32 x, y, z, remission per point; ``labels/NNNNNN.label`` = uint32 per
point, semantic id in the low 16 bits and instance id in the high 16; predictions are written back as uint32 RAW ids.
The raw<->training id maps below are the dataset's published ``learning_map`` / ``learning_map_inv``.

``preprocess_sweep`` replaces the reference's CPU grid subsampling + scikit-learn KDTree with ``ml3d.ops.subsample``
(voxel barycentres, mean features, majority label; stable radix sort) and ``ml3d.ops.knn_search`` (exact 1-NN) on
the GPU; it needs the HIP library and a GPU and raises otherwise (no CPU fallback)."""
import os

import numpy as np

# SemanticKITTI raw id -> training id (0 = unlabeled / ignored) and back
LEARNING_MAP = {
    0: 0, 1: 0, 10: 1, 11: 2, 13: 5, 15: 3, 16: 5, 18: 4, 20: 5, 30: 6, 31: 7, 32: 8, 40: 9, 44: 10, 48: 11, 49: 12,
    50: 13, 51: 14, 52: 0, 60: 9, 70: 15, 71: 16, 72: 17, 80: 18, 81: 19, 99: 0, 252: 1, 253: 7, 254: 6, 255: 8,
    256: 5, 257: 5, 258: 4, 259: 5,
}
LEARNING_MAP_INV = {
    0: 0, 1: 10, 2: 11, 3: 15, 4: 18, 5: 20, 6: 30, 7: 31, 8: 32, 9: 40, 10: 44, 11: 48, 12: 49, 13: 50, 14: 51,
    15: 70, 16: 71, 17: 72, 18: 80, 19: 81,
}


def _lut(mapping):
    lut = np.zeros(max(mapping) + 100, dtype=np.int32)         # (the reference pads its tables by 100 entries too)
    lut[list(mapping.keys())] = list(mapping.values())
    return lut


def load_pc_kitti(pc_path):
    """[N, 4] float32 (x, y, z, remission) of one ``.bin`` sweep."""
    return np.fromfile(pc_path, dtype=np.float32).reshape((-1, 4))


def load_label_kitti(label_path, remap_lut):
    """Training ids [N] int32 of one ``.label`` file: semantic id = low 16 bits, remapped through ``remap_lut``."""
    label = np.fromfile(label_path, dtype=np.uint32).reshape((-1))
    sem_label = label & 0xFFFF
    inst_label = label >> 16
    assert ((sem_label + (inst_label << 16) == label).all())
    return remap_lut[sem_label].astype(np.int32)


class SemanticKITTIFormat:
    """Look-up tables + file naming of the SemanticKITTI dataset object, without the split / cache machinery."""

    def __init__(self, test_result_folder="./test", ignored_label_inds=(0,)):
        self.test_result_folder = test_result_folder
        self.ignored_label_inds = list(ignored_label_inds)
        self.remap_lut_val = _lut(LEARNING_MAP)                 # raw -> training
        self.remap_lut = _lut(LEARNING_MAP_INV)                 # training -> raw

    # -- reading ------------------------------------------------------------------------------------------
    def get_data(self, pc_path, split="test"):
        points = load_pc_kitti(pc_path)
        d, f = os.path.split(pc_path)
        label_path = os.path.join(d, "../labels", f[:-4] + ".label")
        if not os.path.exists(label_path):
            if split not in ["test", "all"]:
                raise FileNotFoundError(f" Label file {label_path} not found")
            labels = np.zeros(np.shape(points)[0], dtype=np.int32)
        else:
            labels = load_label_kitti(label_path, self.remap_lut_val).astype(np.int32)
        return {"point": points[:, 0:3], "feat": points[:, 3:], "label": labels}

    @staticmethod
    def get_attr(pc_path, split="test"):
        d, f = os.path.split(pc_path)
        _, seq = os.path.split(os.path.split(d)[0])
        return {"idx": None, "name": "{}_{}".format(seq, f[:-4]), "path": pc_path, "split": split}

    # -- writing ------------------------------------------------------------------------------------------
    def _store_path(self, name):
        name_seq, name_points = name.split("_")
        save_path = os.path.join(self.test_result_folder, "sequences", name_seq, "predictions")
        return save_path, os.path.join(save_path, name_points + ".label")

    def is_tested(self, attr):
        return os.path.exists(self._store_path(attr["name"])[1])

    def save_test_result(self, results, attr):
        """``results['predict_labels']`` = class indices over the VALID classes; the ignored training ids are
        re-inserted (every index >= an ignored id shifts up by one), mapped to raw ids, written as uint32."""
        pred = np.array(results["predict_labels"]).copy()
        for ign in self.ignored_label_inds:
            pred[pred >= ign] += 1
        save_path, store_path = self._store_path(attr["name"])
        os.makedirs(save_path, exist_ok=True)
        self.remap_lut[pred].astype(np.uint32).tofile(store_path)
        return store_path


def preprocess_sweep(data, grid_size=0.06, split="test", device="cuda:0"):
    """``RandLANet.preprocess`` on the GPU ops: grid-subsample the raw sweep (points barycentre, features mean,
    labels majority) and, for test splits, project every raw point onto its nearest subsampled point.
    Returns numpy arrays like the reference (``point``, ``feat``, ``label``[, ``proj_inds``]); the KDTree of the
    reference is not produced — the path queries neighbours through ``ml3d.ops`` instead."""
    import torch

    from .. import ops

    if not torch.cuda.is_available():
        raise RuntimeError("preprocess_sweep needs a GPU (the HIP library has no CPU fallback)")
    dev = torch.device(device)
    points = np.array(data["point"][:, 0:3], dtype=np.float32)
    labels = (np.zeros((points.shape[0],), dtype=np.int32) if data.get("label") is None
              else np.array(data["label"], dtype=np.int32).reshape((-1,)))
    feat = None if data.get("feat") is None else np.array(data["feat"], dtype=np.float32)
    p = torch.from_numpy(points).to(dev)
    lab = torch.from_numpy(labels).to(dev)
    if feat is None:
        sub_points, sub_labels = ops.subsample(p, classes=lab, sampleDl=grid_size)
        sub_feat = None
    else:
        sub_points, sub_feat, sub_labels = ops.subsample(p, features=torch.from_numpy(feat).to(dev), classes=lab,
                                                         sampleDl=grid_size)
    out = {"point": sub_points.cpu().numpy(), "feat": None if sub_feat is None else sub_feat.cpu().numpy(),
           "label": sub_labels.cpu().numpy().astype(np.int32)}
    if split in ["test", "testing"]:
        idx = ops.knn_search(sub_points, p, 1)
        idx = idx[0] if isinstance(idx, (tuple, list)) else idx
        out["proj_inds"] = idx.reshape(-1).cpu().numpy().astype(np.int32)
    return out
