"""Golden vectors for the SemanticKITTI file formats, produced by the REAL reference loaders /
writer (``/root/reference/ml3d/datasets``) on synthetic sweeps.  Oracle tooling: runs in the build container only.

    python oracle/gen_golden_io.py      ->  tests/golden/semantickitti_io.npz
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    ref_shim.install()
    sys.path.insert(0, ref_shim.REF_ROOT)
    from ml3d.datasets.semantickitti import SemanticKITTI, SemanticKITTISplit  # the reference
    from ml3d.datasets.utils import DataProcessing

    rng = np.random.default_rng(77)
    tmp = tempfile.mkdtemp()
    seq = os.path.join(tmp, "dataset", "sequences", "08")
    os.makedirs(os.path.join(seq, "velodyne"))
    os.makedirs(os.path.join(seq, "labels"))
    n = 5000
    scan = rng.normal(0, 20, (n, 4)).astype(np.float32)
    scan[:, 3] = rng.random(n, dtype=np.float32)
    ds = SemanticKITTI(dataset_path=os.path.join(tmp, "dataset"), cache_dir=os.path.join(tmp, "cache"),
                       test_result_folder=os.path.join(tmp, "test"), use_cache=False)
    raw_ids = np.array(sorted(k for k in range(len(ds.remap_lut_val)) if k < 260), dtype=np.uint32)
    sem = rng.choice(np.array([0, 1, 10, 11, 13, 15, 16, 18, 20, 30, 31, 32, 40, 44, 48, 49, 50, 51, 52, 60, 70, 71, 72,
                               80, 81, 99, 252, 253, 254, 255, 256, 257, 258, 259], dtype=np.uint32), n)
    inst = rng.integers(0, 1000, n).astype(np.uint32)
    raw = (sem | (inst << 16)).astype(np.uint32)
    pc_path = os.path.join(seq, "velodyne", "000123.bin")
    scan.tofile(pc_path)
    raw.tofile(os.path.join(seq, "labels", "000123.label"))

    pts = DataProcessing.load_pc_kitti(pc_path)
    lab = DataProcessing.load_label_kitti(os.path.join(seq, "labels", "000123.label"), ds.remap_lut_val)
    split = SemanticKITTISplit.__new__(SemanticKITTISplit)           # get_data / get_attr without the split listing
    split.path_list = [pc_path]
    split.split = "validation"
    split.remap_lut_val = ds.remap_lut_val
    d = split.get_data(0)
    a = split.get_attr(0)

    pred = rng.integers(0, 19, n).astype(np.int64)                   # class indices over the 19 valid classes
    ds.save_test_result({"predict_labels": pred.copy()}, a)
    saved = np.fromfile(os.path.join(tmp, "test", "sequences", "08", "predictions", "000123.label"), dtype=np.uint32)

    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "semantickitti_io.npz"), scan=scan, raw_labels=raw, points=pts, labels=lab,
                        data_point=d["point"], data_feat=d["feat"], data_label=d["label"], attr_name=a["name"],
                        remap_lut=ds.remap_lut, remap_lut_val=ds.remap_lut_val, raw_ids=raw_ids,
                        predict_labels=pred, saved_labels=saved,
                        ignored_label_inds=np.array(ds.cfg.ignored_label_inds, dtype=np.int64))
    print("written", os.path.join(OUT, "semantickitti_io.npz"), a["name"], saved[:8], lab[:8])


if __name__ == "__main__":
    main()
