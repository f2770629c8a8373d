"""SemanticKITTI sweep readers / prediction writer against golden vectors produced by the REAL
reference loaders (oracle/gen_golden_io.py), and the GPU raw-sweep front end against the oracle."""
import os

import numpy as np
import pytest

from ml3d.datasets import SemanticKITTIFormat, load_label_kitti, load_pc_kitti, preprocess_sweep
from oracle import ops as oops
import synth_data


@pytest.fixture()
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "semantickitti_io.npz"))


def _write_sweep(tmp_path, gold):
    seq = tmp_path / "dataset" / "sequences" / "08"
    (seq / "velodyne").mkdir(parents=True)
    (seq / "labels").mkdir()
    pc = str(seq / "velodyne" / "000123.bin")
    gold["scan"].tofile(pc)
    gold["raw_labels"].tofile(str(seq / "labels" / "000123.label"))
    return pc, str(seq / "labels" / "000123.label")


def test_lookup_tables_match_the_reference(gold):
    fmt = SemanticKITTIFormat()
    assert np.array_equal(fmt.remap_lut, gold["remap_lut"])
    assert np.array_equal(fmt.remap_lut_val, gold["remap_lut_val"])


def test_readers_match_the_reference(tmp_path, gold):
    pc, lab = _write_sweep(tmp_path, gold)
    fmt = SemanticKITTIFormat()
    pts = load_pc_kitti(pc)
    assert pts.dtype == np.float32 and np.array_equal(pts, gold["points"])
    labels = load_label_kitti(lab, fmt.remap_lut_val)
    assert labels.dtype == np.int32 and np.array_equal(labels, gold["labels"])
    d = fmt.get_data(pc, split="validation")
    assert np.array_equal(d["point"], gold["data_point"]) and np.array_equal(d["feat"], gold["data_feat"])
    assert np.array_equal(d["label"], gold["data_label"])
    assert fmt.get_attr(pc, "validation")["name"] == str(gold["attr_name"])


def test_missing_labels(tmp_path, gold):
    pc, lab = _write_sweep(tmp_path, gold)
    os.remove(lab)
    fmt = SemanticKITTIFormat()
    d = fmt.get_data(pc, split="test")                     # test sweeps have no labels: zeros
    assert d["label"].shape == (gold["scan"].shape[0],) and not d["label"].any()
    with pytest.raises(FileNotFoundError):
        fmt.get_data(pc, split="validation")


def test_prediction_writer_matches_the_reference(tmp_path, gold):
    fmt = SemanticKITTIFormat(test_result_folder=str(tmp_path / "test"),
                              ignored_label_inds=[int(x) for x in gold["ignored_label_inds"]])
    attr = {"name": str(gold["attr_name"])}
    assert not fmt.is_tested(attr)
    pred = gold["predict_labels"].copy()
    path = fmt.save_test_result({"predict_labels": pred}, attr)
    assert path.endswith(os.path.join("sequences", "08", "predictions", "000123.label")) and fmt.is_tested(attr)
    assert np.array_equal(np.fromfile(path, dtype=np.uint32), gold["saved_labels"])
    assert np.array_equal(pred, gold["predict_labels"])    # the caller's array is not modified


def test_writer_reader_round_trip(tmp_path):
    """write raw ids -> read with the raw->training table: every valid class index comes back shifted by the one
    ignored id (size-independent property of the two tables)."""
    fmt = SemanticKITTIFormat(test_result_folder=str(tmp_path))
    pred = np.arange(19).repeat(3)
    path = fmt.save_test_result({"predict_labels": pred}, {"name": "11_000007"})
    assert np.array_equal(load_label_kitti(path, fmt.remap_lut_val), pred + 1)


def test_preprocess_sweep_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_preprocess_sweep_matches_oracle")
    with pytest.raises(RuntimeError):
        preprocess_sweep({"point": np.zeros((10, 3), np.float32)}, 0.06, "test")


@pytest.mark.gpu
def test_preprocess_sweep_matches_oracle():
    import torch
    assert torch.cuda.is_available()
    sweep = synth_data.lidar_sweep(3, n_beams=32, n_azimuth=1024)
    rng = np.random.default_rng(5)
    data = {"point": sweep[:, :3], "feat": rng.random((sweep.shape[0], 1), dtype=np.float32),
            "label": rng.integers(0, 20, sweep.shape[0]).astype(np.int32)}
    out = preprocess_sweep(data, grid_size=0.06, split="test")
    sp, sf, sl = oops.subsample(data["point"], features=data["feat"], classes=data["label"], sampleDl=0.06)
    assert np.array_equal(out["point"], sp) and np.array_equal(out["feat"], sf) and np.array_equal(out["label"], sl)
    assert np.array_equal(out["proj_inds"], oops.knn_search(sp, data["point"], 1).reshape(-1))
    assert out["proj_inds"].dtype == np.int32 and out["point"].shape[0] < sweep.shape[0]
    nofeat = preprocess_sweep({"point": sweep[:, :3], "label": data["label"]}, grid_size=0.06, split="training")
    assert nofeat["feat"] is None and "proj_inds" not in nofeat and np.array_equal(nofeat["point"], sp)
