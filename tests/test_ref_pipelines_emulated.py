"""CPU, with the reference checkout: the REFERENCE's own ``run_inference`` pipelines (semantic_segmentation.py:122-187,
object_detection.py:46-75), dataloader, samplers and batchers drive this repository's three model classes — resolved through
the reference's registry from the unchanged YAML configs — with every primitive executed by the HOST EMULATION of the HIP
library (tests/emu_runtime.py: the same .hip sources, test infrastructure), and the results are compared with the reference's
own PyTorch-CPU models on the oracle ops for the same seeds / cloud / weights.  Row b2 ("configs and pipelines unchanged").
The same driver runs on a GPU box at the YAML sizes (tools/gpu_ref_pipelines.sh -> profiles/r03_pipeline_*.log)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("ML3D_REFERENCE_ROOT", "/root/reference")
DRIVER = os.path.join(ROOT, "tools", "ref_pipelines.py")

pytestmark = [pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "ml3d")), reason="needs the reference checkout"),
              pytest.mark.skipif(not emu.available(), reason="clang++ for the host emulator not found")]


def _run(args, out):
    env = dict(os.environ)
    env.pop("OPEN3D_ML_ROOT", None)
    r = subprocess.run([sys.executable, DRIVER, "--ref", REF, "--small", "--out", out] + args, cwd="/tmp", env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    return r.stdout


def test_reference_pipelines_drive_the_native_models_and_match_the_reference_cpu_path(tmp_path):
    out = str(tmp_path)
    emu.lib()                                              # build the emulated library once, outside the timed subprocesses
    ref_log = _run(["--side", "reference"], out)
    nat_log = _run(["--side", "native", "--emu"], out)
    for name in ("randlanet", "kpconv", "pointpillars", "kpconv_deform"):
        assert "[%s/native] model class ml3d_amd.torch.models" % name in nat_log        # the MI355X-native class ...
        assert "pipeline class ml3d.torch.pipelines" in nat_log                         # ... under the reference's pipeline
        assert "[%s/reference] model class ml3d.torch.models" % name in ref_log
    for name in ("randlanet", "kpconv", "kpconv_deform"):       # (kpconv_deform: kpconv_parislille3d.yml, five deformable blocks)
        a = np.load(os.path.join(out, "%s_native_small.npz" % name))
        b = np.load(os.path.join(out, "%s_reference_small.npz" % name))
        assert a["predict_labels"].shape == b["predict_labels"].shape and a["predict_labels"].size > 5000
        assert (a["predict_labels"] == b["predict_labels"]).mean() >= 0.9999
        assert np.abs(a["predict_scores"] - b["predict_scores"]).max() <= 2.0 ** -9      # float16 vote accumulator
    a = np.load(os.path.join(out, "pointpillars_native_small.npz"))
    b = np.load(os.path.join(out, "pointpillars_reference_small.npz"))
    assert a["boxes"].shape == b["boxes"].shape and a["boxes"].shape[0] > 10
    assert np.array_equal(a["labels"], b["labels"])
    assert (np.abs(a["boxes"] - b["boxes"]) / np.maximum(1.0, np.abs(b["boxes"]))).max() <= 1e-4
    assert np.abs(a["scores"] - b["scores"]).max() <= 1e-4


def test_reference_run_pipeline_script_trains_the_native_models_like_the_reference(tmp_path):
    """The reference's UNCHANGED ``scripts/run_pipeline.py torch -c <yaml> --split train`` (SemanticSegmentation.run_train,
    semantic_segmentation.py:317-470: dataloader + transform + batcher, get_optimizer, train-mode forward, get_loss, backward,
    optimizer.step, validation, checkpoint) on the native RandLA-Net and KPFCNN -- shrunken YAML sizes, the library emulated --
    against the reference side: the first step's loss to 2e-4, later steps and the epoch summary within the drift of a few
    optimisation steps (tools/run_pipeline_e2e.py --compare; the MI355X run at the YAML sizes is
    profiles/r04_run_pipeline_train_e2e.log).  The later-step bounds are 6e-2 / 8e-2 here: with 500-point batches the deepest KPConv
    levels hold 1-3 points, BatchNorm over them amplifies a float32 rounding difference ~2500x within three SGD steps -- the torch-autograd
    path (bit-identical CPU kernels to the reference's, 1e-6 apart after one forward) drifts 3.4e-3, the HIP training ops (their own
    summation orders, 1e-5 apart) 2.5e-2; gradients of both are pinned to the reference at 1e-3 per tensor by tests/test_emulated_training.py,
    and on batches with >= 10 points per level the two paths stay within 1e-6 over four steps (profiles/r05_train_e2e_sensitivity.log)."""
    work, out = str(tmp_path / "work"), str(tmp_path / "out")
    emu.lib()
    tool = os.path.join(ROOT, "tools", "run_pipeline_e2e.py")
    env = dict(os.environ)
    env.pop("OPEN3D_ML_ROOT", None)
    for side in (["--side", "reference"], ["--side", "native", "--emu"]):
        r = subprocess.run([sys.executable, tool, "--family", "all", "--split", "train", "--small", "--ref", REF, "--work", work,
                            "--out", out] + side, cwd="/tmp", env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-3000:]
    r = subprocess.run([sys.executable, tool, "--compare", out, out, "--later-tol", "6e-2", "--summary-tol", "8e-2"], cwd="/tmp", env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "[compare] OK" in r.stdout, r.stdout[-4000:]
    assert "randlanet train.log: 4 step losses" in r.stdout and "kpconv train.log: 4 step losses" in r.stdout
