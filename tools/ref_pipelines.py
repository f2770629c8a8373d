#!/usr/bin/env python3
"""Drive the REFERENCE's own inference pipelines (``SemanticSegmentation.run_inference``,
ml3d/torch/pipelines/semantic_segmentation.py:122-187; ``ObjectDetection.run_inference``, object_detection.py:46-75) with
the reference's unchanged YAML configs, once per SIDE, on one seeded synthetic cloud:

  --side native     ``open3d`` = this repository's shim (open3d-ml_amd/open3d), ``OPEN3D_ML_ROOT`` = the checkout: the
                    pipeline, dataloader, samplers and batchers are the reference's files, the three model classes come
                    from the registry — i.e. the MI355X-native ones — and every primitive is a HIP kernel.  Needs a GPU
                    (or ``--emu``: the host emulation of the library, tests/emu_runtime.py, for debugging host glue here).
  --side reference  the reference's own model classes on PyTorch-CPU with the oracle's C ops standing in for the
                    un-installable ``open3d`` wheel (oracle/ref_shim.py) — the "reference PyTorch-CPU path" of north_star.

Both sides use the same seeds (python ``random``, ``numpy.random``, torch), the same raw cloud and the same state_dict, and
write ``<out>/<model>_<side>.npz``.  ``--compare A B`` prints the agreement.  A checkout is needed on both sides
(``--ref``, default $OPEN3D_ML_ROOT or /root/reference); on a GPU box it is unpacked from a scratch tarball by
tools/gpu_ref_pipelines.sh and never enters the repository.

``--small`` shrinks the configs (num_points / radii / canvas) so that the emulator finishes in seconds; without it the
YAML sizes run (45 056-point patches, in_radius 4.0 m, the KITTI range).
"""
import argparse
import copy
import os
import random
import sys
import tempfile
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "open3d-ml_amd")

MODELS = {"randlanet": "randlanet_semantickitti.yml", "kpconv": "kpconv_toronto3d.yml", "pointpillars": "pointpillars_kitti.yml",
          "kpconv_deform": "kpconv_parislille3d.yml"}        # (five deformable blocks, kpconv_parislille3d.yml:28-32)


def _stub_tensorboard():
    try:
        import torch.utils.tensorboard  # noqa: F401
    except Exception:
        import torch.utils as tu

        class _SW:
            def __init__(self, *a, **k):
                pass

            def __getattr__(self, n):
                return lambda *a, **k: None

        tb = types.ModuleType("torch.utils.tensorboard")
        tb.SummaryWriter = _SW
        sys.modules["torch.utils.tensorboard"] = tb
        tu.tensorboard = tb


def setup(side, ref, emu):
    """-> (utils module of the checkout, device string)"""
    sys.path.insert(0, ROOT)
    stubs = os.path.join(ROOT, "tests", "stubs")
    if side == "native":
        os.environ["OPEN3D_ML_ROOT"] = ref
        for p in (stubs, os.path.join(ROOT, "tests"), PKG):
            if p not in sys.path:
                sys.path.insert(0, p)
        _stub_tensorboard()
        import open3d
        assert os.path.abspath(open3d.__file__).startswith(PKG), open3d.__file__
        import open3d.ml as _ml3d
        import open3d.ml.torch  # noqa: F401   (registers the native model classes over the checkout's)
        if emu:
            import emu_runtime
            emu_runtime.install("ml3d_amd")
            dev = "cpu"
        else:
            import torch
            assert torch.cuda.is_available(), "--side native needs an MI355X (or --emu)"
            dev = "cuda"
        return _ml3d.utils, dev
    os.environ["ML3D_REFERENCE_ROOT"] = ref
    from oracle import ref_shim
    ref_shim.install()
    import ml3d.torch  # noqa: F401   (the checkout's: registers its own models / pipelines)
    import ml3d.utils as utils
    return utils, "cpu"


def seed_all(s):
    import torch
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def small_overrides(name, cfg):
    m = cfg.model
    if name == "randlanet":
        m["num_points"] = 1024
        m["dim_output"] = [16, 32, 64, 64]
    elif name.startswith("kpconv"):
        m["first_subsampling_dl"] = 0.3
        m["in_radius"] = 2.5
        m["min_in_points"] = 120
        m["max_in_points"] = 500
        m["batch_limit"] = 500
        m["first_features_dim"] = 32
    else:
        import synth_weights
        s = copy.deepcopy(synth_weights.POINTPILLARS_SMALL_CFG)
        s["voxel_encoder"]["in_channels"] = 4
        for k, v in s.items():
            m[k] = v


def make_data(name, small):
    import synth_data
    if name == "randlanet":
        pts = synth_data.lidar_sweep(4100)
        if small:
            pts = pts[np.linalg.norm(pts[:, :2], axis=1) < 4.2]
        return dict(point=np.ascontiguousarray(pts, np.float32), feat=None,
                    label=(1 + (np.arange(pts.shape[0]) % 19)).astype(np.int32))
    if name.startswith("kpconv"):
        return synth_data.toronto3d_tile(7, half=2.0 if small else 6.0, density=0.1 if small else 0.25)
    sweep = synth_data.kitti_sweep(11)
    return dict(point=np.ascontiguousarray(sweep, np.float32), calib=None, bounding_boxes=[])


def state_dict_for(name, model_cfg):
    import synth_weights
    cfg = {k: (v.to_dict() if hasattr(v, "to_dict") else v) for k, v in dict(model_cfg).items()}
    if name == "randlanet":
        return synth_weights.randlanet_state_dict(cfg, 31)
    if name.startswith("kpconv"):
        return synth_weights.kpconv_state_dict(cfg, 32)
    return synth_weights.pointpillars_state_dict(cfg, 33)


def run_one(name, side, utils, dev, ref, small, out_dir, sampler_index="sklearn"):
    import torch
    cfg = utils.Config.load_from_file(os.path.join(ref, "ml3d", "configs", MODELS[name]))
    if small:
        small_overrides(name, cfg)
    if side == "native" and name.startswith("kpconv") and sampler_index != "gpu":
        # the radius sampler's sphere order is sklearn's tree-traversal order (unsorted by contract); the native class reproduces
        # it only with the reference's own index structure for that ONE query per sphere (ml3d/torch/models/_datapath.py)
        cfg.model["sampler_index"] = sampler_index
    Model = utils.get_module("model", cfg.model.name, "torch")
    Pipeline = utils.get_module("pipeline", cfg.pipeline.name, "torch")
    print("[%s/%s] model class %s.%s   pipeline class %s.%s" % (name, side, Model.__module__, Model.__name__,
                                                                Pipeline.__module__, Pipeline.__name__), flush=True)
    if side == "native":
        assert Model.__module__.startswith("ml3d_amd."), "registry did not resolve to the native class"
    else:
        assert Model.__module__.startswith("ml3d.torch.models"), Model.__module__
    assert Pipeline.__module__.startswith("ml3d.torch.pipelines"), "the pipeline must be the reference's own"
    cwd = os.getcwd()
    work = tempfile.mkdtemp(prefix="ml3d_pipe_")
    os.chdir(work)           # the reference writes ./logs and (KPConv) ./kernels into the working directory
    try:
        seed_all(7)
        model = Model(**cfg.model, device=dev)
        sd = state_dict_for(name, cfg.model)
        missing = model.load_state_dict(sd, strict=True)
        model.eval()
        pipeline = Pipeline(model, dataset=None, device=dev, **cfg.pipeline)
        data = make_data(name, small)
        seed_all(11)
        t0 = time.time()
        res = pipeline.run_inference(data)
        if dev == "cuda":
            torch.cuda.synchronize()
        dt = time.time() - t0
    finally:
        os.chdir(cwd)
    out = dict(seconds=np.float64(dt), n_points=np.int64(data["point"].shape[0]))
    if name == "pointpillars":
        boxes = res[0]
        out["boxes"] = np.array([b.to_xyzwhlr() for b in boxes], np.float32).reshape(-1, 7)
        out["scores"] = np.array([b.confidence for b in boxes], np.float32)
        out["labels"] = np.array([model.name2lbl.get(b.label_class, -1) for b in boxes], np.int64)
        print("\n[%s/%s] %d boxes in %.2f s" % (name, side, len(boxes), dt), flush=True)
    else:
        out["predict_labels"] = np.asarray(res["predict_labels"]).astype(np.int64)
        out["predict_scores"] = np.asarray(res["predict_scores"]).astype(np.float32)
        print("\n[%s/%s] %d points labelled in %.2f s; label histogram %s" % (
            name, side, out["predict_labels"].shape[0], dt, np.bincount(out["predict_labels"]).tolist()), flush=True)
    os.makedirs(out_dir, exist_ok=True)
    np.savez_compressed(os.path.join(out_dir, "%s_%s%s.npz" % (name, side, "_small" if small else "")), **out)
    return out


def compare(a_dir, b_dir, small):
    ok = True
    sfx = "_small" if small else ""
    for name in MODELS:
        fa = os.path.join(a_dir, "%s_native%s.npz" % (name, sfx))
        fb = os.path.join(b_dir, "%s_reference%s.npz" % (name, sfx))
        if not (os.path.exists(fa) and os.path.exists(fb)):
            print("[compare] %s: missing (%s / %s)" % (name, os.path.exists(fa), os.path.exists(fb)))
            continue
        a, b = np.load(fa), np.load(fb)
        if name == "pointpillars":
            same_n = a["boxes"].shape == b["boxes"].shape
            print("[compare] pointpillars: boxes native %d / reference %d" % (a["boxes"].shape[0], b["boxes"].shape[0]))
            if same_n and a["boxes"].shape[0]:
                # both lists are class-major, NMS-ordered: compare row by row
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                from pipeline_loop import box_lists_agree
                # per class a one-to-one matching (candidates whose scores differ by < 1e-4 may swap places in NMS order)
                db = box_lists_agree(a["boxes"], a["labels"], b["boxes"], b["labels"])
                ds = np.abs(a["scores"] - b["scores"]).max()
                lab = bool((a["labels"] == b["labels"]).all())
                print("[compare] pointpillars: labels identical %s, boxes matched one-to-one within rel %.3g (max |box| %.3g), "
                      "max|d score| %.3g" % (lab, db, np.abs(b["boxes"]).max(), ds))
                ok &= lab and db <= 1e-4 and ds <= 1e-4
            else:
                ok &= same_n
        else:
            la, lb = a["predict_labels"], b["predict_labels"]
            agree = float((la == lb).mean()) if la.shape == lb.shape else 0.0
            ds = float(np.abs(a["predict_scores"] - b["predict_scores"]).max()) if la.shape == lb.shape else float("nan")
            print("[compare] %s: %d points, label agreement %.5f, max|d vote| %.4g  (native %.1f s, reference %.1f s)" % (
                name, la.shape[0], agree, ds, float(a["seconds"]), float(b["seconds"])))
            ok &= agree >= 0.99
    print("[compare] %s" % ("OK" if ok else "MISMATCH"))
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", choices=["native", "reference"])
    ap.add_argument("--model", default="all")
    ap.add_argument("--ref", default=os.environ.get("OPEN3D_ML_ROOT") or os.environ.get("ML3D_REFERENCE_ROOT") or "/root/reference")
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--sampler-index", default="sklearn", choices=["sklearn", "gpu"],
                    help="KPFCNN (native side): index behind the radius sampler's single-centre query")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "pipelines"))
    ap.add_argument("--compare", nargs=2, metavar=("NATIVE_DIR", "REFERENCE_DIR"))
    args = ap.parse_args()
    if args.compare:
        sys.exit(0 if compare(args.compare[0], args.compare[1], args.small) else 1)
    ref = os.path.abspath(args.ref)
    utils, dev = setup(args.side, ref, args.emu)
    names = list(MODELS) if args.model == "all" else [args.model]
    for n in names:
        run_one(n, args.side, utils, dev, ref, args.small, args.out, args.sampler_index)


if __name__ == "__main__":
    main()
