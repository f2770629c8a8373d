#!/usr/bin/env python3
"""Run the REFERENCE's own ``scripts/run_pipeline.py torch -c <yaml> --split test`` end to end, once per model family, as a
SUBPROCESS with its unchanged command line, on synthetic dataset directories (tools/synth_datasets.py) and a checkpoint file
holding seeded pseudo-trained weights (there are no datasets / checkpoints offline):

    RandLA-Net    randlanet_semantic3d.yml    Semantic3D-format directory (.txt clouds)      SemanticSegmentation.run_test
    KPConv        kpconv_semantic3d.yml       same directory                                   SemanticSegmentation.run_test
    PointPillars  pointpillars_kitti.yml      KITTI-format directory (velodyne / calib / label_2)   ObjectDetection.run_test
                                                                                               (+ run_valid: loss + mAP)

(randlanet_semantickitti.yml cannot run through the reference's own SemanticKITTI class at all: the class hands over xyz +
intensity, the YAML says in_channels 3, randlanet.py:208-211 raises -- on the reference itself.)

Sides:  --side native     PYTHONPATH = open3d-ml_amd (this repository's ``open3d`` package), OPEN3D_ML_ROOT = the checkout:
                          pipelines, datasets, samplers, dataloaders, batchers are the reference's files, the model classes the
                          MI355X-native ones, every primitive a HIP kernel.  Needs a GPU (--emu: host emulation, tiny sizes).
        --side reference  the checkout's own PyTorch-CPU models on the oracle's C ops (oracle/ref_shim.py).
What ``run_test`` leaves on disk (dataset.save_test_result: Semantic3D ``.labels`` files / KITTI result ``.txt`` files) is
compared by --compare.  Nothing here is product code; tools/e2e_site/sitecustomize.py prepares the interpreter.
"""
import argparse
import glob
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

FAMILIES = {"randlanet": ("randlanet_semantic3d.yml", "semantic3d"), "kpconv": ("kpconv_semantic3d.yml", "semantic3d"),
            "pointpillars": ("pointpillars_kitti.yml", "kitti")}


def plain(x):
    if isinstance(x, dict):
        return {k: plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [plain(v) for v in x]
    return x


def make_checkpoint(family, cfg_path, path):
    import torch
    import yaml
    import synth_weights
    m = plain(yaml.safe_load(open(cfg_path))["model"])
    sd = {"randlanet": synth_weights.randlanet_state_dict, "kpconv": synth_weights.kpconv_state_dict,
          "pointpillars": synth_weights.pointpillars_state_dict}[family](m, 31)
    torch.save({"model_state_dict": sd}, path)


def small_yaml(family, src, dst):
    """--small (emulator only): the YAML with shrunken sizes -- debugging the host glue, NOT the run that is reported."""
    import yaml
    c = yaml.safe_load(open(src))
    m = c["model"]
    if family == "randlanet":
        m.update(num_points=8192, dim_output=[16, 32, 32, 64, 64])
        c["dataset"]["num_points"] = 8192
    elif family == "kpconv":
        m.update(first_subsampling_dl=0.3, in_radius=2.5, min_in_points=120, max_in_points=500, batch_limit=500, first_features_dim=32)
    else:
        import synth_weights
        import copy
        s = copy.deepcopy(synth_weights.POINTPILLARS_SMALL_CFG)
        s["voxel_encoder"]["in_channels"] = 4
        m.update(s)
        c["pipeline"]["overlaps"] = [0.7, 0.5]           # (two classes: Car, Pedestrian)
    yaml.safe_dump(c, open(dst, "w"))


def run(args):
    import synth_datasets
    ref = os.path.abspath(args.ref)
    work = os.path.abspath(args.work)
    out = os.path.abspath(args.out)
    os.makedirs(work, exist_ok=True)
    os.makedirs(out, exist_ok=True)
    ds = {"semantic3d": os.path.join(work, "Semantic3D"), "kitti": os.path.join(work, "KITTI")}
    fams = list(FAMILIES) if args.family == "all" else [args.family]
    train = args.split == "train"
    if train:
        fams = [f for f in fams if f != "pointpillars"]     # (pointpillars_kitti.yml's ObjectSample augmentation needs a ground-truth database)
    if any(FAMILIES[f][1] == "semantic3d" for f in fams) and (not glob.glob(os.path.join(ds["semantic3d"], "*.txt")) or
                                                              (train and not glob.glob(os.path.join(ds["semantic3d"], "*.labels")))):
        # --split train: four labelled clouds, two of them under the names the YAMLs list in `val_files` (semantic3d.py:92-99)
        synth_datasets.write_semantic3d(ds["semantic3d"], half=2.5 if args.small else 9.0, density=0.1 if args.small else 0.35,
                                        n_train=4 if train else 0,
                                        val_names=("bildstein_station1_xyz_intensity_rgb", "domfountain_station1_xyz_intensity_rgb") if train else ())
    if "pointpillars" in fams and not os.path.isdir(ds["kitti"]):
        synth_datasets.write_kitti(ds["kitti"], n_test=1 if args.small else 2, n_train=1 if args.small else 2)
    rc_all = 0
    for fam in fams:
        yml, kind = FAMILIES[fam]
        cfg = os.path.join(ref, "ml3d", "configs", yml)
        if args.small:
            small = os.path.join(work, "small_" + yml)
            small_yaml(fam, cfg, small)
            cfg = small
        ckpt = os.path.join(work, fam + "_ckpt.pth")
        make_checkpoint(fam, cfg, ckpt)
        res_dir = os.path.join(out, "%s_%s" % (fam, args.side))
        shutil.rmtree(res_dir, ignore_errors=True)
        run_dir = os.path.join(work, "run_%s_%s" % (fam, args.side))
        shutil.rmtree(run_dir, ignore_errors=True)
        os.makedirs(run_dir)
        dev = "cuda" if (args.side == "native" and not args.emu) else "cpu"
        argv = ["torch", "-c", cfg, "--dataset_path", ds[kind], "--ckpt_path", ckpt, "--split", args.split, "--device", dev,
                "--main_log_dir", os.path.join(run_dir, "logs"),
                "--dataset.use_cache", "False", "--dataset.test_result_folder", res_dir,
                "--dataset.cache_dir", os.path.join(run_dir, "cache")]
        if fam == "kpconv" and args.side == "native" and args.sampler_index != "gpu":
            # the radius sampler cuts sklearn's UNSORTED query_radius list by position (kpconv.py:433-437 on the reference): the
            # native class reproduces that order only with the reference's own index structure for that one query per sphere
            # (INTEGRATION.md §2.1); with the GPU index the spheres hold the same SETS cut differently -- equivalent, not identical
            argv += ["--model.sampler_index", args.sampler_index]
        if fam == "pointpillars":
            # every training sweep is a validation sweep (kitti.py:64-69), loaders in-process
            argv += ["--dataset.val_split", "0", "--pipeline.num_workers", "0", "--pipeline.pin_memory", "False"]
        if train:
            # ONE epoch (range(0, max_epoch + 1), semantic_segmentation.py:407) of a few optimisation steps + a validation pass,
            # loaders in-process (the transforms call the GPU ops), a checkpoint at the end
            argv += ["--pipeline.max_epoch", "0", "--pipeline.batch_size", "2", "--pipeline.val_batch_size", "2",
                     "--pipeline.num_workers", "0", "--pipeline.pin_memory", "False", "--pipeline.save_ckpt_freq", "1",
                     "--dataset.steps_per_epoch_train", str(args.train_steps * 2), "--dataset.steps_per_epoch_valid", "2"]
        env = dict(os.environ, ML3D_E2E_SIDE=args.side, ML3D_E2E_SEED="7", PYTHONHASHSEED="0", PYTHONWARNINGS="ignore", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "16"))
        pp = [os.path.join(ROOT, "tools", "e2e_site")]
        if args.side == "native":
            pp.append(os.path.join(ROOT, "open3d-ml_amd"))
            env["OPEN3D_ML_ROOT"] = ref
            if args.emu:
                env["ML3D_E2E_EMU"] = "1"
        else:
            env["ML3D_REFERENCE_ROOT"] = ref
        env["PYTHONPATH"] = os.pathsep.join(pp + [env.get("PYTHONPATH", "")]).rstrip(os.pathsep)
        cmd = [sys.executable, os.path.join(ref, "scripts", "run_pipeline.py")] + argv
        print("== [%s/%s] %s" % (fam, args.side, " ".join(["python", "scripts/run_pipeline.py"] + [a.replace(work, "$WORK").replace(ref, "$REF").replace(out, "$OUT") for a in argv])), flush=True)
        t0 = time.time()
        p = subprocess.run(cmd, cwd=run_dir, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, stdin=subprocess.DEVNULL)
        dt = time.time() - t0
        lines = [ln for ln in p.stdout.replace("\r", "\n").splitlines() if ln.strip() and not ln.rstrip().endswith("it/s]")
                 and "it/s" not in ln[-24:] and "s/it" not in ln[-24:]]
        keep = [ln for ln in lines if any(k in ln for k in ("INFO", "Error", "error", "Traceback", "mAP", "loss", "Saved", "Overall", "File \""))]
        for ln in (keep[-40:] if p.returncode == 0 else lines[-60:]):
            print("   | " + ln[:220])
        os.makedirs(res_dir, exist_ok=True)
        if train:
            # what a training run leaves behind: its epoch summary (save_logs, semantic_segmentation.py:520-560) and the checkpoint
            with open(os.path.join(res_dir, "train.log"), "w") as f:
                f.write("\n".join([ln[ln.index("e2e-step-loss"):].strip() for ln in p.stdout.replace("\r", "\n").splitlines() if "e2e-step-loss" in ln] +
                                  [ln.split(" - ", 2)[-1] for ln in lines if " semantic_segmentation - " in ln and
                                   any(k in ln for k in ("Loss train", "Mean acc", "Mean IoU", "EPOCH"))]) + "\n")
            for ck in glob.glob(os.path.join(run_dir, "logs", "**", "ckpt_*.pth"), recursive=True):
                shutil.copy(ck, os.path.join(res_dir, os.path.basename(ck)))
        if fam == "pointpillars":
            # (the reference's ObjectDetectBatch never fills ``attr`` (concat_batcher.py:503-519), so ObjectDetection.run_test's
            #  save_test_result(results, data.attr) writes nothing on either side: what a detection run leaves behind is its log)
            with open(os.path.join(res_dir, "validation.log"), "w") as f:
                f.write("\n".join(ln.split(" - ", 2)[-1] for ln in lines if " object_detection - " in ln and
                                  any(k in ln for k in ("validation -", "mAP", "difficulty", ":  ", "Overall"))) + "\n")
        written = sorted(glob.glob(os.path.join(res_dir, "**", "*.*"), recursive=True))
        print("== [%s/%s] exit %d in %.1f s; result files: %s" % (fam, args.side, p.returncode, dt,
                                                                 [os.path.relpath(w, res_dir) for w in written]), flush=True)
        # (the reference's run_test ends with `self.metric_test.acc()[-1]` -- None when the test split has no labels, as
        #  Semantic3D's has not: semantic_segmentation.py:265-266 raises TypeError AFTER every result has been saved, on the
        #  reference's own models just the same; that exit is the reference's behaviour, not a failure of the side under test)
        ref_quirk = p.returncode != 0 and "metric_test.acc()[-1]" in p.stdout and "'NoneType' object is not subscriptable" in p.stdout
        if ref_quirk:
            print("   (exit 1 = the reference's own final log line on an unlabeled test split, after all results were saved)")
        rc_all |= (p.returncode != 0 and not ref_quirk) or not written
    return rc_all


def _read_kitti_result(path):
    rows = []
    for ln in open(path):
        f = ln.split()
        if f:
            rows.append((f[0], np.array([float(v) for v in f[8:16]])))      # class, h w l x y z ry score
    return rows


def compare(a_dir, b_dir, later_tol=2e-2, summary_tol=3e-2):
    ok = True
    for fam in FAMILIES:
        fa = sorted(glob.glob(os.path.join(a_dir, fam + "_native", "**", "*.*"), recursive=True))
        fb = sorted(glob.glob(os.path.join(b_dir, fam + "_reference", "**", "*.*"), recursive=True))
        if not fa or not fb:
            print("[compare] %s: missing (native %d files, reference %d files)" % (fam, len(fa), len(fb)))
            continue
        if [os.path.basename(f) for f in fa] != [os.path.basename(f) for f in fb]:
            print("[compare] %s: different result files %s / %s" % (fam, fa, fb))
            ok = False
            continue
        for x, y in zip(fa, fb):
            if x.endswith("train.log"):
                import re
                ta, tb = open(x).read(), open(y).read()
                sa = [float(l.split()[-1]) for l in ta.splitlines() if l.startswith("e2e-step-loss")]
                sb = [float(l.split()[-1]) for l in tb.splitlines() if l.startswith("e2e-step-loss")]
                ea = [float(v) for l in ta.splitlines() if not l.startswith("e2e") for v in re.findall(r"[-+]?\d+\.\d+", l)]
                eb = [float(v) for l in tb.splitlines() if not l.startswith("e2e") for v in re.findall(r"[-+]?\d+\.\d+", l)]
                same = len(sa) == len(sb) and len(sa) > 0 and len(ea) == len(eb)
                d0 = abs(sa[0] - sb[0]) if same else float("nan")
                dmax = max(abs(p - q) / max(1.0, abs(q)) for p, q in zip(sa, sb)) if same else float("nan")
                de = max([abs(p - q) for p, q in zip(ea, eb)] or [float("nan")]) if same else float("nan")
                print("[compare] %s train.log: %d step losses -- first step |d| %.2g (native %.6f, reference %.6f), worst later step "
                      "%.2g relative (after Adam updates); epoch summary (loss / acc / IoU, 3 decimals) max |d| %.3g" % (
                          fam, len(sa), d0, sa[0] if sa else 0, sb[0] if sb else 0, dmax, de))
                print("\n".join("      native    | " + l for l in ta.splitlines() if not l.startswith("e2e")))
                print("\n".join("      reference | " + l for l in tb.splitlines() if not l.startswith("e2e")))
                ok &= same and d0 <= 2e-4 and dmax <= later_tol and de <= summary_tol
            elif x.endswith(".pth"):
                import torch
                sa, sb = torch.load(x, map_location="cpu")["model_state_dict"], torch.load(y, map_location="cpu")["model_state_dict"]
                keys = [k for k in sb if sb[k].dtype.is_floating_point]
                rel = sorted(float((sa[k].float() - sb[k].float()).abs().max()) / max(1e-3, float(sb[k].float().abs().max())) for k in keys)
                print("[compare] %s %s: %d tensors after the optimisation steps, relative difference median %.2g, worst %.2g" % (
                    fam, os.path.basename(x), len(sb), rel[len(rel) // 2], rel[-1]))
                ok &= set(sa) == set(sb)
            elif x.endswith(".log"):
                ta, tb = open(x).read(), open(y).read()
                print("[compare] %s %s: %d lines, identical %s\n%s" % (fam, os.path.basename(x), ta.count("\n"), ta == tb,
                                                                       "\n".join("      " + l for l in ta.splitlines()[:3])))
                ok &= ta == tb and "loss_cls" in ta
            elif x.endswith(".labels"):
                la, lb = np.loadtxt(x, dtype=np.int64), np.loadtxt(y, dtype=np.int64)
                agree = float((la == lb).mean()) if la.shape == lb.shape else 0.0
                print("[compare] %s %s: %d points, label agreement %.5f (identical bytes: %s)" % (
                    fam, os.path.basename(x), la.size, agree, open(x, "rb").read() == open(y, "rb").read()))
                ok &= agree >= 0.999
            else:
                ra, rb = _read_kitti_result(x), _read_kitti_result(y)
                same = len(ra) == len(rb) and all(p[0] == q[0] for p, q in zip(ra, rb))
                d = max([float(np.abs(p[1] - q[1]).max()) for p, q in zip(ra, rb)] or [0.0]) if same else float("nan")
                print("[compare] %s %s: %d / %d boxes, classes in the same order %s, max |d field| %.3g (2-decimal text)" % (
                    fam, os.path.basename(x), len(ra), len(rb), same, d))
                ok &= same and d <= 0.011
    print("[compare] %s" % ("OK" if ok else "MISMATCH"))
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", choices=["native", "reference"])
    ap.add_argument("--family", default="all", choices=list(FAMILIES) + ["all"])
    ap.add_argument("--ref", default=os.environ.get("OPEN3D_ML_ROOT") or os.environ.get("ML3D_REFERENCE_ROOT") or "/root/reference")
    ap.add_argument("--work", default="/tmp/ml3d_e2e")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "e2e"))
    ap.add_argument("--small", action="store_true")
    ap.add_argument("--split", default="test", choices=["test", "train"])
    ap.add_argument("--train-steps", type=int, default=3, help="--split train: optimisation steps (batches of 2) in the one epoch")
    ap.add_argument("--sampler-index", default="sklearn", choices=["sklearn", "gpu"])
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--compare", nargs=2, metavar=("NATIVE_OUT", "REFERENCE_OUT"))
    ap.add_argument("--later-tol", type=float, default=2e-2, help="--compare: relative bound on the step losses after the first step")
    ap.add_argument("--summary-tol", type=float, default=3e-2, help="--compare: bound on the epoch summary numbers")
    args = ap.parse_args()
    if args.compare:
        sys.exit(0 if compare(*args.compare, later_tol=args.later_tol, summary_tol=args.summary_tol) else 1)
    sys.exit(run(args))


if __name__ == "__main__":
    main()
