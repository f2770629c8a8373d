"""Generate tests/golden/pipeline_{randlanet,kpconv,pointpillars,kpconv_deform}.npz from the REAL reference pipelines, in this container:

    python -m oracle.gen_golden_pipeline            (needs /root/reference; ~1 min of CPU)

For each of the four unchanged YAML configs (randlanet_semantickitti / kpconv_toronto3d / pointpillars_kitti / kpconv_parislille3d) the
reference's OWN pipeline class runs ``run_inference`` on a seeded synthetic cloud with the reference's OWN PyTorch-CPU model
(the oracle's C ops stand in for the un-installable open3d wheel, oracle/ref_shim.py) — tools/ref_pipelines.py, side
"reference".  The same models then go through tests/pipeline_loop.py (the checkout-free restatement of that loop which the
-m gpu tests use) and the two results must be IDENTICAL: that pins the restatement.  Stored: the model section of the YAML
(json), seeds, predicted labels of every raw point (uint8), the float16 votes of every 8th point, detection boxes.
The GPU box has no /root/reference: tests only read the .npz files written here.
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_pipelines as RP   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
STRIDE = 8


def _plain(x):
    if hasattr(x, "to_dict"):
        x = x.to_dict()
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def main():
    import torch
    import pipeline_loop as PL
    ref = os.path.abspath(os.environ.get("ML3D_REFERENCE_ROOT", "/root/reference"))
    utils, dev = RP.setup("reference", ref, False)
    tmp = tempfile.mkdtemp(prefix="ml3d_golden_pipe_")
    for name in RP.MODELS:
        real = RP.run_one(name, "reference", utils, dev, ref, False, tmp)           # the checkout's pipeline class
        cfg = utils.Config.load_from_file(os.path.join(ref, "ml3d", "configs", RP.MODELS[name]))
        Model = utils.get_module("model", cfg.model.name, "torch")
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            RP.seed_all(7)
            model = Model(**cfg.model, device="cpu")
            model.load_state_dict(RP.state_dict_for(name, cfg.model), strict=True)
            model.eval()
            model.device = torch.device("cpu")
            data = RP.make_data(name, False)
            RP.seed_all(11)
            if name == "pointpillars":
                from ml3d.torch.dataloaders import ConcatBatcher
                boxes = PL.run_detection(model, data, "cpu", ConcatBatcher("cpu", model.cfg.name))[0]
                mine = dict(boxes=np.array([b.to_xyzwhlr() for b in boxes], np.float32).reshape(-1, 7),
                            scores=np.array([b.confidence for b in boxes], np.float32),
                            labels=np.array([model.name2lbl.get(b.label_class, -1) for b in boxes], np.int64))
                for k in ("boxes", "scores", "labels"):
                    assert np.array_equal(mine[k], real[k]), "pipeline_loop.run_detection != ObjectDetection.run_inference (%s)" % k
                out = dict(mine)
            else:
                from ml3d.torch.dataloaders import ConcatBatcher, DefaultBatcher        # the CHECKOUT's batchers
                collate = DefaultBatcher().collate_fn if name == "randlanet" else \
                    ConcatBatcher("cpu", model.cfg.name).collate_fn
                res = PL.run_segmentation(model, data, int(cfg.pipeline.batch_size), collate)
                assert np.array_equal(res["predict_labels"], real["predict_labels"]), \
                    "pipeline_loop.run_segmentation != SemanticSegmentation.run_inference (labels, %s)" % name
                assert np.array_equal(res["predict_scores"].astype(np.float32), real["predict_scores"]), \
                    "pipeline_loop.run_segmentation != SemanticSegmentation.run_inference (votes, %s)" % name
                out = dict(predict_labels=real["predict_labels"].astype(np.uint8),
                           predict_scores_strided=real["predict_scores"][::STRIDE].astype(np.float16),
                           stride=np.int64(STRIDE), steps=np.int64(res["steps"]), batch_size=np.int64(cfg.pipeline.batch_size))
        finally:
            os.chdir(cwd)
        out["model_cfg_json"] = np.array(json.dumps(_plain(cfg.model)))
        out["n_raw_points"] = np.int64(data["point"].shape[0])
        path = os.path.join(OUT, "pipeline_%s.npz" % name)
        np.savez_compressed(path, **out)
        print("wrote %s (%d bytes): restated loop == real pipeline" % (path, os.path.getsize(path)), flush=True)


if __name__ == "__main__":
    main()
