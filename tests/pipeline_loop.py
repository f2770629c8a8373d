"""TEST INFRASTRUCTURE: the reference's whole-cloud inference loop, restated so that the -m gpu tests can drive the native
model classes through it WITHOUT an Open3D-ML checkout (the GPU box has none).

What it restates (and what pins it): ``SemanticSegmentation.run_inference`` + ``update_tests``
(ml3d/torch/pipelines/semantic_segmentation.py:122-187, 271-313), ``InferenceDummySplit``
(ml3d/datasets/inference_dummy.py), ``SemSegSpatiallyRegularSampler`` (ml3d/datasets/samplers/semseg_spatially_regular.py),
``TorchDataloader.__getitem__`` (ml3d/torch/dataloaders/torch_dataloader.py:72-89) and torch's ``DataLoader`` batching with
num_workers = 0; for detection ``ObjectDetection.run_inference`` (object_detection.py:46-75).  The restatement is pinned by
tests/golden/pipeline_*.npz: ``oracle/gen_golden_pipeline.py`` runs the REAL pipeline classes of the checkout on the
reference's CPU models, runs THIS loop on the same models, asserts identical labels / votes, and stores the result.
The random draws (python ``random``, ``numpy.random``) happen in the reference's order.
"""
import random

import numpy as np
import torch


class RegularSampler:
    """SemSegSpatiallyRegularSampler for ONE cloud in the 'test' split."""

    def __init__(self, n_points):
        self.possibility = np.random.rand(n_points) * 1e-3           # initialize_with_dataloader: np.random.rand(N) * 1e-3
        self.min_possibility = float(np.min(self.possibility))

    def cloud_indices(self):
        """gen_test: the cloud is handed out until its least visited point has possibility > 0.5"""
        while self.min_possibility <= 0.5:
            yield 0

    def point_sampler(self, pc=None, num_points=None, radius=None, search_tree=None, **unused):
        n = 0
        while n < 2:
            center_id = np.argmin(self.possibility)
            center = pc[center_id, :].reshape(1, -1)
            if radius is not None:
                idxs = search_tree.query_radius(center, r=radius)[0]
            elif pc.shape[0] < num_points:
                base = np.array(range(pc.shape[0]))
                idxs = np.asarray(list(base) + list(random.choices(base, k=num_points - pc.shape[0])))
            else:
                idxs = search_tree.query(center, k=num_points)[1][0]
            n = len(idxs)
            if n < 2:
                self.possibility[center_id] += 0.001
        random.shuffle(idxs)
        pc = pc[idxs]
        dists = np.sum(np.square((pc - center).astype(np.float32)), axis=1)
        self.possibility[idxs] += np.square(1 - dists / np.max(dists))
        self.min_possibility = float(np.min(self.possibility))
        return pc, idxs, center


def _batches(gen, size):
    """torch's BatchSampler over a generator: ``size`` indices are DRAWN before any item of the batch is fetched."""
    batch = []
    for i in gen:
        batch.append(i)
        if len(batch) == size:
            yield batch
            batch = []
    if batch:
        yield batch


def run_segmentation(model, data, batch_size, collate, num_classes=None, on_batch=None):
    """-> dict(predict_labels, predict_scores, steps): ``run_inference`` of the segmentation pipeline."""
    attr = {'idx': 0, 'name': 'inference', 'path': 'inference_data', 'split': 'test'}
    model.eval()
    processed = model.preprocess(data, {'split': 'test'})
    sampler = RegularSampler(processed['point'].shape[0])
    model.trans_point_sampler = sampler.point_sampler
    n_cls = model.cfg.num_classes if num_classes is None else num_classes
    test_probs = np.zeros((processed['point'].shape[0], n_cls), dtype=np.float16)
    steps = 0
    result = None
    with torch.no_grad():
        for ids in _batches(sampler.cloud_indices(), batch_size):
            items = [{'data': model.transform(processed, attr), 'attr': attr} for _ in ids]
            inputs = collate(items)
            results = model(inputs['data'])
            test_probs = model.update_probs(inputs, results, test_probs)
            steps += 1
            if on_batch is not None:
                on_batch(inputs, results)
            if (sampler.possibility > 0.5).all():
                proj = model.preprocess(data, {'split': 'test'}).get('proj_inds', None)
                if proj is None:
                    proj = np.arange(test_probs.shape[0])
                result = dict(predict_labels=np.argmax(test_probs[proj], 1), predict_scores=test_probs[proj], steps=steps)
    return result


def run_detection(model, data, device, batcher):
    """``ObjectDetection.run_inference`` on one raw dict: ``ConcatBatcher.collate_fn``, ``.to(device)``, forward,
    ``inference_end`` -> list of box lists (object_detection.py:58-75)."""
    model.eval()
    batch = batcher.collate_fn([{'data': data, 'attr': {'split': 'test'}}])
    batch.to(device)
    with torch.no_grad():
        results = model(batch)
        return model.inference_end(results, batch)


def box_lists_agree(boxes_a, labels_a, boxes_b, labels_b, rel=1e-4):
    """Two detection lists describe the same boxes: per class a one-to-one matching with every coordinate within ``rel`` *
    max(1, |value|).  (Row-by-row comparison is too strict: the lists are in NMS = descending-score order, and two candidates
    whose scores differ by less than the 1e-4 float tolerance of the head maps may swap places.)  -> worst relative error."""
    import numpy as np
    boxes_a, boxes_b = np.asarray(boxes_a, np.float64).reshape(-1, 7), np.asarray(boxes_b, np.float64).reshape(-1, 7)
    labels_a, labels_b = np.asarray(labels_a).reshape(-1), np.asarray(labels_b).reshape(-1)
    if boxes_a.shape != boxes_b.shape or not np.array_equal(np.sort(labels_a), np.sort(labels_b)):
        return float("inf")
    worst = 0.0
    for c in np.unique(labels_b):
        A, B = boxes_a[labels_a == c], boxes_b[labels_b == c]
        free = np.ones(len(B), bool)
        for row in A:
            err = (np.abs(B - row) / np.maximum(1.0, np.abs(B))).max(1)
            err[~free] = np.inf
            j = int(np.argmin(err))
            if not np.isfinite(err[j]) or err[j] > rel:
                return float(err[j])
            free[j] = False
            worst = max(worst, float(err[j]))
    return worst
