"""Per-primitive measurement on one MI355X (SURVEY.md §8d component table): GPU time (HIP events on torch's
current stream, where the ops are launched), achieved GB/s on the ALGORITHMIC bytes of the op, fraction of the
8 TB/s HBM peak, and the CPU oracle timed beside it on the same input.
usage: python tools/bench_ops.py [iters]  -> one JSON line per op"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import synth_data
import synth_weights as W
from ml3d import ops
from oracle import ops as oops

PEAK = 8000.0


def gpu_ms(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def cpu_ms(fn, budget=3.0):
    fn()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget and n < 20:
        fn()
        n += 1
    return (time.perf_counter() - t0) / n * 1e3


def report(name, what, nbytes, g_ms, c_ms):
    gbs = nbytes / (g_ms * 1e-3) / 1e9
    print(json.dumps({"op": name, "workload": what, "algorithmic_bytes": int(nbytes), "gpu_ms": round(g_ms, 4),
                      "achieved_GBps": round(gbs, 1), "hbm_frac": round(gbs / PEAK, 4), "cpu_oracle_ms": None if c_ms is None else round(c_ms, 2),
                      "cpu_threads": oops.num_threads(),
                      "speedup_vs_cpu": None if c_ms is None else round(c_ms / g_ms, 1)}), flush=True)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    # ---- exact 16-NN, one 45 056-point SemanticKITTI-shaped patch x 16 (batched row_splits) ----------------
    B, N = 16, 45056
    frames = np.concatenate([synth_data.semantickitti_patch(i, N) for i in range(B)])
    rs = np.arange(B + 1) * N
    tp, trs = t(frames), t(rs)
    g = gpu_ms(lambda: ops.knn_search(tp, tp, 16, trs, trs), iters)
    c = cpu_ms(lambda: oops.knn_search_batched(frames, rs, frames, rs, 16))
    report("knn_search(k=16)", "%d x %d points, self query" % (B, N), B * N * (12 + 64), g, c)

    # ---- fixed-radius search + dense (KPConv layer 0: r = 0.2) on 8 spheres of 10 000 points -----------------
    spheres = [synth_data.toronto3d_sphere(200 + i) for i in range(8)]
    sp = np.concatenate(spheres)
    lens = [len(s) for s in spheres]
    tsp = t(sp)
    dense = ops.radius_neighbors_dense(tsp, tsp, lens, lens, 0.2)
    g = gpu_ms(lambda: ops.radius_neighbors_dense(tsp, tsp, lens, lens, 0.2), iters)
    prs = np.concatenate([[0], np.cumsum(lens)])
    c = cpu_ms(lambda: oops.fixed_radius_search(sp, sp, 0.2, prs, prs))
    report("radius_neighbors_dense(r=0.2)", "8 spheres, %d points, dense [%d, %d]" % (len(sp), dense.shape[0], dense.shape[1]),
           len(sp) * 24 + dense.numel() * 4, g, c)

    # ---- grid subsample 0.06 m of a raw sweep -----------------------------------------------------------------
    sweep = synth_data.lidar_sweep(5)
    tsw = t(sweep)
    out = ops.subsample(tsw, sampleDl=0.06)
    g = gpu_ms(lambda: ops.subsample(tsw, sampleDl=0.06), iters)
    c = cpu_ms(lambda: oops.subsample(sweep, sampleDl=0.06))
    report("subsample(0.06)", "%d -> %d points" % (len(sweep), out.shape[0]), len(sweep) * 12 + out.shape[0] * 12, g, c)

    # ---- voxelize (PointPillars KITTI) ---------------------------------------------------------------------------
    cfg = W.POINTPILLARS_KITTI_CFG
    cloud = W.crop_for_cfg(synth_data.kitti_sweep(0), cfg)
    tc = t(cloud)
    vz, pcr = cfg["voxelize"], cfg["point_cloud_range"]
    args = (torch.tensor([0, len(cloud)]), torch.tensor(vz["voxel_size"]), torch.tensor(pcr[:3]), torch.tensor(pcr[3:]),
            vz["max_num_points"], vz["max_voxels"][1])
    v = ops.voxelize(tc[:, :3], *args)
    M, K = v.voxel_coords.shape[0], v.voxel_point_indices.shape[0]
    g = gpu_ms(lambda: ops.voxelize(tc[:, :3], *args), iters)
    c = cpu_ms(lambda: oops.voxelize(cloud[:, :3], [0, len(cloud)], vz["voxel_size"], pcr[:3], pcr[3:], vz["max_num_points"],
                                     vz["max_voxels"][1]))
    report("voxelize", "%d points -> %d pillars" % (len(cloud), M), len(cloud) * 12 + M * 12 + K * 8 + (M + 1) * 8, g, c)

    # ---- fused pillar gather + PFN + scatter (canvas zero + write) -------------------------------------------------
    from ml3d.torch.models.point_pillars import PointPillars
    m = PointPillars(device=dev, **cfg)
    m.load_state_dict(W.pointpillars_state_dict(cfg, 1))
    P = m.packed_params(dev)
    ve, vl = m.voxel_encoder, m.voxel_layer
    ny, nx = m.middle_encoder.ny, m.middle_encoder.nx
    rs1 = torch.tensor([0, len(cloud)], dtype=torch.int64, device=dev)
    vox = ops.voxelize(tc[:, :3], rs1, *args[1:])
    f = lambda: ops.pillar_features(tc, vox, ve.raw_channels, vl.max_num_points, ve.vx, ve.vy, ve.x_offset, ve.y_offset,
                                    nx, ny, P['pfn'], 1)
    g = gpu_ms(f, iters)
    report("pillar_features (gather+PFN+scatter)", "%d pillars -> canvas %dx%dx64" % (M, ny, nx),
           len(cloud) * 16 + M * 256 + ny * nx * 64 * 4, g, None)

    # ---- rotated NMS, nms_pre = 4096 --------------------------------------------------------------------------------
    rng = np.random.default_rng(0)
    n = 4096
    cxy = rng.random((n, 2), dtype=np.float32) * 90
    wh = 0.5 + rng.random((n, 2), dtype=np.float32) * 3
    bx = np.concatenate([cxy - wh / 2, cxy + wh / 2, (rng.random((n, 1), dtype=np.float32) * 2 - 1) * np.pi], 1).astype(np.float32)
    sc = rng.random(n, dtype=np.float32)
    tb, ts = t(bx), t(sc)
    g = gpu_ms(lambda: ops.nms(tb, ts, 0.3), iters)
    c = cpu_ms(lambda: oops.nms(bx, sc, 0.3))
    report("nms(4096 boxes)", "latency-bound; bytes = boxes + bit mask", n * 24 + n * n // 8, g, c)

    # ---- raw-sweep front end (row f3): subsample 0.06 + features/labels + raw->sub 1-NN projection ----------------
    from ml3d.datasets import preprocess_sweep
    rng = np.random.default_rng(9)
    data = {"point": sweep[:, :3], "feat": rng.random((len(sweep), 1), dtype=np.float32),
            "label": rng.integers(0, 20, len(sweep)).astype(np.int32)}
    o = preprocess_sweep(data, 0.06, "test")
    g = gpu_ms(lambda: preprocess_sweep(data, 0.06, "test"), max(3, iters // 4))

    def cpu_front():
        sp_, sf_, sl_ = oops.subsample(data["point"], features=data["feat"], classes=data["label"], sampleDl=0.06)
        return oops.knn_search(sp_, data["point"], 1)
    c = cpu_ms(cpu_front)
    report("preprocess_sweep (f3, incl. H2D/D2H of the sweep)", "%d -> %d points + proj_inds" % (len(sweep), o["point"].shape[0]),
           len(sweep) * (12 + 4 + 4 + 4) + o["point"].shape[0] * 20, g, c)

    # ---- patch sampler query (row f1): the 45 056 nearest points to a centre -----------------------------------------
    subp = t(o["point"])
    ctr = torch.tensor(o["point"][len(o["point"]) // 2])
    if subp.shape[0] >= 45056:
        g = gpu_ms(lambda: ops.nearest_to_center(subp, ctr, 45056), iters)
        c = cpu_ms(lambda: oops.knn_search(o["point"], o["point"][len(o["point"]) // 2][None], 45056))
        report("nearest_to_center(k=45056) (f1)", "%d sub-sampled points" % subp.shape[0], subp.shape[0] * 12 + 45056 * 4, g, c)


if __name__ == "__main__":
    main()
