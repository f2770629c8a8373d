#!/usr/bin/env python
"""bench.py — RandLA-Net SemanticKITTI inference frames/s on MI355X (BASELINE.json configs[1]).

A step = one pass of the hot path over one batch of synthetic SemanticKITTI-shaped frames
already resident in HBM: GPU neighbour pyramid (4x 16-NN + 4x 1-NN, replaces the CPU
knn_search calls of RandLANet.transform) + fused RandLA-Net forward -> logits [B, 45056, 19].
N > 1: one process per GPU (torchrun), frames sharded across ranks (weak scaling), and the
only collective is an RCCL gather of the predicted labels to rank 0 inside the timed region.

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel, timed live with
HIP events recorded by the library around that kernel's launch, on the launch stream) and
`cpu_baseline` (the CPU oracle = port of the reference path, timed on this box's host cores on
a bounded sample of the same frames).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

import synth_weights

CFG = dict(synth_weights.RANDLANET_SEMANTICKITTI_CFG)  # randlanet_semantickitti.yml:17-33

PEAK_F32_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 vector == f32-input MFMA peak
PEAK_HBM_GBS = 8000.0

# the kernel the per-round rocprof summary (profiles/) shows as dominant:
#   forward tag 8*layer + {1: lfa_stage<D,1>, 2: lfa_stage<D,2>}
DOMINANT_FWD_TAG = int(os.environ.get("ML3D_BENCH_TRACE_TAG", 8 * 1 + 2))


def lfa_flops(cfg, layer, stage, n_points):
    """Algorithmic flops of what ONE attention-kernel launch computes (2 x MACs of the reference's matmuls,
    SURVEY.md §8d), per point with K neighbours: stage 1 = lse1 MLP (10 -> h) + score Linear (d x d) +
    softmax-weighted sum (d); stage 2 = lse2 MLP (h x h) + score Linear + weighted sum.  The re-computation of
    lse1 inside the stage-2 kernel is NOT counted (the reference computes it once); the pool / mlp2 / shortcut
    Linears run in separate GEMM launches and are not part of this kernel."""
    d = cfg["dim_output"][layer]
    h = d // 2
    K = cfg["num_neighbors"]
    mac = K * (10 * h if stage == 1 else h * h) + K * d * d + K * d
    return 2.0 * mac * n_points


def lfa_flops_executed(cfg, layer, stage, n_points):
    """Flops the attention kernel EXECUTES for the same result: the feature half of the score Linear is linear in a
    per-point quantity, W . [f[nb] ; r] = (W_top . f)[nb] + W_bot . r, so the kernel gathers a precomputed per-point
    row and multiplies only the position half (K x h x d); stage 2 re-computes lse1 (K x 10 x h).  The per-point GEMM
    (h x d per point, 1/16 of what it replaces) runs in its own launch and is not part of this kernel's time."""
    d = cfg["dim_output"][layer]
    h = d // 2
    K = cfg["num_neighbors"]
    mac = K * 10 * h + (K * h * h if stage == 2 else 0) + K * h * d + K * d
    return 2.0 * mac * n_points


def knn_bytes(cfg, n0):
    """Algorithmic HBM bytes of the neighbour pyramid per frame (SURVEY.md §8d row a1, int32 indices)."""
    n, tot = n0, 0
    for r in cfg["sub_sampling_ratio"]:
        tot += 12 * n + 4 * cfg["num_neighbors"] * n          # self k-NN: xyz read + idx write
        tot += 12 * (n + n // r) + 4 * n                      # 1-NN interp: xyz of both levels + idx write
        n //= r
    return tot


def cpu_baseline(frames, sd, budget_s=20.0, max_frames=6, gpu_labels=None):
    """The CPU oracle (port of the reference path: kd-tree knn_search + PyTorch-CPU forward), timed
    on a bounded sample of the same frames."""
    from oracle import ops as oops
    from oracle import randlanet_ref as R
    done, t_total = 0, 0.0
    R.forward(sd, CFG, R.build_inputs(frames[:1, :2048].copy(), frames[:1, :2048].copy(), CFG, oops.knn_search))
    cpu_labels = []
    for i in range(min(max_frames, frames.shape[0])):
        f = frames[i:i + 1]
        t0 = time.perf_counter()
        inp = R.build_inputs(f, f.copy(), CFG, oops.knn_search)
        logits = R.forward(sd, CFG, inp)
        t_total += time.perf_counter() - t0
        done += 1
        try:
            cpu_labels.append(np.asarray(logits.argmax(-1)).reshape(-1))
        except Exception:
            cpu_labels = None
        if t_total > budget_s:
            break
    out = {"value": done / t_total, "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "port",
           "sample": "%d frames of the same synthetic batch, oracle kd-tree kNN (OpenMP, %d threads) + "
                     "PyTorch-CPU forward restating the reference (%d threads)"
                     % (done, oops.num_threads(), torch.get_num_threads())}
    # second half of the headline metric ("+ mIoU parity vs CPU ref"): the labels of the GPU path against the oracle's
    # on the sampled frames (SemSegMetric-style IoU per class from the confusion matrix, mean over the classes present)
    try:
        if gpu_labels is not None and cpu_labels:
            g = np.concatenate([np.asarray(gpu_labels[i]).reshape(-1) for i in range(len(cpu_labels))]).astype(np.int64)
            c = np.concatenate(cpu_labels).astype(np.int64)
            nc = int(CFG["num_classes"])
            conf = np.bincount(c * nc + g, minlength=nc * nc).reshape(nc, nc).astype(np.float64)
            tp = np.diag(conf)
            denom = conf.sum(0) + conf.sum(1) - tp
            present = denom > 0
            out["miou_vs_cpu_oracle"] = float((tp[present] / denom[present]).mean())
            out["label_agreement"] = float((g == c).mean())
    except Exception:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-step", type=int, default=int(os.environ.get("ML3D_BENCH_BATCH", 64)))
    ap.add_argument("--distinct-frames", type=int, default=8,
                    help="distinct synthetic frames generated per rank (tiled with seeded rigid transforms)")
    ap.add_argument("--workload", choices=["randlanet", "kpconv", "pointpillars"], default="randlanet",
                    help="randlanet = BASELINE.json configs[1] (the headline metric); the other two are configs[2] / [3]")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the neighbour pyramid and the forward of a batch back to back on one stream instead of "
                         "overlapping batch i+1's pyramid with batch i's forward on two streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--breakdown", action="store_true", help="also time every kernel once (after the timed region)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE under torchrun")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    from ml3d import dist as mdist
    if world > 1:
        import torch.distributed as dist
        mdist.init("nccl", dev)

    if args.workload != "randlanet":
        import bench_models
        fn = bench_models.run_kpconv if args.workload == "kpconv" else bench_models.run_pointpillars
        out = fn(args, rank, world, dev, dist)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return out

    import synth_data
    from ml3d.engine import PipelinedRandLAEngine, RandLAInferenceEngine, make_trace

    B, N = args.frames_per_step, CFG["num_points"]
    # ---- synthetic frames: distinct sweeps per rank, tiled by seeded z-rotations to fill the batch
    nd = max(1, min(args.distinct_frames, B))
    base = np.stack([synth_data.semantickitti_patch(rank * 1000 + i, N) for i in range(nd)])
    rng = np.random.default_rng(77 + rank)
    frames = np.empty((B, N, 3), np.float32)
    for b in range(B):
        f = base[b % nd]
        if b >= nd:
            a = rng.uniform(0, 2 * np.pi)
            rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
            f = (f @ rot.T)[rng.permutation(N)]
        frames[b] = f
    sd = synth_weights.randlanet_state_dict(CFG, 2024)   # deterministic pseudo-trained weights (no checkpoints offline)
    overlap = not args.no_overlap
    eng = PipelinedRandLAEngine(CFG, sd, B, N, dev) if overlap else RandLAInferenceEngine(CFG, sd, B, N, dev)
    pts = torch.from_numpy(frames).to(dev)
    feats = pts.clone()   # in_channels = 3: features are the xyz themselves (randlanet.py:208-209)
    # predicted labels travel as uint8 (19 classes); two buffers so the gather of step i overlaps step i + 1
    lab_dtype = torch.uint8 if CFG["num_classes"] <= 256 else torch.int32
    labels = [torch.empty((B, N), dtype=lab_dtype, device=dev) for _ in range(2)]
    recv = [[torch.empty_like(labels[0]) for _ in range(world)] if (world > 1 and rank == 0) else None for _ in range(2)]
    pending = [None, None]
    step_no = [0]

    def one_step(knn_trace=None, fwd_trace=None):
        if overlap:
            scores = eng.submit(pts, feats, knn_trace, fwd_trace)
            if world > 1:
                torch.cuda.current_stream().wait_stream(eng.compute)
        else:
            scores = eng.step(pts, feats, knn_trace, fwd_trace)
        if world > 1:      # the only data-path collective: predicted labels of every rank's frames -> rank 0
            i = step_no[0] & 1
            step_no[0] += 1
            if pending[i] is not None:
                pending[i].wait()
            labels[i].copy_(torch.argmax(scores, dim=2))
            _, pending[i] = mdist.gather_predictions(labels[i], dst=0, out=recv[i], async_op=True)

    def drain():
        if overlap:
            eng.synchronize()
        for w in pending:
            if w is not None:
                w.wait()
        pending[0] = pending[1] = None

    for _ in range(args.warmup):
        one_step()
    drain()
    K = args.steps
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    kev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    for a, b in ev + kev:
        a.record(); b.record()           # materialise the hipEvent_t handles
    traces = [make_trace(DOMINANT_FWD_TAG, a, b) for a, b in ev]
    ktraces = [make_trace(0, a, b) for a, b in kev]   # tag 0: 16-NN query kernel of level 0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        one_step(ktraces[i], traces[i])
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dom_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    knn_ms = float(np.mean([a.elapsed_time(b) for a, b in kev]))

    out = None
    if rank == 0:
        layer, stage = DOMINANT_FWD_TAG // 8, DOMINANT_FWD_TAG % 8
        n_l = eng.n[layer] * B
        flops = lfa_flops(CFG, layer, stage, n_l)
        flops_exec = lfa_flops_executed(CFG, layer, stage, n_l)
        achieved = flops / (dom_ms * 1e-3) / 1e12
        # HBM-side bytes per launch of that kernel from the PMC passes (FETCH_SIZE x2 gfx950 correction +
        # WRITE_SIZE, KiB -> bytes; profiles/r01_*_pmc_{fetch,write}.csv), scaled to this run's frames per step
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj["dominant_bytes_per_launch_per_frame"] * B
            except Exception:
                traffic = None
        # secondary: the merged 16-NN query launch (all pyramid levels) against the HBM roofline (algorithmic bytes)
        kb = sum((12 + 4 * 16) * n_l for n_l in eng.n[:CFG["num_layers"]]) * B
        out = {
            "metric": "point-cloud frames/sec (RandLA-Net SemanticKITTI inference: kNN pyramid + forward)",
            "value": B * K * world / dt, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RandLA-Net SemanticKITTI inference, %d synthetic 45056-point frames per step per GPU "
                                   "(randlanet_semantickitti.yml), GPU kNN pyramid + fused forward" % B,
                       "frames_per_step_per_gpu": B, "num_points": N, "parallelism": "frame-parallel x%d" % world},
            "roofline": {"bound": "mfma", "kernel": "lfa_attn_wave<%d,%d> (layer %d)" % (CFG["dim_output"][layer], stage, layer),
                         "achieved": achieved, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_TFLOPS, "traffic": traffic, "avg_launch_ms": dom_ms,
                         "flops_per_launch": flops,
                         # same launch priced on the flops it executes after the algebraic split of the score Linear
                         "executed_flops_per_launch": flops_exec,
                         "frac_executed": flops_exec / (dom_ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS},
            "roofline_knn": {"bound": "hbm", "kernel": "knn_query_multi<16> (all pyramid levels)", "achieved": kb / (knn_ms * 1e-3) / 1e9,
                             "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": kb / (knn_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                             "avg_launch_ms": knn_ms, "bytes_per_launch": kb, "traffic": None},
        }
        if args.breakdown:
            bd = {}
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            tags_f = [1000] + [8 * l + s for l in range(CFG["num_layers"]) for s in range(8)] + [1001] + \
                     [1100 + i for i in range(CFG["num_layers"])] + [1200, 1201, 1202]
            e1 = eng.eng[0] if overlap else eng        # kernels timed one at a time, nothing else on the GPU
            for tg in tags_f:
                e1.step(pts, feats, None, make_trace(tg, a, b)); torch.cuda.synchronize()
                bd["fwd:%d" % tg] = a.elapsed_time(b)
            for tg in [100 + l for l in range(CFG["num_layers"] + 1)] + list(range(2 * CFG["num_layers"])):
                e1.step(pts, feats, make_trace(tg, a, b), None); torch.cuda.synchronize()
                bd["knn:%d" % tg] = a.elapsed_time(b)
            out["breakdown_ms"] = bd
        if not args.no_cpu_baseline and world == 1:
            gpu_labels = None
            try:        # labels of the first frames of the batch from one more (untimed) pass of the same engine
                e1 = eng.eng[0] if overlap else eng
                sc = e1.step(pts, feats)
                torch.cuda.synchronize()
                gpu_labels = torch.argmax(sc[:6], dim=2).cpu().numpy()
            except Exception:
                gpu_labels = None
            out["cpu_baseline"] = cpu_baseline(frames, sd, gpu_labels=gpu_labels)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
