#!/usr/bin/env python
"""bench.py — RandLA-Net SemanticKITTI inference frames/s on MI355X (BASELINE.json configs[1]).

A step = one pass of the hot path over one batch of synthetic SemanticKITTI-shaped frames: upload of the batch's xyz
from pinned host memory (inside the timed region, on a copy stream), GPU neighbour pyramid (4x 16-NN +
4x 1-NN, replaces the CPU knn_search calls of RandLANet.transform) and the fused RandLA-Net forward -> logits
[B, 45056, 19] -> argmax -> uint8 labels (ml3d.dist.PredictionGather, on its own stream; the same step at every N).  N > 1:
one process per GPU (torchrun), frames sharded across ranks (weak scaling); the only collective is the RCCL gather of those
labels to rank 0 inside the timed region.

``python bench.py --gpus N`` WITHOUT torchrun launches its own N ranks (``launch_ranks``: one child process per GPU with the
torchrun environment, rendezvous on 127.0.0.1 -- what the reference's scripts/run_pipeline.py:195-206 does with mp.spawn);
under torchrun (RANK / WORLD_SIZE set) it is one rank.  N ranks on fewer than N devices is an error, never a silent N = 1.

Prints ONE JSON line (rank 0).  Extra objects:
  ranks_seen      world size as torch.distributed reports it + the all-gathered device identities of the ranks.
  roofline        the kernel with the longest average launch among the traced ones (the neighbour-search launch, layer-0 and layer-1
                  attention kernels), timed live with HIP events recorded by the library around that kernel's launch on the
                  launch stream; `roofline_other` carries the rest.  MFMA kernels: `frac` prices the flops the kernel
                  EXECUTES, `frac_reference_formulation` the flops of the reference's formulation of the same result.
  cpu_baseline    the CPU oracle (port of the reference path) on this box's host cores, best of a thread sweep.
  latency         the model-class API (transform -> batcher -> forward -> update_probs) at the YAMLs' batch sizes 1 and 4:
                  per-frame median / p95 over >= 200 frames (N = 1).
  workloads       BASELINE.json configs[2] / [3] (KPConv Toronto3D, PointPillars KITTI): `bench.py --workload ...` run as a child
                  process each, after the headline measurement (N = 1).
"""
import argparse
import json
import os
import re
import sys
import time

# The step runs on five HIP streams (caller, upload, neighbour search, forward, label post-processing).  ROCm multiplexes
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in first-use order, and two streams that share a queue run
# back to back: with five streams on four queues whether the search overlaps the forward (+9 % frames/s) was decided by
# which stream happened to be touched first (measured: profiles/r03_search_gate_ab.log).  Eight queues: one per stream.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "open3d-ml_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

import synth_weights

CFG = dict(synth_weights.RANDLANET_SEMANTICKITTI_CFG)  # randlanet_semantickitti.yml:17-33

PEAK_F32_TFLOPS = 157.3
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA (MI355X_MICROARCH.md)   # MI355X_MICROARCH.md: f32 vector == f32-input MFMA peak
PEAK_HBM_GBS = 8000.0

# kernels traced live (library tag -> description).  k-NN tag 0 = the merged neighbour-search launch;
# forward tags: 8 * layer + stage (1 / 2 = the two attention kernels of the layer)
TRACED = [("knn", 0), ("fwd", 8 * 0 + 1), ("fwd", 8 * 1 + 2)]


def lfa_flops(cfg, layer, stage, n_points):
    """Algorithmic flops of what ONE attention-kernel launch computes in the REFERENCE's formulation (2 x MACs of its
    matmuls), per point with K neighbours: stage 1 = lse1 MLP (10 -> h) + score Linear (d x d) +
    softmax-weighted sum (d); stage 2 = lse2 MLP (h x h) + score Linear + weighted sum."""
    d = cfg["dim_output"][layer]
    h = d // 2
    K = cfg["num_neighbors"]
    mac = K * (10 * h if stage == 1 else h * h) + K * d * d + K * d
    return 2.0 * mac * n_points


def lfa_flops_executed(cfg, layer, stage, n_points):
    """Flops the attention kernel EXECUTES for the same result: the feature half of the score Linear is linear in a
    per-point quantity, W . [f[nb] ; r] = (W_top . f)[nb] + W_bot . r, so the kernel gathers a precomputed per-point
    row and multiplies only the position half (K x h x d); stage 2 re-computes lse1 (K x 10 x h)."""
    d = cfg["dim_output"][layer]
    h = d // 2
    K = cfg["num_neighbors"]
    mac = K * 10 * h + (K * h * h if stage == 2 else 0) + K * h * d + K * d
    return 2.0 * mac * n_points


def knn_bytes(cfg, n_levels, batch):
    """Algorithmic HBM bytes of the neighbour-search launch (SURVEY.md §8d row a1, int32 indices): per level the self
    k-NN (12 B of xyz read + 4 * k B written per query) and the 1-NN interpolation search (xyz of both levels read, 4 B
    written per query) -- 5.69 MB per 45056-point frame."""
    tot = 0
    for l in range(cfg["num_layers"]):
        n, nn = n_levels[l], n_levels[l + 1]
        tot += 12 * n + 4 * cfg["num_neighbors"] * n
        tot += 12 * (n + nn) + 4 * n
    return tot * batch


def _sq(kernel_key):
    """VALU-issue roofline of a kernel from the committed SQ counter pass (profiles/traffic.json "sq", tools/make_traffic.py --sq):
    SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles; None if not collected."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return tj["sq"][kernel_key]
    except Exception:
        return None


TRAFFIC_SOURCE = ("committed PMC pass (profiles/traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of the op alone, "
                  "2 x FETCH_SIZE + WRITE_SIZE, scaled to this batch) -- NOT measured in this run")


def _traffic(kernel_key, batch):
    """HBM-side bytes per launch from the committed PMC passes (profiles/traffic.json: per-frame figures)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return float(tj["kernels"][kernel_key]["bytes_per_launch_per_frame"]) * batch
    except Exception:
        return None


def cpu_baseline(frames, sd, budget_s=24.0, max_frames=6, gpu_labels=None):
    """The CPU oracle (port of the reference path: kd-tree knn_search + PyTorch-CPU forward), timed on a bounded sample
    of the same frames.  Thread count: best of a sweep on one frame (oversubscribing a big host slows torch's small
    per-layer matmuls down), then the sample is timed at that setting."""
    from oracle import ops as oops
    from oracle import randlanet_ref as R
    ncpu = os.cpu_count() or 1
    R.forward(sd, CFG, R.build_inputs(frames[:1, :2048].copy(), frames[:1, :2048].copy(), CFG, oops.knn_search))

    def one(i):
        f = frames[i:i + 1]
        t0 = time.perf_counter()
        logits = R.forward(sd, CFG, R.build_inputs(f, f.copy(), CFG, oops.knn_search))
        return time.perf_counter() - t0, logits

    sweep = {}
    t_sweep0 = time.perf_counter()
    # (capped at 64 threads: on the 256-thread GPU box one frame at 256 threads took 44 s in round 2 -- pure oversubscription)
    for nt in sorted({t for t in (8, 16, 32, 64, min(ncpu, 64)) if t <= ncpu}):
        torch.set_num_threads(nt)
        sweep[nt] = one(0)[0]
        if time.perf_counter() - t_sweep0 > budget_s * 0.5:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    done, t_total, cpu_labels = 0, 0.0, []
    for i in range(min(max_frames, frames.shape[0])):
        dt, logits = one(i)
        t_total += dt
        done += 1
        cpu_labels.append(np.asarray(logits.argmax(-1)).reshape(-1))
        if t_total > budget_s * 0.5:
            break
    out = {"value": done / t_total, "unit": "frames/s", "cores": int(best), "kind": "port",
           "host_threads_available": int(ncpu),
           "thread_sweep_s_per_frame": {str(k): round(v, 3) for k, v in sweep.items()},
           "sample": "%d frames of the same synthetic batch: oracle kd-tree kNN (OpenMP, %d threads) + PyTorch-CPU forward "
                     "restating the reference, torch threads = %d (best of the sweep)" % (done, oops.num_threads(), best)}
    # second half of the headline metric ("+ mIoU parity vs CPU ref"): the labels of the GPU path against the oracle's
    # on the sampled frames (SemSegMetric-style IoU per class from the confusion matrix, mean over the classes present)
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
    return out


def latency(dev, sd, frames_timed=200, frames_warm=20):
    """Per-frame latency of the MODEL-CLASS API at the YAMLs' small batch sizes (randlanet_semantickitti.yml:38-45:
    ``test_batch_size: 1``, ``batch_size: 4`` -- what ``run_inference`` / ``run_test`` of the reference's pipeline issue):
    ``RandLANet.transform`` (patch sampler query + crop + recentre + GPU neighbour pyramid) -> ``DefaultBatcher`` ->
    ``forward`` -> ``update_probs`` (softmax + float16 vote update), one synthetic SemanticKITTI sweep, host-timed with a
    device synchronisation per step (>= 20 warm-up and >= 200 timed frames, median and p95)."""
    import synth_data
    from ml3d.torch.dataloaders import DefaultBatcher
    from ml3d.torch.models import RandLANet
    cfg = dict(CFG, grid_size=0.06, augment={"recenter": {"dim": [0, 1]}})
    model = RandLANet(**cfg, device=dev, seed=5)
    model.load_state_dict(sd)
    sweep = synth_data.lidar_sweep(5000)
    t0 = time.perf_counter()
    model.inference_begin(dict(point=sweep, feat=None, label=np.zeros(sweep.shape[0], np.int32)))
    torch.cuda.synchronize()
    pre_ms = (time.perf_counter() - t0) * 1e3
    attr = {"split": "test"}
    collate = DefaultBatcher().collate_fn
    out = {"api": "RandLANet.transform -> DefaultBatcher -> forward -> update_probs (float16 votes on the device)",
           "hip_graphs": bool(getattr(model, "use_graphs", False)),
           "cloud_points_raw": int(sweep.shape[0]), "cloud_points_sub": int(model.inference_data["point"].shape[0]),
           "preprocess_ms_once_per_cloud": pre_ms, "frames_timed": frames_timed, "frames_warmup": frames_warm}
    for B in (1, 4):
        def step():
            items = [{"data": model.transform(model.inference_data, attr), "attr": attr} for _ in range(B)]
            inputs = collate(items)
            scores = model(inputs["data"])
            model.update_probs(inputs, scores, model.test_probs)
            torch.cuda.synchronize()
        for _ in range(max(1, frames_warm // B)):
            step()
        prof = None
        if os.environ.get("ML3D_BENCH_PROFILE"):        # debugging aid: where the host time of a step goes
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        ts = []
        for _ in range(max(1, frames_timed // B)):
            t0 = time.perf_counter()
            step()
            ts.append((time.perf_counter() - t0) * 1e3 / B)
        if prof is not None:
            import pstats
            prof.disable()
            print("---- latency profile, batch %d ----" % B, file=sys.stderr)
            pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(22)
        ts = np.asarray(ts)
        out["batch_%d" % B] = {"ms_per_frame_median": float(np.median(ts)), "ms_per_frame_p95": float(np.percentile(ts, 95)),
                               "frames_per_s": float(1e3 / np.median(ts)), "steps": int(ts.size)}
    st = getattr(model, "_dev_loop", None) or {}
    out["hip_graphs_captured"] = {"patch": st.get("graph") is not None, "forward": st.get("fwd_graph") is not None,
                                  "failed": st.get("graph_failed") or st.get("fwd_failed")}
    return out


def synthetic_batch(rank, B, N, nd):
    """Distinct synthetic sweeps per rank, tiled by seeded z-rotations + re-shuffles to fill the batch."""
    import synth_data
    nd = max(1, min(nd, B))
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
    return frames


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv, stub=False):
    """``python bench.py --gpus N`` without torchrun: start N ranks of this script (one process per GPU, torchrun's
    environment, rendezvous on 127.0.0.1), wait for all of them, return the worst exit code.  Rank 0's stdout is ours, so
    the ONE JSON line still is.  The reference launches its ranks the same way (scripts/run_pipeline.py:195-206)."""
    import subprocess
    if not stub:
        nd = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if nd < n:
            raise SystemExit("bench.py --gpus %d: %d ranks, %d device%s visible -- refusing to run fewer ranks than asked for"
                             % (n, n, nd, "" if nd == 1 else "s"))
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, ML3D_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stdin=subprocess.DEVNULL))
    rc = 0
    try:
        # a rank that dies leaves the others in a collective: poll, and take the rest down (by PID) when one has failed
        alive = list(procs)
        while alive:
            for p in list(alive):
                code = p.poll()
                if code is None:
                    continue
                alive.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in alive:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def device_identity(dev):
    """What tells two ranks' GPUs apart (the UUID where this torch exposes it, the PCI address otherwise)."""
    if dev.type != "cuda":
        return "cpu-stub:pid%d" % os.getpid()
    pr = torch.cuda.get_device_properties(dev)
    uuid = getattr(pr, "uuid", None)
    if uuid is not None:
        return "%s uuid=%s" % (pr.name, uuid)
    return "%s pci=%04x:%02x:%02x" % (pr.name, getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                      getattr(pr, "pci_device_id", 0))


def emit(out):
    """The ONE JSON line, as the LAST line of stdout: RCCL writes its version banner through C stdio, which a pipe or a file buffers until
    the process exits -- i.e. it would land BEHIND a line printed from Python (seen on the MI355X: tools/r06_calls/call_aa.sh).  Everything
    C has buffered goes out first."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def flush_c():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def finish(out, rank, world, dist):
    """Leave the process group FIRST, then let rank 0 print: every rank's stdout ends up in one stream under torchrun, and whatever a rank's
    RCCL still holds in C stdio is written when that rank exits -- behind rank 0's JSON line if that was printed before the group was torn
    down.  All ranks flush C stdio after the teardown; at N > 1 rank 0 gives the others half a second to exit."""
    if world > 1 or FORCE_GROUP:
        dist.barrier()
        dist.destroy_process_group()
        flush_c()
        if rank == 0 and world > 1:
            time.sleep(0.5)
    if rank == 0 and out is not None:
        emit(out)


# ML3D_DIST_FORCE_GROUP=1: join a process group and take the gather / barrier / all-reduce / checksum path of N > 1 even at world size 1 --
# the only way to put the RCCL calls of the multi-GPU path on real hardware from a one-GPU box (tools/r06_calls/call_aa.sh)
FORCE_GROUP = os.environ.get("ML3D_DIST_FORCE_GROUP") == "1"


def ranks_seen(dev, world, dist, stub=False, placement=None):
    """{"world_size", "devices", "cpu_map"}: the process group's own view of the job (every rank's device identity and host
    placement -- NUMA node of its GPU, the CPUs it is pinned to, its host thread count -- all-gathered).
    Two ranks on one GPU would share its HBM and CUs and still print n_gpus = N: that is an error here."""
    me = (device_identity(dev), placement)
    if (world > 1 or FORCE_GROUP):
        both = [None] * world
        dist.all_gather_object(both, me)
        ws = dist.get_world_size()
    else:
        both, ws = [me], 1
    ids = [b[0] for b in both]
    if not stub and len(set(ids)) != len(ids):
        raise SystemExit("bench.py: %d ranks share devices %s -- one process per GPU" % (ws, ids))
    return {"world_size": int(ws), "devices": ids, "cpu_map": [b[1] for b in both]}


class _HostEvent:
    """--stub stand-in for torch.cuda.Event (the stub runs the step logic on CPU tensors over gloo)."""

    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class _StubFrameStream:
    """--stub stand-in for ml3d.engine.RandLAFrameStream: scores are a pure function of (rank, step) on the CPU, so rank 0
    can rebuild what every rank must have sent.  Everything around it -- launcher, process group, PredictionGather, the
    barriers and the max-over-ranks clock -- is the code the GPU run executes."""
    compute_stream = None

    def __init__(self, rank, B, N, C):
        self.rank, self.B, self.N, self.C, self.k = rank, B, N, C, 0
        self.n = [N]

    @staticmethod
    def scores_of(rank, step, B, N, C):
        return torch.rand((B, N, C), generator=torch.Generator().manual_seed(7919 * rank + step))

    def submit(self, host, feats=None, knn_trace=None, fwd_trace=None, done=None):
        sc = self.scores_of(self.rank, self.k, self.B, self.N, self.C)
        self.k += 1
        if done is not None:
            done.record()
        return sc

    def synchronize(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=None,
                    help="units per step and GPU; default: 128 frames (randlanet; ML3D_BENCH_BATCH overrides), 96 spheres (kpconv), "
                         "32 sweeps (pointpillars)")
    ap.add_argument("--distinct-frames", type=int, default=8,
                    help="distinct synthetic frames generated per rank (tiled with seeded rigid transforms)")
    ap.add_argument("--workload", choices=["randlanet", "kpconv", "pointpillars"], default="randlanet",
                    help="randlanet = BASELINE.json configs[1] (the headline metric); the other two are configs[2] / [3]")
    ap.add_argument("--no-overlap", action="store_true",
                    help="upload, neighbour pyramid and forward of a batch back to back on one stream instead of the "
                         "three-stream pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true", help="skip the KPConv / PointPillars side measurements")
    ap.add_argument("--no-latency", action="store_true", help="skip the small-batch (B = 1 / 4) model-API latency measurement")
    ap.add_argument("--breakdown", action="store_true", help="also time every kernel once (after the timed region)")
    ap.add_argument("--train", action="store_true",
                    help="SURVEY.md §8 f4: time a DATA-PARALLEL TRAINING step of RandLA-Net instead of inference -- forward + loss + backward "
                         "on the HIP training kernels, DistributedDataParallel's gradient all-reduce over RCCL (nccl) at N > 1 -- "
                         "and check that every rank ends the step with the same gradients")
    ap.add_argument("--stub", action="store_true",
                    help="CPU / gloo dry run of the launcher and of every N > 1 branch with a stand-in for the GPU step "
                         "(tests/test_bench_launcher.py); prints a line marked \"stub\": true that is NOT a measurement")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no torchrun around us: be the launcher (N ranks, one per GPU), never a silent single rank
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], stub=args.stub))

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", args.gpus)))
    # host placement: every rank is pinned to its own slice of the CPUs of ITS GPU's NUMA node (/sys/bus/pci/devices/<id>/
    # numa_node), and its host-side glue (collate, small CPU tensor ops, the launches and read-backs of a KPConv batch build --
    # that step is host-bound) uses at most 16 of them: an OpenMP fork-join across a 256-thread host costs milliseconds per
    # tiny op, and 8 ranks share the node (cpu_baseline runs its own thread sweep at N = 1)
    from ml3d import dist as mdist
    pci = None
    if not args.stub and torch.cuda.is_available() and torch.cuda.device_count() >= local_world:
        pci = []
        for i in range(local_world):
            pr = torch.cuda.get_device_properties(i)
            pci.append("%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                             getattr(pr, "pci_device_id", 0)))
    placement = mdist.bind_rank(local, local_world, pci)
    HOST_THREADS = placement["host_threads"]
    torch.set_num_threads(HOST_THREADS)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    stub = args.stub
    if stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
        if local >= torch.cuda.device_count():
            raise SystemExit("bench.py: local rank %d of %d ranks, %d device(s) visible -- one process per GPU"
                             % (local, world, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    if (world > 1 or FORCE_GROUP):
        import torch.distributed as dist
        mdist.init("gloo" if stub else "nccl", dev)
    seen = ranks_seen(dev, world, dist, stub, placement)

    if args.train:
        import bench_models
        out = bench_models.run_randlanet_train(args, rank, world, dev, dist)
        if rank == 0:
            out["ranks_seen"] = seen
        finish(out if rank == 0 else None, rank, world, dist)
        return out
    if args.workload != "randlanet":
        import bench_models
        fn = bench_models.run_kpconv if args.workload == "kpconv" else bench_models.run_pointpillars
        out = fn(args, rank, world, dev, dist)
        if rank == 0:
            out["ranks_seen"] = seen
        finish(out if rank == 0 else None, rank, world, dist)
        return out

    from ml3d.engine import RandLAFrameStream, make_trace

    # 128 frames per step (round 5; 64 until then): same box, alternating -- 6153 / 6220 frames/s at 64, 6292 at 96, 6301 / 6316 at 128,
    # 6273 at 192 (profiles/r05_batch_sweep.log): the tails of the step's ~50 launches are amortised over twice the work
    B, N = args.frames_per_step or int(os.environ.get("ML3D_BENCH_BATCH", 128)), CFG["num_points"]
    overlap = not args.no_overlap and not stub
    if stub:
        N = 257
        stream, hosts = _StubFrameStream(rank, B, N, CFG["num_classes"]), [None, None]
        Event, sync = _HostEvent, (lambda: None)
    else:
        Event, sync = torch.cuda.Event, torch.cuda.synchronize
        frames = synthetic_batch(rank, B, N, args.distinct_frames)
        sd = synth_weights.randlanet_state_dict(CFG, 2024)   # deterministic pseudo-trained weights (no checkpoints offline)
        stream = RandLAFrameStream(CFG, sd, B, N, dev, overlap=overlap)
        # what a data loader hands over: pinned host xyz (in_channels = 3).  Two DIFFERENT batches alternate (the second holds
        # the frames in reverse order), so consecutive steps never upload / search / classify the same bytes.
        hosts = [torch.from_numpy(frames).pin_memory(), torch.from_numpy(np.ascontiguousarray(frames[::-1])).pin_memory()]
    step_no = [0]
    gather = mdist.PredictionGather(B, N, CFG["num_classes"], dev)

    # Every step ends with the argmax of its scores ("forward + softmax/argmax" per frame) into a uint8 label
    # buffer and -- the only data-path collective -- the gather of those labels to rank 0.  At N = 1 the gather degenerates to
    # nothing, the argmax and the stream choreography are the SAME code, so N = 1 and N > 1 time the same step.  Both run on
    # their OWN stream behind that step's forward.  (Making the caller's stream wait for the compute stream instead would
    # also hold back the NEXT step's upload and neighbour search, which are ordered after the caller's stream: the
    # search-under-forward overlap would be lost exactly when scaling is measured.)
    post = torch.cuda.Stream(device=dev) if overlap else None
    labelled = [torch.cuda.Event(), torch.cuda.Event()] if overlap else None

    def one_step(knn_trace=None, fwd_trace=None, done=None):
        slot = step_no[0] & 1
        if post is not None and step_no[0] >= 2:
            # the pipelined engine has two score buffers in ping-pong: this step's forward overwrites the one whose argmax was
            # enqueued TWO steps ago -- wait for that one only (waiting for the previous step's argmax, as until round 3, put
            # 0.15 ms of label post-processing between consecutive forwards)
            stream.compute_stream.wait_event(labelled[slot])
        scores = stream.submit(hosts[slot], None, knn_trace, fwd_trace, done)
        step_no[0] += 1
        if post is not None:
            with torch.cuda.stream(post):
                post.wait_stream(stream.compute_stream)          # this step's forward
                gather.push(scores)
                labelled[slot].record(post)
        else:
            gather.push(scores)

    def drain():
        stream.synchronize()
        if post is not None:
            post.synchronize()
        gather.drain()

    for _ in range(args.warmup):
        one_step()
    drain()
    K = args.steps
    tev = [(Event(enable_timing=True), Event(enable_timing=True)) for _ in range(K)]
    done_ev = [Event(enable_timing=True) for _ in range(K + 1)]
    for a, b in tev:
        a.record(); b.record()           # materialise the hipEvent_t handles
    sync()
    if (world > 1 or FORCE_GROUP):
        dist.barrier()
    sync()
    done_ev[0].record(stream.compute_stream)
    t0 = time.perf_counter()
    for i in range(K):
        kind, tag = TRACED[i % len(TRACED)]
        tr = None if stub else make_trace(tag, *tev[i])
        one_step(tr if kind == "knn" else None, tr if kind == "fwd" else None, done_ev[i + 1])
    drain()
    sync()
    if (world > 1 or FORCE_GROUP):
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if (world > 1 or FORCE_GROUP):
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # N > 1 on the hardware: the first multi-GPU run is also a CORRECTNESS run -- rank 0 recomputes, on the label buffers the RCCL
    # gather delivered for the LAST step, the checksums every rank computed on its own labels (outside the timed region)
    gather_check = None
    if (world > 1 or FORCE_GROUP):
        last_slot = (args.warmup + K - 1) % gather.depth
        mine = mdist.label_checksum(gather.labels[last_slot])
        allsums = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allsums, mine)
        if rank == 0:
            got = gather.gathered(last_slot)
            ok = [bool(torch.equal(mdist.label_checksum(got[r]), allsums[r])) for r in range(world)]
            gather_check = {"ranks_checked": world, "all_match": all(ok), "per_rank": ok,
                            "what": "rank 0 recomputed every rank's label checksum (sum, position-weighted sum, xor-fold) on the buffers "
                                    "the RCCL gather delivered for the last timed step"}
    out = None
    if stub:
        # the dry run's check: rank 0 must hold every rank's labels of the LAST step (the async gather's second buffer)
        if rank == 0:
            last = args.warmup + K - 1
            got = gather.gathered(last % gather.depth)
            assert len(got) == world, (len(got), world)
            for r in range(world):
                want = _StubFrameStream.scores_of(r, last, B, N, CFG["num_classes"]).argmax(2).to(torch.uint8)
                assert torch.equal(got[r], want), "rank %d's labels did not arrive on rank 0" % r
            out = {"stub": True, "metric": "NOT A MEASUREMENT: bench.py --stub (launcher + N > 1 step logic on CPU tensors over gloo)",
                   "value": B * K * world / dt, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
                   "ms_per_step": dt / K * 1e3, "scaling": "weak", "ranks_seen": seen,
                   "self_launched": bool(os.environ.get("ML3D_BENCH_SELF_LAUNCHED")),
                   "gathered_ranks_checked": world, "gather_self_check": gather_check}
    elif rank == 0:
        n_lv = stream.n
        # the same three kernels with NOTHING else on the GPU (after the timed region): what the overlap with the other stream
        # costs each of them -- the neighbour-search launch runs under the forward on purpose and takes ~1.5x its solo time there
        alone = {}
        try:
            e1 = stream.single_engine()
            a_ev, b_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_ev.record(); b_ev.record()
            torch.cuda.synchronize()
            for kind, tag in TRACED:
                ts = []
                for _ in range(3):
                    tr = make_trace(tag, a_ev, b_ev)
                    e1.step(stream.pts[0], stream.pts[0], tr if kind == "knn" else None, tr if kind == "fwd" else None)
                    torch.cuda.synchronize()
                    ts.append(a_ev.elapsed_time(b_ev))
                alone[(kind, tag)] = float(np.median(ts))
        except Exception:
            alone = {}
        # per-step intervals on the compute stream (completion of step i-1 -> completion of step i)
        iv = np.array([done_ev[i].elapsed_time(done_ev[i + 1]) for i in range(K)])
        per_tag = {}
        for i in range(K):
            per_tag.setdefault(TRACED[i % len(TRACED)], []).append(tev[i][0].elapsed_time(tev[i][1]))
        cands = []
        for (kind, tag), ts in per_tag.items():
            ms = float(np.mean(ts))
            if kind == "knn":
                kb = knn_bytes(CFG, n_lv, B)
                cands.append({"bound": "hbm", "kernel": "knn_query_multi<16, true> (16-NN + prefix 1-NN of all pyramid levels, one launch)",
                              "achieved": kb / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": kb / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": _traffic("knn_query_multi<16, true>", B),
                              "avg_launch_ms": ms, "bytes_per_launch": kb, "avg_launch_ms_alone": alone.get((kind, tag)),
                              "frac_alone": (kb / (alone[(kind, tag)] * 1e-3) / 1e9 / PEAK_HBM_GBS) if alone.get((kind, tag)) else None,
                              "note": "algorithmic bytes per SURVEY.md §8d (5.69 MB / frame); the search is VALU-issue bound, "
                                      "not HBM-bound (valu_frac below; DESIGN.md §3.2)"})
                sq = _sq("knn_query_multi<16, true>")
                if sq:
                    # the bound the kernel actually sits on: SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles (PMC pass of the launch alone)
                    cands[-1]["valu_frac"] = sq.get("valu_frac_of_busy_cycles")
                    cands[-1]["valu_frac_at_2p4ghz"] = sq.get("valu_frac_at_2p4ghz")
                    # (the PMC pass profiles a 64-frame launch: tools/knn_only.py; the count scales with the frames)
                    cands[-1]["valu_insts_per_launch"] = sq.get("insts_valu_per_launch") and sq["insts_valu_per_launch"] * B / 64.0
                    cands[-1]["valu_source"] = "profiles/%s" % sq.get("source")
            else:
                layer, stage = tag // 8, tag % 8
                d = CFG["dim_output"][layer]
                fl_ref = lfa_flops(CFG, layer, stage, n_lv[layer] * B)
                fl_ex = lfa_flops_executed(CFG, layer, stage, n_lv[layer] * B)
                name = ("lfa_attn_mfma16<%d>" % stage) if d == 16 else ("lfa_attn_wave_b3<%d,%d>" % (d, stage))
                cands.append({"bound": "mfma", "kernel": "%s (layer %d)" % (name, layer),
                              "achieved": fl_ex / (ms * 1e-3) / 1e12, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                              "frac": fl_ex / (ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                              "frac_reference_formulation": fl_ref / (ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                              "traffic": _traffic(name, B), "avg_launch_ms": ms, "avg_launch_ms_alone": alone.get((kind, tag)),
                              "executed_flops_per_launch": fl_ex,
                              "reference_flops_per_launch": fl_ref})
                if d != 16:
                    # round 6: the deep products of this stage (lse2, the score Linear's position half) run on the bf16 pipe as six
                    # bf16 MFMAs per 16-deep step (three-way split of both operands, float32-equivalent): `achieved` / `frac` stay
                    # float32-equivalent flops against the f32 matrix peak; the bf16 flops the pipe actually executes are 3x the
                    # float32-equivalent flops of those products (lse1 stays on the f32 MFMA)
                    cands[-1]["note"] = ("float32-equivalent flops; lse2 + score product executed as 6 bf16 MFMAs per 16-deep step "
                                         "(v_mfma_f32_32x32x16_bf16), dense bf16 peak %.0f TFLOP/s" % PEAK_BF16_TFLOPS)
        for c in cands:
            c["traffic_source"] = TRAFFIC_SOURCE if c.get("traffic") is not None else None
        cands.sort(key=lambda c: -c["avg_launch_ms"])
        out = {
            "metric": "point-cloud frames/sec (RandLA-Net SemanticKITTI inference: H2D + kNN pyramid + forward)",
            "value": B * K * world / dt, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "step_ms_median": float(np.median(iv)), "step_ms_p95": float(np.percentile(iv, 95)),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "ranks_seen": seen,
            "config": {"workload": "RandLA-Net SemanticKITTI inference, %d synthetic 45056-point frames per step per GPU "
                                   "(randlanet_semantickitti.yml): host->device upload of xyz + GPU kNN pyramid + fused forward" % B,
                       "frames_per_step_per_gpu": B, "num_points": N, "parallelism": "frame-parallel x%d" % world,
                       "h2d_in_timed_region": True, "argmax_in_timed_region": True, "streams": 4 if overlap else 1},
            "roofline": cands[0], "roofline_other": cands[1:],
            # the whole step on the reference's formulation (~14 GFLOP per frame through LFA x 4, the per-point MLPs and
            # the decoder): what the GPU delivers end to end, searches, upload and argmax included -- independent of how streams overlap
            "end_to_end": {"tflops": 14.0e9 * (B * K * world / dt) / world / 1e12,
                           "frac_of_f32_mfma_peak": 14.0e9 * (B * K * world / dt) / world / 1e12 / PEAK_F32_TFLOPS,
                           "note": "14 GFLOP per frame (SURVEY.md §8d, reference formulation) x frames/s per GPU"},
        }
        if gather_check is not None:
            out["gather_self_check"] = gather_check
        if args.breakdown:
            bd = {}
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            tags_f = [1000] + [8 * l + s for l in range(CFG["num_layers"]) for s in range(8)] + [1001] + \
                     [1100 + i for i in range(CFG["num_layers"])] + [1200, 1201, 1202]
            e1 = stream.single_engine()           # kernels timed one at a time, nothing else on the GPU
            pts = stream.pts[0]
            for tg in tags_f:
                e1.step(pts, pts, None, make_trace(tg, a, b)); torch.cuda.synchronize()
                bd["fwd:%d" % tg] = a.elapsed_time(b)
            for tg in [100 + l for l in range(CFG["num_layers"])] + [0]:
                e1.step(pts, pts, make_trace(tg, a, b), None); torch.cuda.synchronize()
                bd["knn:%d" % tg] = a.elapsed_time(b)
            out["breakdown_ms"] = bd
        if not args.no_cpu_baseline and world == 1:
            e1 = stream.single_engine()           # labels of the first frames from one more (untimed) pass
            sc = e1.step(stream.pts[0], stream.pts[0])
            torch.cuda.synchronize()
            gpu_labels = torch.argmax(sc[:6], dim=2).cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(frames, sd, gpu_labels=gpu_labels)
            torch.set_num_threads(HOST_THREADS)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_latency:
            try:
                out["latency"] = latency(dev, sd)
            except Exception as e:          # a side measurement must never take the headline line down
                out["latency"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_workloads:
            # each side workload in its OWN process (`bench.py --workload ...`, what a deployment runs: one process per GPU and
            # model).  In-process after the headline they inherit its five HIP streams' hardware-queue assignments -- ROCm maps
            # streams to queues in first-use order -- and PointPillars' two lanes then share a queue: 1156 instead of 1390-1406
            # frames/s (gpurun_out/r3fin2 against r3p / r3q).
            import subprocess
            del stream
            torch.cuda.empty_cache()
            wl = {}
            for name in ("kpconv", "pointpillars"):
                # warm-up: 30 KPConv steps / 12 PointPillars steps.  The FIRST process on a cold box needs them: its caching-allocator
                # pools (two builder streams + two forward streams for KPConv) keep growing for ~25 steps, and every growth is a
                # hipMalloc that stalls the step for 20-80 ms (profiles/r05_kp_cold_box.log: 5822 spheres/s with 8 warm-up steps as
                # the first process on a box, 9071 / 9010 in the next two processes)
                # ... and timed regions of ~1.3 s (120 KPConv steps, 60 PointPillars steps) instead of 0.3 s: a cold box showed sporadic
                # GPU-wide stalls of 30-80 ms in its first process whatever the pipeline configuration (same log), and one of them
                # in a 0.3 s region is a 15-25 % error
                cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", "120" if name == "kpconv" else "60",
                       "--warmup", "30" if name == "kpconv" else "12"]       # (a fresh process: the
                # first steps still grow the allocator pools of both streams -- with 3 warm-up steps a cold box showed 13 ms outliers
                # among 8.8 ms steps)
                if args.no_cpu_baseline:
                    cmd.append("--no-cpu-baseline")
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, stdin=subprocess.DEVNULL)
                    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                    wl[name] = json.loads(lines[-1]) if lines else {"error": "rc %d: %s" % (r.returncode, r.stderr[-400:])}
                except Exception as e:      # a side measurement must never take the headline line down
                    wl[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["workloads"] = wl
            # the HBM-bound primitives each side workload timed in ITS timed region (a10 radius search, a11 grid
            # subsample, a15 voxelize, a17 pillar scatter) join this line's list, so that one line covers all eight components
            for name in ("kpconv", "pointpillars"):
                for e in (wl.get(name) or {}).get("roofline_other", []) or []:
                    out["roofline_other"].append(dict(e, workload=name))
    finish(out if rank == 0 else None, rank, world, dist)
    return out


if __name__ == "__main__":
    main()
