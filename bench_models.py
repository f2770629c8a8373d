"""bench.py --workload kpconv | pointpillars — the other two model paths of BASELINE.json (configs[2], configs[3]).

Same JSON contract as the RandLA-Net line (bench.py): `value` = units/s with inputs resident in HBM, `roofline` for
the dominant op timed live with HIP events on the launch stream (torch's current stream: these ops are launched
from Python through the C ABI on it), `cpu_baseline` = the CPU oracle (port of the reference path) on a bounded
sample.  Units: KPConv = input spheres (batch build on the GPU + forward), PointPillars = sweeps (voxelize +
pillar features + backbone + heads).
"""
import time

import os

import numpy as np
import torch

PEAK_F32_TFLOPS = 157.3
PEAK_BF16_TFLOPS = 2500.0          # MI355X_MICROARCH.md: dense bf16 MFMA (32x32x16), measured 2495
PEAK_HBM_GBS = 8000.0


def _hbm_entry(component, kernel, nbytes, ms_in, ms_alone, traffic_key, units, note, launches=None, extra=None):
    """One HBM-bound primitive of SURVEY.md §8(d) as a roofline object: ALGORITHMIC bytes of one batch-size launch / its event
    time (inside the timed region when there is one, alone otherwise) against the 8 TB/s peak; `traffic` = PMC bytes of the
    same op run alone (profiles/traffic.json, tools/roofline_ops.py), per launch."""
    ms = ms_in if ms_in else ms_alone
    e = {"bound": "hbm", "component": component, "kernel": kernel, "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
         "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": _traffic(traffic_key, units),
         "avg_launch_ms": ms, "timed": "inside the timed region (co-running streams)" if ms_in else "alone, after the timed region",
         "avg_launch_ms_alone": ms_alone,
         "frac_alone": (nbytes / (ms_alone * 1e-3) / 1e9 / PEAK_HBM_GBS) if ms_alone else None,
         "bytes_per_launch": float(nbytes), "units_per_launch": units, "note": note}
    e["traffic_source"] = TRAFFIC_SOURCE if e["traffic"] is not None else None
    if launches is not None:
        e["launches_timed"] = launches
    if extra:
        e.update(extra)
    return e


TRAFFIC_SOURCE = ("committed PMC pass (profiles/traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of the op alone, "
                  "2 x FETCH_SIZE + WRITE_SIZE, scaled to this batch) -- NOT measured in this run")


def _traffic(key, units):
    """HBM-side bytes per launch of the roofline op from the committed PMC passes (profiles/traffic.json: per-unit figures
    = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 / units, collected as MI355X_MICROARCH.md prescribes); None if not collected."""
    import json
    import os
    try:
        tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")))
        return float(tj["kernels"][key]["bytes_per_launch_per_frame"]) * units
    except Exception:
        return None


def _rocprof_name(key, default):
    """the kernel's name as rocprofv3 prints it in the committed PMC pass (profiles/traffic.json), so that the bench line and the
    profile tables name the same kernel"""
    import json
    import os
    try:
        tj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")))
        k = tj["kernels"][key]
        name = k.get("rocprof_name") or k["rocprof_names"][0]
        return name.replace("void ml3d::", "").replace("ml3d::", "").split("(")[0]
    except Exception:
        return default


class _CallTimer:
    """Wraps one function of ml3d.ops and brackets its `which`-th call of every step with HIP events."""

    def __init__(self, ops, name, which, method=False):
        """``method=True``: ``ops`` is a class and ``name`` one of its methods (installed as a plain function so that it binds)"""
        self.ops, self.name, self.which = ops, name, which
        self.orig = getattr(ops, name)
        self.count = 0
        self.events = []
        self.shapes = None
        if method:
            timer = self
            setattr(ops, name, lambda *a, **k: timer(*a, **k))
        else:
            setattr(ops, name, self)

    def new_step(self):
        self.count = 0

    def __call__(self, *a, **k):
        if self.count == self.which:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.orig(*a, **k)
            e1.record()
            self.events.append((e0, e1))
            self.shapes = (a, r)
            self.kwargs = k
        else:
            r = self.orig(*a, **k)
        self.count += 1
        return r

    def samples_ms(self):
        return [float(a.elapsed_time(b)) for a, b in self.events]

    def mean_ms(self):
        """median of the timed calls (a cold allocator block inside one call would dominate a mean of three)"""
        return float(np.median(self.samples_ms())) if self.events else None

    def restore(self):
        setattr(self.ops, self.name, self.orig)


def _timed(step, K, W, world, dist, dev, ev_stream=None, intervals=None):
    """``intervals`` (list): filled with the K completion-to-completion times (ms) of the steps on ``ev_stream``.
    (``dev`` = cpu only in bench.py --stub, the CPU / gloo dry run of the N > 1 branches: no events, no device syncs.)"""
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    if dev.type != "cuda":
        intervals = None
    for _ in range(W):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)] if intervals is not None else None
    if evs:
        evs[0].record(ev_stream() if callable(ev_stream) else ev_stream)
    t0 = time.perf_counter()
    for i in range(K):
        step()
        if evs:
            evs[i + 1].record(ev_stream() if callable(ev_stream) else ev_stream)
    sync()
    if evs:
        intervals.extend(float(evs[i].elapsed_time(evs[i + 1])) for i in range(K))
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


class _StubDetections:
    """bench.py --stub: stand-in for ml3d.engine.PointPillarsStream -- a step's detections are a pure function of (rank, step)
    with a DIFFERENT number of boxes on every rank and step, handed over one step late like the real stream."""
    compute = None

    def __init__(self, rank, sweeps):
        self.rank, self.sweeps, self.k, self.prev = rank, sweeps, 0, None

    @staticmethod
    def result_of(rank, step, sweeps):
        g = torch.Generator().manual_seed(104729 * rank + step)
        n = [3 + (5 * rank + 2 * step + i) % 7 for i in range(sweeps)]
        return ([torch.rand((k, 7), generator=g) for k in n], [torch.rand((k,), generator=g) for k in n],
                [torch.randint(0, 3, (k,), generator=g) for k in n])

    def submit(self, hosts):
        out, self.prev = self.prev, self.result_of(self.rank, self.k, self.sweeps)
        self.k += 1
        return out

    def flush(self):
        out, self.prev = self.prev, None
        return out


class _StubLogits:
    """bench.py --stub: stand-in for ml3d.engine.KPConvPipeline -- logits [n, 8] with a rank- and step-dependent n."""
    compute = None

    class _Res:
        def __init__(self, t):
            self.t = t

        def wait(self):
            return self.t

    def __init__(self, rank):
        self.rank, self.k, self.prev = rank, 0, None

    @staticmethod
    def logits_of(rank, step):
        g = torch.Generator().manual_seed(1299709 * rank + step)
        return torch.rand((500 + 41 * rank + 13 * step, 8), generator=g)

    def submit(self, pts, lens):
        out, self.prev = self.prev, self._Res(self.logits_of(self.rank, self.k))
        self.k += 1
        return out

    def flush(self):
        out, self.prev = self.prev, None
        return out


def _stub_line(name, units, args, world, dt, checked):
    return {"stub": True, "metric": "NOT A MEASUREMENT: bench.py --stub --workload %s (N > 1 step logic on CPU tensors over gloo)" % name,
            "value": units * args.steps * world / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "scaling": "weak", "gathered_ranks_checked": checked}


def run_pointpillars(args, rank, world, dev, dist):
    from ml3d import dist as mdist
    import os
    stub = bool(getattr(args, "stub", False))
    # sweeps per step: 32 on two threaded lanes (round 5; 16 until then).  One box, threaded lanes: 2 x 16: 1387-1424 frames/s,
    # 2 x 32: 1451 / 1462, 3 x 24: 1410-1479, 3 x 48: 1453, 4 x 32: 1405 (profiles/r05_pp_lanes_sweep.log)
    B = args.frames_per_step or 32
    n_boxes = [0]
    last_det = [None]
    last = [None, 0]            # (what rank 0 received for the last delivered step, number of delivered steps)
    overlap = not getattr(args, "no_overlap", False)
    if stub:
        hosts, pipe = None, _StubDetections(rank, B)
    else:
        import synth_data
        from ml3d import ops
        from ml3d.torch.models.point_pillars import PointPillars
        import synth_weights as W
        cfg = W.POINTPILLARS_KITTI_CFG
        sd = W.pointpillars_state_dict(cfg, 2024)
        m = PointPillars(device=dev, **cfg)
        m.load_state_dict(sd)
        clouds_np = [W.crop_for_cfg(synth_data.kitti_sweep(rank * 100 + i), cfg) for i in range(B)]
        # what a data loader hands over: pinned HOST sweeps; their upload is part of every timed step
        hosts = [torch.from_numpy(c).pin_memory() for c in clouds_np]
        from ml3d.engine import PointPillarsStream
        # ML3D_PP_LANES (A/B knob, default 2): the step's sweeps are dealt to this many independent pipelines (own HIP streams)
        # -- while one lane's convolution drains its last, partly filled round of tiles the other lane's kernels fill the idle CUs
        lanes = max(1, min(B, int(os.environ.get("ML3D_PP_LANES", "2"))))
        # ML3D_PP_THREADED (A/B knob, default 1): one host thread per lane
        pipe = PointPillarsStream(m, dev, lanes=lanes, threaded=os.environ.get("ML3D_PP_THREADED", "1") == "1")

    def deliver(res):
        """a step's detections (host tensors): counted; N > 1: every rank's [n_i, 9] rows -> rank 0 (the ragged gather)"""
        if res is None:
            return
        boxes, scores, labels = res
        last_det[0] = res
        n_boxes[0] = sum(int(b.shape[0]) for b in boxes)
        if world > 1:
            rows = torch.cat([torch.cat([b, s[:, None], l[:, None].to(b.dtype)], 1) for b, s, l in zip(boxes, scores, labels)])
            last[0] = mdist.gather_ragged(rows.reshape(-1).to(dev), dst=0)
        last[1] += 1

    def step():
        # upload (copy stream) -> voxelize / pillar features / backbone / heads -> batched box decode + rotated NMS
        # (Anchor3DHead.get_bboxes, point_pillars.py:945-1025) -> detections back in pinned host memory: the step ends with the
        # boxes, not with the head maps
        if overlap:
            deliver(pipe.submit(hosts))
        else:
            outs = m([h.to(dev, non_blocking=True) for h in hosts])
            deliver(tuple([t.cpu() for t in lst] for lst in m.bbox_head.get_bboxes(*outs)))
    iv = []
    # the roofline kernel, timed IN the timed region: SECOND's second convolution (3x3 64 -> 64, stride 1, 248 x 216) of the
    # first lane's forward of every step -- the launch shape the timed path issues (B / lanes sweeps), with the other lane
    # co-running, bracketed by HIP events on the lane's own compute stream (the stream the kernel is launched on)
    timer = None if stub else _CallTimer(ops, "conv2d_nhwc", 1)
    # the two HBM-bound front-end primitives (a15 voxelize, a16-a17 pillar gather + PFN + canvas scatter), the
    # first lane's launch of every step, bracketed the same way
    t_vox = None if stub else _CallTimer(ops, "voxelize", 0)
    t_pf = None if stub else _CallTimer(ops, "pillar_features", 0)
    timed_step = step
    if timer is not None:
        def timed_step():
            timer.new_step(); t_vox.new_step(); t_pf.new_step()
            step()
    dt = _timed(timed_step, args.steps, args.warmup, world, dist, dev, ev_stream=pipe.compute if overlap else
                (lambda: torch.cuda.current_stream(dev)), intervals=iv)
    deliver(pipe.flush())
    if stub:
        if rank == 0 and world > 1:
            # rank 0 must hold every rank's rows of the last delivered step, each at that rank's own length
            k = last[1] - 1
            assert len(last[0]) == world
            for r in range(world):
                b, sc, lb = _StubDetections.result_of(r, k, B)
                want = torch.cat([torch.cat([x, y[:, None], z[:, None].to(x.dtype)], 1) for x, y, z in zip(b, sc, lb)]).reshape(-1)
                assert torch.equal(last[0][r], want), "rank %d's detections did not arrive intact" % r
        return _stub_line("pointpillars", B, args, world, dt, world) if rank == 0 else None
    torch.cuda.synchronize()
    timer.restore(); t_vox.restore(); t_pf.restore()
    in_region = timer.samples_ms()[args.warmup:]            # (the warm-up steps' launches are not part of the figure)
    shapes = timer.shapes
    vox_in, pf_in = t_vox.samples_ms()[args.warmup:], t_pf.samples_ms()[args.warmup:]
    vox_shapes, pf_shapes = t_vox.shapes, t_pf.shapes
    # the same convolution ALONE on the GPU, as a whole-batch launch and as a lane-shaped one: what the co-running lane costs it
    clouds = [h.to(dev) for h in hosts]
    alone = {}
    prim_alone = {}
    for tag, sub in (("lane", clouds[:max(1, B // lanes)] if overlap else clouds), ("batch", clouds)):
        t2 = _CallTimer(ops, "conv2d_nhwc", 1)
        tv, tp = _CallTimer(ops, "voxelize", 0), _CallTimer(ops, "pillar_features", 0)
        for _ in range(5):
            t2.new_step(); tv.new_step(); tp.new_step()
            m(sub)
            torch.cuda.synchronize()
        t2.restore(); tv.restore(); tp.restore()
        alone[tag] = t2.mean_ms()
        prim_alone[tag] = (tv.mean_ms(), tp.mean_ms(), tv.shapes, tp.shapes)
    # self-check (outside the timed region): the LAST timed step's detections, produced by the pipelined lanes with their kernels
    # overlapping on the GPU, against a quiet single-stream pass over the same sweeps (a kernel that is not stable under co-running
    # kernels shows up here: DESIGN.md 9.10)
    pipe_check = None
    try:
        if overlap and last_det[0] is not None:
            qb, qs, ql = ([t.cpu() for t in lst] for lst in m.bbox_head.get_bboxes(*m(clouds)))
            torch.cuda.synchronize()
            pb, ps, pl = last_det[0]
            same = [len(a) == len(b) and bool(torch.equal(a.cpu(), b)) for a, b in zip(pl, ql)]
            ds = max([float((a.cpu() - b).abs().max()) for a, b, ok in zip(ps, qs, same) if ok and len(b)] or [0.0])
            db = max([float((a.cpu() - b).abs().max()) for a, b, ok in zip(pb, qb, same) if ok and len(b)] or [0.0])
            pipe_check = {"sweeps": len(ql), "sweeps_with_identical_labels": int(sum(same)), "max_score_delta": ds, "max_box_delta": db,
                          "what": "last timed step of the two-lane pipeline against a quiet single-stream forward + decode of the same sweeps"}
    except Exception as e:       # (the check must never cost the bench line)
        pipe_check = {"error": repr(e)[:200]}
    # one sweep at a time, synchronised per sweep: what a detection request sees (upload + voxelize + forward + decode + NMS + D2H)
    lat = []
    for i in range(24):
        t0 = time.perf_counter()
        boxes = m.bbox_head.get_bboxes(*m([hosts[i % B].to(dev, non_blocking=True)]))
        _ = [t.cpu() for t in boxes[0]]
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = lat[4:]
    if rank != 0:
        return None
    (x, w, *_), y = shapes
    Bm, OH, OW, Co = y.shape
    flops = 2.0 * Bm * OH * OW * Co * w.shape[0]
    ms = float(np.mean(in_region))
    # the matrix pipe the convolution ran on.  bf16x3 (default): every float32 product is six bf16 MFMA products (three-way split of
    # both operands, gemm_tile_bf3) -- `achieved` counts the bf16 flops the kernel EXECUTES (6 x the algorithmic ones) against the
    # dense bf16 peak; `f32_equivalent_tflops` is the algorithmic rate, which the f32 MFMA pipe (157.3 TF) could not reach
    bf3 = getattr(timer, "kwargs", {}).get("packed") is not None
    mult, peak = (6.0, PEAK_BF16_TFLOPS) if bf3 else (1.0, PEAK_F32_TFLOPS)
    # the whole forward's algorithmic flops (68.3 GFLOP per KITTI frame through backbone + neck + heads)
    e2e_tflops = 68.3e9 * (B * args.steps * world / dt) / world / 1e12
    out = {"metric": "point-cloud frames/sec (PointPillars KITTI inference: voxelize + pillar features + BEV backbone + heads)",
           "value": B * args.steps * world / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "step_ms_median": float(np.median(iv)),
           "step_ms_p95": float(np.percentile(iv, 95)), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "dtype_note": ("float32 tensors and accumulation; SECOND's 3x3 convolutions multiply on the bf16 matrix pipe with both operands "
                          "split exactly into three bf16 (six MFMA products per float32 product): error against float64 equal to the f32 "
                          "MFMA kernel's (profiles/r05_bf16x3_conv.log), the 1e-4 parity tests run on this path") if bf3 else
                         "float32 end to end (ML3D_PP_CONV=f32)",
           "config": {"workload": "PointPillars KITTI detection, %d synthetic KITTI-shaped sweeps per step per GPU "
                                  "(pointpillars_kitti.yml): host->device upload + voxelize + pillar features + BEV backbone + "
                                  "heads + box decode + rotated NMS" % B, "frames_per_step_per_gpu": B,
                      "h2d_in_timed_region": True, "decode_nms_in_timed_region": True, "d2h_of_detections_in_timed_region": True,
                      "boxes_last_step": n_boxes[0], "streams": 2 * lanes if overlap else 1, "lanes": lanes if overlap else 1,
                      "points_per_sweep": [int(len(c)) for c in clouds_np][:4], "parallelism": "frame-parallel x%d" % world},
           "latency_single_sweep_ms": {"median": float(np.median(lat)), "p95": float(np.percentile(lat, 95)), "sweeps": len(lat)},
           "pipeline_matches_quiet_run": pipe_check,
           "roofline": {"bound": "mfma", "kernel": "%s (SECOND block 0, 3x3 %d->%d on %dx%d)" % (
                            _rocprof_name("pp_conv3x3_64", "gemm_tile_bf3<ConvLoader2, 64>") if bf3 else "gemm_tile2<ConvLoader2, 64, 32, false>",
                            x.shape[3], Co, OH, OW),
                        "matrix_pipe": "bf16 MFMA (v_mfma_f32_32x32x16_bf16), 6 products per float32 product: exact three-way bf16 split of "
                                       "both operands, float accumulation -- float32-equivalent results" if bf3 else "f32 MFMA",
                        "achieved": mult * flops / (ms * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                        "frac": mult * flops / (ms * 1e-3) / 1e12 / peak,
                        "f32_equivalent_tflops": flops / (ms * 1e-3) / 1e12,
                        "traffic": _traffic("pp_conv3x3_64", Bm), "traffic_source": TRAFFIC_SOURCE, "avg_launch_ms": ms, "flops_per_launch": flops,
                        "executed_flops_per_launch": mult * flops,
                        "sweeps_per_launch": int(Bm), "launches_timed": len(in_region),
                        "timed": "inside the timed region, on the lane's compute stream, the other lane co-running",
                        "avg_launch_ms_alone_lane_shape": alone["lane"], "avg_launch_ms_alone_whole_batch": alone["batch"],
                        # (two lanes of 16 sweeps each fill the GPU on their own: in the step they TIME-SHARE it, so the in-step launch
                        #  time is a share of the GPU, not the kernel's efficiency -- that is `frac_alone_lane_shape` / `end_to_end_frac`)
                        "frac_alone_lane_shape": mult * flops / (alone["lane"] * 1e-3) / 1e12 / peak,
                        "f32_equivalent_tflops_alone_lane_shape": flops / (alone["lane"] * 1e-3) / 1e12,
                        "frac_alone_whole_batch": mult * 2.0 * B * OH * OW * Co * w.shape[0] / (alone["batch"] * 1e-3) / 1e12 / peak,
                        "end_to_end_tflops": e2e_tflops, "end_to_end_frac": e2e_tflops / PEAK_F32_TFLOPS,
                        "end_to_end_note": "68.3 GFLOP per frame (SURVEY.md §8d) x frames/s per GPU: every kernel of the step, "
                                           "H2D, voxelize, decode and NMS included"}}
    # ---- the HBM-bound front end as roofline objects ------------------------------------------
    def vox_bytes(shapes):
        (pts3, *_), v = shapes
        n, mv, kk = int(pts3.shape[0]), int(v.voxel_coords.shape[0]), int(v.voxel_point_indices.shape[0])
        # xyz read + (x, y, z) int32 voxel coordinates + int64 row splits + int64 point indices written: what the op's boundary
        # moves.  (§8d's figure also counts a dense [M, 32, 4] gather -- 512 B per pillar -- that this design never writes: the
        # pillar kernel gathers straight from the points)
        return 12.0 * n + 12.0 * mv + 8.0 * (mv + 1) + 8.0 * kk, n, mv, kk

    def pf_bytes(shapes):
        (pts4, v, *_), canvas = shapes
        n, mv = int(pts4.shape[0]), int(v.voxel_coords.shape[0])
        return 16.0 * n + 256.0 * mv + 4.0 * canvas.numel(), n, mv     # §8d: canvas write (54.9 MB per sweep) + 256 B per pillar
    lane_units = max(1, B // lanes) if overlap else B
    vb, vn, vm, vk = vox_bytes(vox_shapes)
    pb, pn, pm = pf_bytes(pf_shapes)
    va, pa = prim_alone["lane" if overlap else "batch"][:2]
    other = [
        _hbm_entry("a15 voxelize (point_pillars.py:328-382)", "vox_keys32 + 3 x fs_pass + vox_group32 + vox_fill32 (fused stable radix voxelize: 8 launches, count + fill)",
                   vb, float(np.mean(vox_in)) if vox_in else None, va, "pp_voxelize", lane_units,
                   "%d in-range points -> %d pillars (%d kept points) of %d sweeps in one call; the interval holds the op's one host "
                   "read-back (pillar count) between count and fill" % (vn, vm, vk, lane_units), len(vox_in)),
        _hbm_entry("a16-a17 pillar gather + PFN + canvas scatter (point_pillars.py:512-616)",
                   "pillar_pfn_v4 + canvas fill (grid_zero)", pb, float(np.mean(pf_in)) if pf_in else None, pa,
                   "pp_pillar_features", lane_units,
                   "%d pillars of %d sweeps -> NHWC canvas %s: zero fill + one 256-byte row per pillar" %
                   (pm, lane_units, "x".join(str(int(d)) for d in pf_shapes[1].shape)), len(pf_in)),
    ]
    out["roofline_other"] = other
    if not args.no_cpu_baseline and world == 1:
        from oracle import pointpillars_ref as P          # the checker, used here only as the timed CPU baseline
        pts = [torch.from_numpy(c) for c in clouds_np[:1]]
        P.forward(sd, cfg, pts)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 15 and n < 8:
            P.forward(sd, cfg, pts)
            n += 1
        out["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "frames/s",
                               "cores": int(torch.get_num_threads()), "kind": "port",
                               "sample": "%d passes over 1 sweep, oracle voxelize + PyTorch-CPU forward restating the reference" % n}
    else:
        out["cpu_baseline"] = None
    return out


def run_kpconv(args, rank, world, dev, dist):
    from ml3d import dist as mdist
    import os
    stub = bool(getattr(args, "stub", False))
    # spheres per step: the batch build is launch/latency-bound (~550 small launches + 9 host read-backs per batch whatever
    # its size), so throughput follows the batch (round 2, pipelined: 16 -> 3219, 32 -> 4400, 48 -> 4748, 63 -> 5002 spheres/s)
    # 96 spheres per step by default (round 5; 64 until then).  Same box: 8257 / 8268 spheres/s at 64, 8708 at 96, 8609 / 8526 at 128
    # (profiles/r05_batch_sweep.log) -- the build chain's per-launch latencies are amortised over more rows
    B = args.frames_per_step or 96
    overlap = not getattr(args, "no_overlap", False)
    last = [None, 0]
    builders = fwd_streams = 1
    if stub:
        host_pts, lens, pipe = torch.zeros((4, 3)), [4], _StubLogits(rank)
    else:
        import synth_data
        from ml3d import ops
        from ml3d.torch.models.kpconv import KPFCNN, KPConvBatch
        import synth_weights as W
        cfg = dict(W.TORONTO3D_CFG)
        sd = W.kpconv_state_dict(cfg, 2024)
        m = KPFCNN(**cfg, device=dev)
        m.load_state_dict(sd)
        spheres = [synth_data.toronto3d_sphere(rank * 100 + i) for i in range(B)]
        lens = [len(s) for s in spheres]
        host_pts = torch.from_numpy(np.concatenate(spheres)).pin_memory()     # the stacked spheres of a step arrive from the HOST
        pts = host_pts.to(dev)
        np.random.seed(0)
        from ml3d.engine import KPConvPipeline, KPConvPipelineN
        # ML3D_KP_BUILDERS (A/B knob): batch builds in flight, each on its own stream and host thread (1 = the two-stream
        # pipeline of rounds 2-4: one build under one forward)
        builders = max(1, int(os.environ.get("ML3D_KP_BUILDERS", "2")))
        # ML3D_KP_FORWARD_STREAMS (A/B knob, default 2): consecutive batches' forwards alternate between two compute streams -- the deep
        # layers' small kernels of one batch run under the large ones of the next: 9076 / 9034 / 9012 against 8587 / 8560 / 8601 spheres/s
        # alternating on one box (profiles/r05_kp_forward_streams_ab.log); three builders + two streams: 8563
        fwd_streams = max(1, int(os.environ.get("ML3D_KP_FORWARD_STREAMS", "2")))
        pipe = KPConvPipelineN(m, cfg, dev, builders=builders, forward_streams=fwd_streams) if (builders > 1 or fwd_streams > 1) \
            else KPConvPipeline(m, cfg, dev)

    def finish(res):
        if res is not None and world > 1:       # (every rank's batch has its own point count: the ragged gather)
            last[0] = mdist.gather_ragged(torch.argmax(res.wait(), 1).to(torch.uint8), dst=0)
        if res is not None:
            last[1] += 1

    def step():
        pts = host_pts.to(dev, non_blocking=True)              # H2D inside the timed step
        if overlap:
            # the batch build of this step (9 host read-backs) on one stream under the forward of the previous step on another:
            # every timed step = one build + one forward, as in the sequential loop
            finish(pipe.submit(pts, lens))
        else:
            batch = KPConvBatch(pts, lens, cfg, device=dev)
            logits = m(batch)
            if world > 1:
                mdist.gather_ragged(torch.argmax(logits, 1).to(torch.uint8), dst=0)
    iv = []
    # the roofline op, timed IN the timed region: the first resnet block's KPConv (32 -> 32 on the full-resolution layer) of every
    # step's forward, bracketed by HIP events on the stream it is launched on (the pipeline's compute stream), the next batch's
    # build co-running on the other stream
    timer = None if stub else _CallTimer(ops, "kpconv_rigid", 1)
    # the HBM-bound primitives of the batch build (a10 fixed-radius search, a11 grid subsample): the full-resolution
    # layer's conv search (grid build + gather, then expand) and pooling subsample (count, then fill) of every step's build,
    # bracketed by HIP events the one-call build records on ITS stream (ML3DKpBatchDesc.trace_events), the previous step's
    # forward co-running
    from ml3d.ops import search as _search
    prim_ev = []

    def new_trace():
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        for e in evs:
            e.record()                   # materialises the hipEvent_t handles
        _search.KPBATCH_TRACE.append(evs)
        return evs
    timed_step = step
    if timer is not None:
        def timed_step():
            timer.new_step()
            prim_ev.append(new_trace())
            step()
    dt = _timed(timed_step, args.steps, args.warmup, world, dist, dev, ev_stream=(lambda: pipe.compute) if overlap else
                (lambda: torch.cuda.current_stream(dev)), intervals=iv)
    rest = pipe.flush()
    for r in (rest if isinstance(rest, list) else [rest]):
        finish(r)
    if stub:
        if rank == 0 and world > 1:
            k = last[1] - 1
            assert len(last[0]) == world
            for r in range(world):
                want = torch.argmax(_StubLogits.logits_of(r, k), 1).to(torch.uint8)
                assert torch.equal(last[0][r], want), "rank %d's labels did not arrive intact" % r
        return _stub_line("kpconv", B, args, world, dt, world) if rank == 0 else None
    torch.cuda.synchronize()
    timer.restore()
    # self-check (outside the timed region): the LAST pipelined batch -- built on its own stream under other batches' forwards -- and its
    # logits against a quiet rebuild with the same grid rotations + a quiet forward
    pipe_check = None
    try:
        lastres = (rest if isinstance(rest, list) else [rest])[-1] if overlap and rest else None
        if lastres is not None:
            pb = lastres.batch
            qbatch = KPConvBatch(host_pts.to(dev), lens, cfg, rotations=pb.rotations, device=dev)
            qlogits = m(qbatch)
            torch.cuda.synchronize()
            same = all(len(a) == len(b) and all(x.shape == y.shape and bool(torch.equal(x, y)) for x, y in zip(a, b))
                       for a, b in ((pb.points, qbatch.points), (pb.neighbors, qbatch.neighbors), (pb.pools, qbatch.pools),
                                    (pb.upsamples, qbatch.upsamples)))
            pl = lastres.logits
            pipe_check = {"batch_matrices_identical": bool(same),
                          "logits_max_delta": float((pl - qlogits).abs().max()) if pl.shape == qlogits.shape else None,
                          "labels_identical": bool(pl.shape == qlogits.shape and torch.equal(pl.argmax(1), qlogits.argmax(1))),
                          "what": "last pipelined batch (build + forward under co-running streams) against a quiet rebuild with the same grid rotations + forward"}
    except Exception as e:       # (the check must never cost the bench line)
        pipe_check = {"error": repr(e)[:200]}
    in_region = timer.samples_ms()[max(0, args.warmup - 1):]     # (the pipeline runs a step's forward one submit later)
    shapes = timer.shapes
    del _search.KPBATCH_TRACE[:]

    def prim_ms(evs):
        """(conv search ms, subsample ms) of one build from its 8 events: two intervals each (the size read-back between a
        search's gather and its expand, between the subsample's count and its fill, is not in the figure)"""
        try:
            return (evs[0].elapsed_time(evs[1]) + evs[2].elapsed_time(evs[3]), evs[4].elapsed_time(evs[5]) + evs[6].elapsed_time(evs[7]))
        except Exception:
            return None
    prim_in = [x for x in (prim_ms(e) for e in prim_ev[args.warmup:]) if x]
    # the same op with nothing else on the GPU: five SEQUENTIAL steps after the timed region
    m(KPConvBatch(pts, lens, cfg, device=dev))           # (untimed: the caller-stream allocator pool is cold after a pipelined run)
    torch.cuda.synchronize()
    t2 = _CallTimer(ops, "kpconv_rigid", 1)
    alone_ev = []
    for _ in range(5):
        t2.new_step()
        alone_ev.append(new_trace())
        last_batch = KPConvBatch(pts, lens, cfg, device=dev)
        m(last_batch)
        torch.cuda.synchronize()
    t2.restore()
    del _search.KPBATCH_TRACE[:]
    prim_alone = [x for x in (prim_ms(e) for e in alone_ev) if x]
    # one sphere at a time, synchronised per sphere: upload + batch build (9 read-backs) + forward + arg-max back on the host
    lat = []
    for i in range(24):
        sp = torch.from_numpy(spheres[i % B]).pin_memory()
        t0 = time.perf_counter()
        lab = torch.argmax(m(KPConvBatch(sp.to(dev, non_blocking=True), [len(sp)], cfg, device=dev)), 1).to(torch.uint8).cpu()
        torch.cuda.synchronize()
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = lat[4:]
    if rank != 0:
        return None
    (q, s, inds, x, kp, w, *_), y = shapes
    nq, H, cin, cout = q.shape[0], inds.shape[1], x.shape[1], y.shape[1]
    # EXECUTED flops: the dense index matrix is padded with shadow entries up to the longest row of the batch (H columns);
    # the aggregation kernel stops at a row's last real neighbour, so a row costs its REAL length (sum over the rows below),
    # not H.  `frac` prices that; `frac_reference_formulation` the reference's dense [N, H] formulation (kpconv.py:1105-1118).
    real = int((inds < s.shape[0]).sum().item())
    flops_exec = 2.0 * 15 * real * cin + nq * 2.0 * 15 * cin * cout
    flops_dense = nq * (2.0 * 15 * H * cin + 2.0 * 15 * cin * cout)
    ms = float(np.mean(in_region))
    flops = flops_exec
    out = {"metric": "point-cloud spheres/sec (KPConv rigid Toronto3D inference: GPU batch build + forward)",
           "value": B * args.steps * world / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "step_ms_median": float(np.median(iv)),
           "step_ms_p95": float(np.percentile(iv, 95)), "step_ms_max": float(np.max(iv)),
           "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "KPConv (rigid) Toronto3D inference, %d synthetic 10000-point input spheres per step per "
                                  "GPU (kpconv_toronto3d.yml): radius search + grid subsample batch build, then forward%s" % (B, (" (%d one-call batch builds in flight on their own HIP streams / host threads under the forwards on another)" % builders if builders > 1 else " (build of step i+1 overlapped with the forward of step i on two HIP streams)") if overlap else ""),
                      "frames_per_step_per_gpu": B, "points_per_step": int(sum(lens)), "h2d_in_timed_region": True,
                      "builds_in_flight": builders if overlap else 1, "forward_streams": fwd_streams if overlap else 1,
                      "parallelism": "frame-parallel x%d" % world},
           "latency_single_sphere_ms": {"median": float(np.median(lat)), "p95": float(np.percentile(lat, 95)), "spheres": len(lat)},
           "pipeline_matches_quiet_run": pipe_check,
           "roofline": {"bound": "mfma", "kernel": "kp_agg_gemm32 (KPConv %d->%d, %d queries x %d neighbour columns: MFMA aggregation + the [480 x 32] product in one kernel)" % (cin, cout, nq, H),
                        "achieved": flops / (ms * 1e-3) / 1e12, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                        "frac": flops / (ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                        "frac_reference_formulation": flops_dense / (ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                        "traffic": _traffic("kpconv_block_32_32", B), "traffic_source": TRAFFIC_SOURCE, "avg_launch_ms": ms,
                        "executed_flops_per_launch": flops_exec, "reference_flops_per_launch": flops_dense,
                        "real_neighbours_per_query": real / float(nq), "launches_timed": len(in_region),
                        "timed": "inside the timed region, on the pipeline's compute stream, the next batch's build co-running",
                        "avg_launch_ms_alone": t2.mean_ms(),
                        "frac_alone": flops_exec / (t2.mean_ms() * 1e-3) / 1e12 / PEAK_F32_TFLOPS,
                        # the op's ALGORITHMIC HBM bytes: index matrix + positions + feature rows in, output rows out
                        "algorithmic_bytes_per_launch": 4.0 * nq * H + 12.0 * (nq + s.shape[0]) + 4.0 * cin * s.shape[0] + 4.0 * cout * nq,
                        # with two forward streams (and two builds) in flight an in-step launch time is a SHARE of the GPU, not the
                        # kernel's efficiency: the whole step on the reference's formulation is the figure that does not depend on overlap
                        "end_to_end_tflops": 11.5e9 * (B * args.steps * world / dt) / world / 1e12,
                        "end_to_end_frac": 11.5e9 * (B * args.steps * world / dt) / world / 1e12 / PEAK_F32_TFLOPS,
                        "end_to_end_note": "11.5 GFLOP per sphere (SURVEY.md §8d: KPConv aggregation + products + unary / decoder Linears, "
                                           "reference formulation) x spheres/s per GPU: every kernel of the step, H2D and the batch build included"}}
    # ---- the HBM-bound primitives of the build as roofline objects ------------------------------
    rnq = rns = int(last_batch.points[0].shape[0])
    rH = int(last_batch.neighbors[0].shape[1])
    rad_bytes = 12.0 * (rnq + rns) + 4.0 * rnq * rH                    # §8d: xyz of queries + supports read, dense int32 rows written
    sn, sm = rnq, int(last_batch.points[1].shape[0])
    sub_bytes = 12.0 * sn + 12.0 * sm                                  # §8d: points read, barycentres written
    col = lambda rows, i: float(np.mean([r[i] for r in rows])) if rows else None
    out["build"] = {"library_calls": 1, "host_syncs_per_batch": int(getattr(last_batch, "host_syncs", -1)),
                    "note": "ml3d_kpconv_batch_build: the 5-layer chain (13 radius searches, 4 grid subsamplings) enqueued from C++, "
                            "one blocking size read-back per layer"}
    out["roofline_other"] = [
        _hbm_entry("a10 fixed-radius search -> dense rows (kpconv.py:2002-2034)", "grid build + radius_gather + radius_expand",
                   rad_bytes, col(prim_in, 0), col(prim_alone, 0), "kp_radius_dense", B,
                   "layer-0 conv search of the batch: %d queries = supports, r = %.3g m -> int32 [%d, %d] padded with the shadow "
                   "index; two event intervals summed (grid build + gather, then expand): the host read-back of the longest row "
                   "between them is not in the figure" %
                   (rnq, cfg['first_subsampling_dl'] * cfg['conv_radius'], rnq, rH), len(prim_in)),
        _hbm_entry("a11 batch grid subsample (kpconv.py:2037-2164)", "rotate_rows_k + sub_items_k<count> + sub_items_k<fill> (one workgroup per sphere, grouping in LDS; round 5: 64-bit radix sort, ~30 launches)",
                   sub_bytes, col(prim_in, 1), col(prim_alone, 1), "kp_subsample", B,
                   "layer-0 pooling grid of the batch: %d points -> %d barycentres at dl = %.3g m on randomly oriented grids; two "
                   "event intervals summed (rotation + count, fill + rotation back): the read-back of the pooled size between them "
                   "is not in the figure" % (sn, sm, 2 * cfg['first_subsampling_dl']), len(prim_in)),
    ]
    if not args.no_cpu_baseline and world == 1:
        from oracle import kpconv_ref as K                # the checker, used here only as the timed CPU baseline
        sp = spheres[0]
        feats = torch.ones((len(sp), 1))
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 15 and n < 8:
            seg = K.segmentation_inputs(sp, [len(sp)], cfg)
            K.forward(sd, cfg, K.to_torch_batch(seg), feats)
            n += 1
        out["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "frames/s",
                               "cores": int(torch.get_num_threads()), "kind": "port",
                               "sample": "%d passes over 1 sphere, oracle radius search / subsample (OpenMP) + PyTorch-CPU "
                                         "forward restating the reference" % n}
    else:
        out["cpu_baseline"] = None
    return out


def run_randlanet_train(args, rank, world, dev, dist):
    """one DATA-PARALLEL training step of RandLA-Net per timed step (forward + weighted cross entropy + backward on the
    HIP training kernels of csrc/train.hip + SGD update), ``torch.nn.parallel.DistributedDataParallel`` around the native model class:
    at N > 1 its bucketed gradient all-reduce runs over RCCL (backend nccl) and overlaps the backward.  The reference refuses this
    for semantic segmentation (ml3d/torch/pipelines/base_pipeline.py:44-47) and wraps only its detection models
    (object_detection.py:340).  After the timed region every rank's gradient checksum is gathered: all ranks must hold the SAME
    averaged gradients.  ``--stub``: the same code on CPU tensors over gloo with the emulated library is tests/test_ddp_training_emulated.py."""
    import time
    import synth_data
    import synth_weights as W
    from ml3d.torch.models import RandLANet
    cfg = dict(W.RANDLANET_SEMANTICKITTI_CFG)
    B = args.frames_per_step or 4                                   # randlanet_semantickitti.yml: batch_size 4
    N = cfg["num_points"]
    torch.manual_seed(0)
    model = RandLANet(**cfg, device=dev)
    model.load_state_dict(W.randlanet_state_dict(cfg, 2024))
    model.to(dev)
    model.train()
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], bucket_cap_mb=8)
    opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
    loss_obj = type("L", (), {"weighted_CrossEntropyLoss": torch.nn.CrossEntropyLoss()})()
    pts = torch.from_numpy(np.stack([synth_data.semantickitti_patch(1000 * rank + i, N) for i in range(B)])).to(dev)
    labels = torch.randint(1, cfg["num_classes"], (B, N), generator=torch.Generator().manual_seed(rank))
    nbr, itp = model.neighbor_pyramid(pts)
    inputs = {"coords": [pts], "features": pts.clone(), "neighbor_indices": nbr, "interp_idx": itp}

    def step():
        opt.zero_grad(set_to_none=True)
        logits = net(inputs)
        loss, _, _ = model.get_loss(loss_obj, logits, {"data": {"labels": labels}}, dev)
        loss.backward()
        opt.step()
        return loss

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # one more backward WITHOUT the optimiser step: the averaged gradients every rank holds must be identical
    opt.zero_grad(set_to_none=True)
    logits = net(inputs)
    l2, _, _ = model.get_loss(loss_obj, logits, {"data": {"labels": labels}}, dev)
    l2.backward()
    g = torch.cat([p.grad.reshape(-1).double() for p in model.parameters() if p.grad is not None])
    chk = torch.stack([g.sum(), g.abs().sum(), (g * torch.arange(1, g.numel() + 1, device=dev, dtype=torch.float64)).sum()])
    same = None
    if world > 1:
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        same = all(bool(torch.equal(allc[0], c)) for c in allc)
    return {"metric": "RandLA-Net SemanticKITTI TRAINING steps/sec (DDP: forward + loss + backward on HIP kernels + RCCL gradient all-reduce + SGD)",
            "value": args.steps * world * B / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "loss": float(loss),
            "config": {"workload": "RandLA-Net SemanticKITTI training step, %d x %d points per GPU (randlanet_semantickitti.yml batch_size 4)" % (B, N),
                       "parallelism": "ddp x%d" % world, "train_ops": os.environ.get("ML3D_TRAIN_OPS", "hip")},
            "ddp_gradients_identical_on_all_ranks": same, "grad_checksum": [float(v) for v in chk.tolist()]}
