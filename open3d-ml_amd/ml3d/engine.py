"""Preallocated RandLA-Net inference engine (one process per GPU).

Everything a step needs lives in HBM before the step starts: packed BatchNorm-folded weights,
the int32 neighbour pyramid, the forward workspace and the score tensor.  A step is two C-ABI
calls on torch's current stream — ``ml3d_randla_knn_pyramid`` (replaces the 8 CPU knn_search
calls of RandLANet.transform, reference randlanet.py:218-229) and ``ml3d_randla_forward``
(replaces RandLANet.forward, randlanet.py:241-298) — with no host synchronisation.
"""
import ctypes as C
import os

import torch

from . import _abi
from .ops import pyramid_sizes
from .torch.models import _randla_pack


class RandLAInferenceEngine:

    def __init__(self, cfg, state_dict, batch, num_points, device, tile_order=None):
        """``tile_order``: walk the attention tiles in the cell-sorted order of the neighbour pyramid's grids (same
        scores bit for bit, better locality of the neighbour gathers).  False / 0 = off, True = every level, an int n =
        the n finest levels only.  Default: the two finest levels (measured at 64 frames of 45 056 points, two runs each:
        5290 / 5273 frames/s off, 5483 / 5483 every level, 5563 / 5555 two levels -- the gather sets of the coarse levels
        fit the L2 anyway and their kernels lose 7 % to the indirection).  ``ML3D_TILE_ORDER`` overrides with the SAME meaning
        as the argument: 0 = off, n = the n finest levels, ``all`` = every level."""
        self.lib = _abi.get()
        if tile_order is None:
            env = (os.environ.get("ML3D_TILE_ORDER", "2") or "0").strip().lower()
            tile_order = True if env == "all" else int(env)
        self.tile_levels = (int(cfg["num_layers"]) if tile_order is True else int(tile_order or 0))
        self.tile_order = self.tile_levels > 0
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("RandLAInferenceEngine needs an MI355X device; there is no CPU fallback")
        self.B, self.N = int(batch), int(num_points)
        self.L = int(cfg["num_layers"])
        self.K = int(cfg["num_neighbors"])
        self.desc = _abi.make_desc(cfg, self.B, self.N)
        off = _abi.randla_param_offsets(self.lib, self.desc)
        self.params = torch.from_numpy(_randla_pack.pack(state_dict, cfg, off)).to(self.device)
        self.n = pyramid_sizes(self.N, cfg["sub_sampling_ratio"])
        dev = self.device
        self.nbr = [torch.empty((self.B, self.n[l], self.K), dtype=torch.int32, device=dev) for l in range(self.L)]
        self.itp = [torch.empty((self.B, self.n[l], 1), dtype=torch.int32, device=dev) for l in range(self.L)]
        self.scores = torch.empty((self.B, self.N, int(cfg["num_classes"])), dtype=torch.float32, device=dev)
        self.ratios = (C.c_int32 * self.L)(*[int(r) for r in cfg["sub_sampling_ratio"]])
        self.pyr_ws_bytes = self.lib.ml3d_randla_pyramid_workspace_bytes(self.B, self.N, self.L, self.ratios)
        self.fwd_ws_bytes = self.lib.ml3d_randla_forward_workspace_bytes(C.byref(self.desc))
        if self.pyr_ws_bytes == 0 or self.fwd_ws_bytes == 0:
            raise RuntimeError("RandLAInferenceEngine: invalid model/pyramid description")
        self.pyr_ws = torch.empty(self.pyr_ws_bytes, dtype=torch.uint8, device=dev)
        self.fwd_ws = torch.empty(self.fwd_ws_bytes, dtype=torch.uint8, device=dev)
        self._t_n = _abi.ptr_table([t.data_ptr() for t in self.nbr])
        self._t_i = _abi.ptr_table([t.data_ptr() for t in self.itp])
        self.order = [torch.empty(self.B * self.n[l], dtype=torch.int32, device=dev) if l < self.tile_levels else None
                      for l in range(self.L)] if self.tile_order else None
        self._t_o = _abi.ptr_table([0 if t is None else t.data_ptr() for t in self.order]) if self.tile_order else None

    def _check(self, points, features):
        if tuple(points.shape) != (self.B, self.N, 3) or points.dtype != torch.float32 or \
                not points.is_cuda or not points.is_contiguous():
            raise RuntimeError("engine: points must be a contiguous float32 CUDA tensor [%d, %d, 3]" % (self.B, self.N))
        if tuple(features.shape) != (self.B, self.N, int(self.cfg["in_channels"])) or \
                features.dtype != torch.float32 or not features.is_cuda or not features.is_contiguous():
            raise RuntimeError("engine: features must be a contiguous float32 CUDA tensor [B, N, in_channels]")

    def neighbors(self, points, trace=None):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = self.lib.ml3d_randla_knn_pyramid_ordered(points.data_ptr(), self.B, self.N, self.L, self.ratios, self.K,
                                                      self._t_n, self._t_i, self._t_o, self.pyr_ws.data_ptr(),
                                                      self.pyr_ws_bytes, st, trace)
        _abi.check(rc, "ml3d_randla_knn_pyramid")
        return self.nbr, self.itp

    def forward(self, points, features, trace=None):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = self.lib.ml3d_randla_forward_ordered(C.byref(self.desc), self.params.data_ptr(), features.data_ptr(),
                                                  points.data_ptr(), self._t_n, self._t_i, self._t_o,
                                                  self.scores.data_ptr(), self.fwd_ws.data_ptr(), self.fwd_ws_bytes, st,
                                                  trace)
        _abi.check(rc, "ml3d_randla_forward")
        return self.scores

    def step(self, points, features, knn_trace=None, fwd_trace=None):
        """One hot-path pass over a batch of frames: neighbour pyramid + forward -> scores [B, N, classes]."""
        self._check(points, features)
        with torch.cuda.device(self.device):
            self.neighbors(points, knn_trace)
            return self.forward(points, features, fwd_trace)


class PipelinedRandLAEngine:
    """Two engines in ping-pong on two HIP streams: the neighbour pyramid of batch i+1 (VALU-bound grid search)
    runs on the search stream while the forward of batch i (MFMA / LDS-bound) runs on the compute stream, so the
    two kinds of kernels share the CUs.  Every batch still gets its own pyramid and its own forward; only the
    ORDER of independent work changes.  ``submit`` returns the scores tensor of that batch, valid once the
    compute stream (``self.compute``) has been synchronised or waited for."""

    def __init__(self, cfg, state_dict, batch, num_points, device):
        self.eng = [RandLAInferenceEngine(cfg, state_dict, batch, num_points, device) for _ in range(2)]
        self.eng[1].params = self.eng[0].params            # one weight replica
        self.device = self.eng[0].device
        with torch.cuda.device(self.device):
            self.search = torch.cuda.Stream()             # (HIP stream priorities were measured in round 3: no effect)
            self.compute = torch.cuda.Stream()
            self.knn_done = [torch.cuda.Event(), torch.cuda.Event()]
            self.fwd_done = [torch.cuda.Event(), torch.cuda.Event()]
            # WHEN the next batch's search may start, relative to the running forward.  The search launch is VALU-bound
            # (profiles/r03_pmc_knn_sq.csv: 0.7-0.8 of the VALU issue cycles): whatever shares its SIMDs gets the leftover
            # issue slots.  Left to itself it lands on the forward's first layer, whose per-point chain is a short,
            # bandwidth-bound persistent kernel that then runs 6x longer (0.16 -> 1.0 ms, r03 kernel tables).  The gate is an
            # event the LIBRARY records inside the forward at the start of the kernel with this trace tag (8 * layer + stage,
            # ml3d_hip.h); the search stream waits for the previous forward's gate.  ML3D_SEARCH_GATE: tag, -1 = no gate.
            self.gate_tag = int(os.environ.get("ML3D_SEARCH_GATE", "9"))
            self.gate = [torch.cuda.Event(), torch.cuda.Event()]
            for g in self.gate:
                g.record()                        # materialise the hipEvent_t -- on the CALLER's stream: the first use of a
                                                  # stream decides its hardware queue (see bench.py, GPU_MAX_HW_QUEUES)
        self.gate_rec = [None, None]
        self.i = 0
        self.n = self.eng[0].n

    def submit(self, points, features, knn_trace=None, fwd_trace=None, ready=None):
        """``ready``: event after which ``points`` / ``features`` are valid (e.g. recorded on a copy stream); default:
        everything already enqueued on the caller's current stream."""
        e = self.eng[self.i & 1]
        slot = self.i & 1
        self.i += 1
        e._check(points, features)
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            # everything the caller has enqueued comes first: its inputs, and its READS of the scores buffer this engine
            # slot is about to overwrite (bench.py's argmax of two steps ago runs on the caller's stream)
            self.search.wait_stream(cur)
            self.compute.wait_stream(cur)
            if ready is not None:
                self.search.wait_event(ready)
                self.compute.wait_event(ready)
            points.record_stream(self.search)     # ... and must outlive the side-stream kernels that read them
            points.record_stream(self.compute)
            features.record_stream(self.compute)
            with torch.cuda.stream(self.search):
                self.search.wait_event(self.fwd_done[slot])      # the forward that last read these index buffers
                if self.gate_tag >= 0 and self.i > 1:
                    self.search.wait_event(self.gate[slot ^ 1])  # ... and the gate inside the forward enqueued just before
                e.neighbors(points, knn_trace)
                self.knn_done[slot].record(self.search)
            with torch.cuda.stream(self.compute):
                self.compute.wait_event(self.knn_done[slot])
                tr = fwd_trace
                if self.gate_tag >= 0:
                    g = _abi.Trace()
                    g.tag, g.ev_start, g.ev_stop = self.gate_tag, C.c_void_p(self.gate[slot].cuda_event), None
                    if fwd_trace is not None:
                        g.next = C.pointer(fwd_trace)
                    self.gate_rec[slot] = (g, fwd_trace)         # (keep the host records alive through the call)
                    tr = g
                out = e.forward(points, features, tr)
                self.fwd_done[slot].record(self.compute)
        return out

    def synchronize(self):
        self.search.synchronize()
        self.compute.synchronize()


class RandLAFrameStream:
    """Host batches in, scores out: what ``bench.py`` times.  Three HIP streams: the pinned host batch of step i + 1 is
    uploaded on the copy stream (two device buffers) while the neighbour pyramid of step i + 1 runs on the search stream
    and the forward of step i on the compute stream (``PipelinedRandLAEngine``); ``overlap=False`` keeps everything on
    the caller's stream in program order.  ``in_channels == 3``: the features ARE the coordinates
    (randlanet.py:208-209), one upload serves both."""

    def __init__(self, cfg, state_dict, batch, num_points, device, overlap=True):
        self.device = torch.device(device)
        self.overlap = bool(overlap)
        self.cin = int(cfg["in_channels"])
        self.B, self.N = int(batch), int(num_points)
        self.engine = PipelinedRandLAEngine(cfg, state_dict, batch, num_points, device) if overlap else \
            RandLAInferenceEngine(cfg, state_dict, batch, num_points, device)
        self.n = self.engine.n
        with torch.cuda.device(self.device):
            self.h2d = torch.cuda.Stream() if overlap else None
            self.pts = [torch.empty((self.B, self.N, 3), dtype=torch.float32, device=self.device) for _ in range(2)]
            self.feat = self.pts if self.cin == 3 else \
                [torch.empty((self.B, self.N, self.cin), dtype=torch.float32, device=self.device) for _ in range(2)]
            self.uploaded = [torch.cuda.Event(), torch.cuda.Event()]
            self.consumed = [torch.cuda.Event(), torch.cuda.Event()]
        self.k = 0

    @property
    def compute_stream(self):
        return self.engine.compute if self.overlap else torch.cuda.current_stream(self.device)

    def single_engine(self):
        return self.engine.eng[0] if self.overlap else self.engine

    def submit(self, host_points, host_features=None, knn_trace=None, fwd_trace=None, done=None):
        """host tensors (pinned for a truly asynchronous copy) -> the scores tensor of this step, valid once
        ``compute_stream`` has reached ``done`` (an event recorded there when given) / has been synchronised."""
        slot = self.k & 1
        self.k += 1
        if self.cin != 3 and host_features is None:
            raise RuntimeError("RandLAFrameStream: in_channels = %d needs host_features" % self.cin)
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            up = self.h2d if self.overlap else cur
            with torch.cuda.stream(up):
                up.wait_event(self.consumed[slot])       # the step that last read this buffer pair has finished
                self.pts[slot].copy_(host_points, non_blocking=True)
                if self.cin != 3:
                    self.feat[slot].copy_(host_features, non_blocking=True)
                self.uploaded[slot].record(up)
            if self.overlap:
                out = self.engine.submit(self.pts[slot], self.feat[slot], knn_trace, fwd_trace, ready=self.uploaded[slot])
                self.consumed[slot].record(self.engine.compute)
                if done is not None:
                    done.record(self.engine.compute)
            else:
                out = self.engine.step(self.pts[slot], self.feat[slot], knn_trace, fwd_trace)
                self.consumed[slot].record(cur)
                if done is not None:
                    done.record(cur)
        return out

    def synchronize(self):
        if self.overlap:
            self.h2d.synchronize()
            self.engine.synchronize()
        else:
            torch.cuda.current_stream(self.device).synchronize()


def make_trace(tag, ev_start, ev_stop):
    """Trace record from two ``torch.cuda.Event(enable_timing=True)`` objects (must be created on
    the current device; ``.record()`` once beforehand materialises the underlying hipEvent_t)."""
    t = _abi.Trace()
    t.tag = int(tag)
    t.ev_start = C.c_void_p(ev_start.cuda_event)
    t.ev_stop = C.c_void_p(ev_stop.cuda_event)
    return t


class KPConvPipeline:
    """Batch build of step i + 1 under the forward of step i (KPConv segmentation inference, frame-parallel rows a10-a14).

    The batch build of ``KPConvBatch`` (13 radius searches + 4 grid subsamplings per 5-layer batch) needs ~9 host read-backs of
    result sizes; run back to back with the forward on one stream, every read-back drains the GPU.  Here the build runs on its
    own HIP stream and the forward of the PREVIOUS batch is enqueued on a second stream first, so the read-backs of the build
    only stall the host while the forward keeps the GPU busy.  ``submit`` returns the result of the previous batch (or None);
    ``flush`` runs the forward of the last one.  Results are bit-identical to ``model(KPConvBatch(...))`` run in sequence."""

    class Result:
        def __init__(self, logits, done, batch):
            self.logits, self.done, self.batch = logits, done, batch

        def wait(self, stream=None):
            """make ``stream`` (default: the caller's current stream) wait for the logits; returns them"""
            s = stream or torch.cuda.current_stream(self.logits.device)
            s.wait_event(self.done)
            self.logits.record_stream(s)
            return self.logits

    def __init__(self, model, cfg, device, threaded=False):
        """``threaded``: the forward of the previous batch is ENQUEUED by a worker thread while the caller's thread builds the
        next batch (the ~1.5 ms of Python that enqueue a forward leave the build's chain of ~100 library calls and 9 blocking
        read-backs).  Measured: no gain -- 4316 vs 4430 spheres/s at 32 spheres per step; the overlapped step is GPU-bound, not
        host-bound -- so it is off by default; results are identical either way."""
        self.model, self.cfg = model, cfg
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("KPConvPipeline needs an MI355X device; there is no CPU fallback")
        with torch.cuda.device(self.device):
            # (the build is the critical path of a step: its stream gets the higher HIP priority, +1 % measured)
            self.build = torch.cuda.Stream(priority=-1)
            self.compute = torch.cuda.Stream(priority=0)
        self.pending = None          # (batch, built event)
        self.alive = []              # results whose forward may still be reading the batch tensors (allocated on `build`)
        self.pool = None
        if threaded:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="kpconv-forward")

    def _forward(self, pending):
        if pending is None:
            return None
        batch, built = pending
        with torch.cuda.device(self.device), torch.cuda.stream(self.compute):
            self.compute.wait_event(built)
            logits = self.model(batch)
            done = torch.cuda.Event()
            done.record(self.compute)
        res = KPConvPipeline.Result(logits, done, batch)
        # the batch's tensors live in the build stream's allocator pool: they must not be handed out again (to the NEXT build)
        # before this forward has finished -- keep the batch referenced until its event has completed
        self.alive = [r for r in self.alive if not r.done.query()]
        self.alive.append(res)
        while len(self.alive) > 2:
            self.alive.pop(0).done.synchronize()
        return res

    def _take_pending(self):
        pending, self.pending = self.pending, None
        return pending

    def submit(self, points, lengths, features=None, rotations="random"):
        from .torch.models.kpconv import KPConvBatch
        # the previous batch's forward goes first: it runs on the GPU while the host waits on this build's read-backs
        pending = self._take_pending()
        fut = self.pool.submit(self._forward, pending) if self.pool is not None else None
        prev = None if fut is not None else self._forward(pending)
        with torch.cuda.device(self.device):
            cur = torch.cuda.current_stream()
            self.build.wait_stream(cur)             # the caller's inputs
            with torch.cuda.stream(self.build):
                batch = KPConvBatch(points, lengths, self.cfg, features=features, rotations=rotations, device=self.device)
                built = torch.cuda.Event()
                built.record(self.build)
            if torch.is_tensor(points) and points.is_cuda:
                points.record_stream(self.build)
        if fut is not None:
            prev = fut.result()
        self.pending = (batch, built)
        return prev

    def flush(self):
        return self._forward(self._take_pending())

    def synchronize(self):
        self.build.synchronize()
        self.compute.synchronize()


class KPConvPipelineN:
    """``KPConvPipeline`` with SEVERAL batch builds in flight (default two), each on its own HIP stream, driven by its own host
    thread.  Why: the one-call build (``ml3d_kpconv_batch_build``) is a DEPENDENT chain of ~300 small launches cut by one
    blocking size read-back per layer -- ~6-7 ms of stream time for ~3 ms of kernels (round-5 profile: the chain is bound by
    per-launch latency, not by the host and not by the CUs), so ONE chain at a time leaves the GPU half idle.  The library call
    releases the interpreter lock while it waits, so two builder threads keep two chains in flight while the caller's thread
    enqueues the forward of the oldest finished batch on the compute stream.

    ``submit(points, lengths)`` starts the build of this batch and returns the ``Result`` of the batch submitted ``builders``
    calls earlier (None until then); ``flush()`` returns the list of the remaining results, oldest first.  Batches are
    forwarded in submission order; the random grid orientations are drawn on the caller's thread at submit time, in the order
    the sequential loop draws them (kpconv.py:2059-2080), so results are bit-identical to ``model(KPConvBatch(...))`` run in
    sequence with the same seed."""
    Result = KPConvPipeline.Result

    def __init__(self, model, cfg, device, builders=2, forward_streams=1, reuse_buffers=True):
        """``forward_streams``: consecutive batches' forwards alternate between this many compute streams (the deep layers' small
        kernels of one batch under the large ones of the next); 1 = every forward on one stream.
        ``reuse_buffers`` (default): the builds draw their workspace and output arena from a ring of ``builders + forward_streams +
        2`` buffer sets owned by the pipeline instead of fresh allocations -- nothing of the build goes through the caching
        allocator in steady state (on a cold box every pool growth was a 20-80 ms stall: ``profiles/r05_kp_cold_box.log``).  A
        ``Result.batch``'s tensors alias its ring entry: they are valid until that many more batches have been submitted (the logits
        are the result's own)."""
        from concurrent.futures import ThreadPoolExecutor
        self.model, self.cfg = model, cfg
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError("KPConvPipelineN needs an MI355X device; there is no CPU fallback")
        self.n = max(1, int(builders))
        with torch.cuda.device(self.device):
            self.build_streams = [torch.cuda.Stream(priority=-1) for _ in range(self.n)]
            self.computes = [torch.cuda.Stream(priority=0) for _ in range(max(1, int(forward_streams)))]
            self.compute = self.computes[0]       # (the stream of the most recently enqueued forward)
        self.forwards = 0
        self.ring = [dict() for _ in range(self.n + len(self.computes) + 2)] if reuse_buffers else None
        self.ring_done = [None] * (len(self.ring) if self.ring else 0)       # the forward that last read a ring entry's arena
        self.pool = ThreadPoolExecutor(max_workers=self.n, thread_name_prefix="kpconv-build")
        import threading
        self._slot_of_thread, self._slot_lock = {}, threading.Lock()
        self.inflight = []            # futures of (batch, built event), submission order
        self.alive = []
        self.count = 0
        self.pool_layers = sum(1 for b in cfg['architecture'] if 'pool' in b or 'strided' in b)
        # the model's folded / split weights (packed_params: uploads + pack kernels on the caller's stream) are built lazily in its
        # first forward: with forwards alternating between streams the second stream would read planes the first is still writing
        if hasattr(model, 'packed_params') and not getattr(model, 'training', False):
            with torch.cuda.device(self.device):
                model.packed_params(self.device)
                torch.cuda.current_stream(self.device).synchronize()

    def _build(self, points, lengths, features, rotations, ready, ring_index):
        from .torch.models.kpconv import KPConvBatch
        # The build stream belongs to the WORKER THREAD, not to the batch number: with `count % builders` the builds k and k + n shared
        # a stream and its pinned read-back scratch, and k + n starts as soon as ANY worker is free -- i.e. while k may still be
        # running when a later build overtook it: two host threads then drove one dependent chain each through the same stream and read
        # each other's size records (round 6: 2 of 8 runs of tests/test_gpu_kpconv.py [3 builders] -- logits of another batch's shapes,
        # or `ml3d_gather_pool: invalid argument`; never seen with 2 builders, the same hazard in principle).
        import threading
        tid = threading.get_ident()
        with self._slot_lock:
            slot = self._slot_of_thread.setdefault(tid, len(self._slot_of_thread)) % self.n
        st = self.build_streams[slot]
        buffers = None
        if self.ring is not None:
            buffers = self.ring[ring_index]
            prev = self.ring_done[ring_index]
            if prev is not None:
                st.wait_event(prev)                # the forward that read this entry's previous batch (enqueued long ago)
        with torch.cuda.device(self.device), torch.cuda.stream(st):
            st.wait_event(ready)                   # the caller's inputs
            batch = KPConvBatch(points, lengths, self.cfg, features=features, rotations=rotations, device=self.device, buffers=buffers)
            batch._ring_index = ring_index
            built = torch.cuda.Event()
            built.record(st)
            if torch.is_tensor(points) and points.is_cuda:
                points.record_stream(st)
        return batch, built

    def _forward(self, fut):
        batch, built = fut.result()
        st = self.compute = self.computes[self.forwards % len(self.computes)]
        self.forwards += 1
        with torch.cuda.device(self.device), torch.cuda.stream(st):
            st.wait_event(built)
            logits = self.model(batch)
            done = torch.cuda.Event()
            done.record(st)
        res = KPConvPipeline.Result(logits, done, batch)
        if self.ring is not None:
            self.ring_done[getattr(batch, '_ring_index', 0)] = done
        self.alive = [r for r in self.alive if not r.done.query()]
        self.alive.append(res)
        while len(self.alive) > self.n + len(self.computes):
            self.alive.pop(0).done.synchronize()
        return res

    def submit(self, points, lengths, features=None, rotations="random"):
        from .torch.models.kpconv import random_grid_rotations
        if isinstance(rotations, str):            # drawn HERE, in submission order (the builder threads must not race on np.random)
            rotations = [random_grid_rotations(len(lengths)) for _ in range(self.pool_layers)]
        with torch.cuda.device(self.device):
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
        ring_index = self.count % len(self.ring) if self.ring is not None else 0
        self.inflight.append(self.pool.submit(self._build, points, lengths, features, rotations, ready, ring_index))
        self.count += 1
        if len(self.inflight) > self.n:
            return self._forward(self.inflight.pop(0))
        return None

    def flush(self):
        out = []
        while self.inflight:
            out.append(self._forward(self.inflight.pop(0)))
        return out

    def synchronize(self):
        for s in self.build_streams + self.computes:
            s.synchronize()


class _PointPillarsLane:
    """One lane of ``PointPillarsStream``.  Host sweeps in, detections out (PointPillars detection, frame-parallel rows a15-a19).
    Two HIP streams: the pinned host sweeps of step i + 1 are uploaded on the copy stream while step i
    runs voxelize -> pillar features -> backbone -> heads -> batched box decode + NMS on the compute stream, and the few
    hundred KB of detections (rows + per-sample counts) travel back into pinned host buffers asynchronously; ``submit``
    returns the detections of the PREVIOUS step (three lists like ``Anchor3DHead.get_bboxes``; None on the first call) after
    waiting only for that step's copy event -- no device-wide synchronisation anywhere, the host runs one step ahead.
    ``flush`` returns the last step's detections."""

    def __init__(self, model, device):
        self.model = model
        self.device = _abi.require_gpu(device, "PointPillarsStream")
        with torch.cuda.device(self.device):
            self.h2d = torch.cuda.Stream()
            self.compute = torch.cuda.Stream()
            self.uploaded = [torch.cuda.Event(), torch.cuda.Event()]
            self.done = [torch.cuda.Event(), torch.cuda.Event()]
        self.host = [None, None]       # (rows pinned, total pinned)
        self.keep = [None, None]       # device tensors of a slot's step, referenced until its results have been fetched
        self.k = 0
        self.pending = None

    def _fetch(self, slot):
        self.done[slot].synchronize()
        rows, total = self.host[slot]
        out = self.model.bbox_head.split_rows(rows, total)
        self.keep[slot] = None
        return [b.clone() for b in out[0]], [s.clone() for s in out[1]], out[2]

    @torch.no_grad()
    def submit(self, host_clouds):
        slot = self.k & 1
        self.k += 1
        prev, self.pending = self.pending, slot
        with torch.cuda.device(self.device):
            with torch.cuda.stream(self.h2d):
                clouds = [h.to(self.device, non_blocking=True) for h in host_clouds]
                self.uploaded[slot].record(self.h2d)
            with torch.cuda.stream(self.compute):
                self.compute.wait_event(self.uploaded[slot])
                for c in clouds:
                    c.record_stream(self.compute)
                rows, total = self.model.detect(clouds)
                res = None
                if prev is not None:
                    res = self._fetch(prev)          # (its pinned buffers are free again before this step's copies are enqueued)
                if self.host[slot] is None or self.host[slot][0].shape != rows.shape:
                    self.host[slot] = (torch.empty(rows.shape, dtype=rows.dtype).pin_memory(),
                                       torch.empty(total.shape, dtype=total.dtype).pin_memory())
                self.host[slot][0].copy_(rows, non_blocking=True)
                self.host[slot][1].copy_(total, non_blocking=True)
                self.done[slot].record(self.compute)
                self.keep[slot] = (clouds, rows, total)
        return res

    def flush(self):
        prev, self.pending = self.pending, None
        return None if prev is None else self._fetch(prev)

    def synchronize(self):
        self.h2d.synchronize()
        self.compute.synchronize()


class PointPillarsStream:
    """Host sweeps in, detections out: what ``bench.py --workload pointpillars`` times.  The sweeps of a step are dealt to
    ``lanes`` independent pipelines (``_PointPillarsLane``: upload on a copy stream, voxelize -> pillar features -> backbone ->
    heads -> batched box decode + NMS on a compute stream, detections copied back to pinned host memory asynchronously, results
    handed out one step later after waiting only for that step's copy event).  Why lanes: a BEV convolution of 16 sweeps is 4.4
    rounds of tiles on the resident workgroups, a deeper one 3.3 or 1.6 -- the last, partly filled round of every launch leaves
    CUs idle, and nothing else is queued behind it on ONE stream.  With two lanes the other half-batch's kernels fill them:
    1244 -> 1404 frames/s at 16 sweeps per step (four lanes: 1060 -- the kernels get too small; ``gpurun_out/r3p``).
    ``submit`` returns the detections of the PREVIOUS step (three lists in the order of that step's sweeps; None on the first
    call), ``flush`` the last step's."""

    def __init__(self, model, device, lanes=2, threaded=True):
        """``threaded`` (default since round 5): every lane's ``submit`` runs on its own host thread.  A lane's step blocks the host
        once, in the voxelization's size read-back; with one host thread lane 1 is not even enqueued while lane 0 waits there.
        Measured, alternating on one box (``profiles/r05_pp_threaded_ab.log``): 1408 / 1407 / 1421 frames/s threaded against 1347 /
        1404 / 1339 -- the single-threaded pipeline's two per-process modes (rounds 3-4: 1310 or 1400, drawn per process) are the
        slow and the fast interleaving of the two lanes' host work; with a thread per lane only the fast one is left."""
        self.lanes = [_PointPillarsLane(model, device) for _ in range(max(1, int(lanes)))]
        # the folded / split weights are built lazily by the model (packed_params: uploads + pack kernels on the CALLER's stream):
        # built here, once, and waited for -- otherwise the lanes' threads would race to build them in their first step and a lane
        # could multiply with planes another lane's stream is still writing
        if hasattr(model, 'packed_params'):
            dev = self.lanes[0].device
            with torch.cuda.device(dev):
                model.packed_params(dev)
                torch.cuda.current_stream(dev).synchronize()
        self.compute = self.lanes[-1].compute          # (the stream whose completion events pace a step in the bench)
        self.pool = None
        if threaded and len(self.lanes) > 1:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=len(self.lanes), thread_name_prefix="pp-lane")

    @staticmethod
    def _merge(parts):
        parts = [p for p in parts if p is not None]
        if not parts:
            return None
        return tuple(sum((list(p[i]) for p in parts), []) for i in range(3))

    def submit(self, host_clouds):
        """A step with fewer sweeps than lanes (a final short step) uses the first lanes only; the idle lanes still hand over
        THEIR share of the previous step here, so the merged result always is exactly the previous step's detections, in order."""
        n, k = len(host_clouds), len(self.lanes)
        use = max(1, min(k, n))
        if self.pool is not None and use > 1:
            futs = [self.pool.submit(self.lanes[i].submit, host_clouds[i * n // use:(i + 1) * n // use]) for i in range(use)]
            parts = [f.result() for f in futs]
        else:
            parts = [self.lanes[i].submit(host_clouds[i * n // use:(i + 1) * n // use]) for i in range(use)]
        parts += [self.lanes[i].flush() for i in range(use, k)]
        return self._merge(parts)

    def flush(self):
        return self._merge([lane.flush() for lane in self.lanes])

    def synchronize(self):
        for lane in self.lanes:
            lane.synchronize()
