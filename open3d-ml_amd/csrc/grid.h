// grid.h — counting-sorted uniform grid shared by the k-NN and fixed-radius kernels.
//
// One grid per batch item ("segment").  Build = bbox -> multi-resolution
// occupancy probe -> cell size -> histogram -> scan -> scatter.  The sorted
// copy stores (x, y, z, local index) as one float4 so a candidate is ONE 16-byte
// load and a cell's points are contiguous (coalesced, L1/L2 friendly).
//
// The cell size only affects speed: the query kernels expand shells until the
// k-th distance is provably final, so results are exact for any grid.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ml3d {

constexpr int GRID_CAP = 8;      // dense cell table: at most GRID_CAP * n + 64 cells per segment
constexpr int GRID_SLACK = 64;
constexpr int GRID_LEVELS = 5;   // occupancy probed at cell sizes c0 * {1,2,4,8,16}

// How batch items are laid out in the point array.
//  splits != nullptr : item s = rows [splits[s], splits[s+1])           (packed == global row)
//  splits == nullptr : item s = rows [s*stride, s*stride + n_uniform)   (prefix of a [B, stride, 3] tensor)
struct Segs {
    const int64_t* splits;
    int64_t stride;
    int64_t n_uniform;
    int batch;
};

struct GridSeg {
    float lo[3];
    float c, inv_c;
    float margin;      // absolute slack subtracted from shell guarantee distances
    int dims[3];
    int cell_base;     // first cell of this segment in the cell table
    int n;             // points in the segment
    int sorted_base;   // first slot of this segment in the sorted array (== packed begin)
    float c0;          // finest probe cell size
    int dims0[3];
    float ext[3];      // bbox extent
    float dim_est;     // local dimension of the cloud estimated by the occupancy probe (1..3)
    int probe_stride;  // the probe looked at every probe_stride-th point
};

// Device-side view of a built grid.
struct GridView {
    const GridSeg* segs;
    const int* cell_start;   // start(j) = cell_start[j], end(j) = cell_start[j + 1]
    const float4* sorted;    // (x, y, z, bits(local index)), grouped by cell, segment-contiguous
    int batch;
};

// Workspace carve for one grid (all device memory, 256-byte aligned pieces).
struct GridWs {
    GridSeg* segs;
    unsigned* bbox;      // [batch][6] order-preserving uint encoding of float min/max
    unsigned* occ;       // [batch][GRID_LEVELS] occupied-cell counters
    unsigned* bitmap;    // [GRID_LEVELS][bitmap_words]
    int* cells;          // cell table, total_cells + 2
    int* block_sums;     // scan scratch
    float4* sorted;      // [n_total]
    int64_t total_cells;
    int64_t bitmap_words;
    int64_t n_total;
    int batch;
};

constexpr int GRID_SORTED_SLACK = 4;     // float4 entries kept readable behind GridWs::sorted (see grid_ws_bytes)
size_t grid_ws_bytes(int64_t n_total, int64_t batch);
// carve `ws` (must hold grid_ws_bytes) — returns false if too small
bool grid_ws_carve(void* ws, size_t bytes, int64_t n_total, int64_t batch, GridWs* out);
// enqueue the build; `points` row = 3 floats.  target_occ <= 0 -> default.
int grid_build(const float* points, Segs segs, const GridWs& ws, float target_occ, hipStream_t stream);
// Build the grid of a SUBSET of the points of `parent` (same batch items, e.g. the RandLA prefix
// sub-cloud): reuses the parent's bounding box and dimension estimate, cell size scaled for the
// thinner sampling — skips the bbox and occupancy passes.
int grid_build_derived(const float* points, Segs segs, const GridWs& ws, const GridWs& parent, hipStream_t stream);
// Build with a caller-chosen cell size (fixed-radius search: cell ~ radius, so a query's box is 3x3x3
// cells); skips the occupancy probe.  The size is grown until the dense table fits.
int grid_build_fixed(const float* points, Segs segs, const GridWs& ws, float cell, hipStream_t stream);
// bounding boxes only: bbox[6*s + {0..2}] = ordered-uint(min), [3..5] = ordered-uint(max)
int bbox_compute(const float* points, Segs segs, int64_t n_total, unsigned* bbox, unsigned* occ_scratch,
                 hipStream_t stream);
// in-place inclusive scan of a[0 .. n) (int32); block_sums must hold (n + 1023) / 1024 + 1 ints
int scan_inclusive_i32(int* a, int64_t n, int* block_sums, hipStream_t stream);

// zero `bytes` (a multiple of 4) at a 16-byte aligned device address with a fill KERNEL: what every op that may be captured into a
// HIP graph uses instead of hipMemsetAsync (ROCm 7.2: a graph's memset node did not clear its target on later replays, grid.hip)
void zero_async(void* ptr, size_t bytes, hipStream_t stream);

// order-preserving float <-> uint so atomicMin/atomicMax work on floats
__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// wave-level rendezvous + LDS/global visibility between the lanes of ONE wave
__device__ __forceinline__ void wave_sync() {
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
}

// the same rendezvous for LDS traffic ONLY: a wave's LDS instructions execute in issue order, so a ds_write is
// visible to a later ds_read of any lane of the same wave without draining the memory counters -- outstanding
// global loads (prefetches) and stores stay in flight across it.  (wave_sync's workgroup fence waits vmcnt(0).)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// max WITHOUT operand canonicalisation: fmaxf() first quiets possible signalling NaNs (v_max_f32 x, x, x), 3
// instructions for one max.  median(a, b, HUGE) = max(a, b) for a, b < HUGE is ONE v_med3_f32 (compiler-generated,
// so MFMA -> VALU hazards are still handled -- hand-written asm here raced with the MFMA results it consumed).
// On gfx950 every VALU instruction is issue time the f32 MFMAs cannot use, so the hot epilogues use this.
__device__ __forceinline__ float fmax_raw(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, 3.0e38f); }

// inclusive scan / sum across the 64 lanes of a wave
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// workgroup barrier that orders LDS traffic ONLY.  __syncthreads() is a full workgroup fence: it drains vmcnt too,
// i.e. every wave sits at the barrier until its outstanding global loads have landed and its global stores are
// acknowledged.  Kernels whose cross-wave communication is all in LDS use this one instead, so prefetches and
// output stores stay in flight across the barrier.
__device__ __forceinline__ void block_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

inline GridView grid_view(const GridWs& ws) {
    GridView v;
    v.segs = ws.segs;
    v.cell_start = ws.cells;
    v.sorted = ws.sorted;
    v.batch = ws.batch;
    return v;
}

// ---- device helpers ---------------------------------------------------------------------------

__device__ __forceinline__ int64_t seg_begin_global(const Segs& S, int s) {
    return S.splits ? S.splits[s] : (int64_t)s * S.stride;
}
__device__ __forceinline__ int64_t seg_begin_packed(const Segs& S, int s) {
    return S.splits ? S.splits[s] : (int64_t)s * S.n_uniform;
}
__device__ __forceinline__ int64_t seg_len(const Segs& S, int s) {
    return S.splits ? (S.splits[s + 1] - S.splits[s]) : S.n_uniform;
}
// packed index -> (segment, local index)
__device__ __forceinline__ void seg_locate(const Segs& S, int64_t packed, int& s, int64_t& local) {
    if (!S.splits) {
        s = (int)(packed / S.n_uniform);
        local = packed - (int64_t)s * S.n_uniform;
        return;
    }
    int lo = 0, hi = S.batch;  // find largest s with splits[s] <= packed; skips empty items
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (S.splits[mid] <= packed) lo = mid; else hi = mid;
    }
    s = lo;
    local = packed - S.splits[lo];
}

// The same for a WAVE-UNIFORM packed index, every lane of the wave calling: lane b holds splits[b + 1] (two loads cover 128 items) and the
// segment is a ballot count -- ONE memory round trip instead of the log2(batch) dependent ones of the binary search (7 for the 96
// spheres of a KPConv batch, in front of everything else a wave-per-query kernel does).  Returns the segment's first packed index too.
__device__ __forceinline__ void seg_locate_wave(const Segs& S, int64_t packed, int lane, int& s, int64_t& local, int64_t& begin) {
#ifdef ML3D_SEG_BINARY      // A/B build (tools/build_variant.sh): the binary search everywhere
    const bool search = true;
#else
    const bool search = !S.splits || S.batch > 128;
#endif
    if (search) {
        seg_locate(S, packed, s, local);
        begin = packed - local;
        return;
    }
    const long long big = 0x7fffffffffffffffll;
    const long long first = S.splits[0];
    const long long v0 = lane + 1 < S.batch ? (long long)S.splits[lane + 1] : big;
    long long v1 = big;
    if (S.batch > 64 && lane + 65 < S.batch) v1 = S.splits[lane + 65];
    int cnt = __popcll(__ballot(v0 <= (long long)packed));
    if (S.batch > 64) cnt += __popcll(__ballot(v1 <= (long long)packed));
    s = cnt;
    const long long src = cnt > 64 ? v1 : v0;
    const int from = cnt > 64 ? cnt - 65 : (cnt > 0 ? cnt - 1 : 0);
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(src & 0xffffffffll), from);
    const int hi = __builtin_amdgcn_readlane((int)(src >> 32), from);
    const long long held = (long long)(unsigned)lo | ((long long)hi << 32);
    begin = cnt > 0 ? held : first;
    local = packed - begin;
}

__device__ __forceinline__ int cell_coord(float p, float lo, float inv_c, int dim) {
    int v = (int)((p - lo) * inv_c);
    v = v < 0 ? 0 : v;
    return v > dim - 1 ? dim - 1 : v;
}

// canonical squared distance: ((dx*dx)+(dy*dy))+(dz*dz), every step rounded (no fma)
__device__ __forceinline__ float dist2_canon(float qx, float qy, float qz, float px, float py, float pz) {
    float dx = __fsub_rn(qx, px), dy = __fsub_rn(qy, py), dz = __fsub_rn(qz, pz);
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

}  // namespace ml3d
