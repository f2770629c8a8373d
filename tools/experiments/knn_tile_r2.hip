// knn.hip — exact k-nearest-neighbour search on the counting-sorted grid (gfx950).
//
// Replaces o3c.nns.NearestNeighborSearch.knn_search as called from
// ml3d/datasets/utils/dataprocessing.py:99-103 (RandLANet.transform,
// ml3d/torch/models/randlanet.py:218-229).  Result order is the oracle's
// canonical one: ascending (d2, index), d2 = ((dx*dx)+(dy*dy))+(dz*dz) in f32
// without fma — so indices are bit-exact against oracle/ml3d_oracle.c.
//
// One thread per query.  Queries are visited in the CELL-SORTED order of their
// own grid, so the 64 lanes of a wave sit in the same or adjacent cells: the
// 16-byte candidate loads of neighbouring lanes hit the same lines (L1/L2), and
// loop trip counts are similar across the wave.  The running best-k list lives
// in registers as K packed 64-bit keys (bits(d2) << 32 | index): one unsigned
// compare orders (d2, index) pairs exactly.
//
// Roofline: HBM-bound by design intent (12 B/query read + 4*k B/query written),
// in practice VALU/latency-bound on the candidate loop; see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <gfx950_ops.h>

#include "grid.h"
#include "ml3d_hip.h"

namespace ml3d {

typedef unsigned long long u64;
constexpr u64 KEY_EMPTY = 0x7ff0000000000000ull;   // +inf as a double: above every real key

// Where queries come from:
//  sorted_q != nullptr : query t = sorted_q[t] (xyz + local index) of grid `qsegs` (packed order)
//  else                : query t = raw[global(t)]
struct QuerySrc {
    const float4* sorted_q;
    const GridSeg* qsegs;   // segs of the query grid (for sorted_base -> segment lookup)
    const float* raw;
    Segs segs;              // layout of the raw queries / output rows
    int64_t n_total;
};

// The best-k list is kept as K doubles whose BIT PATTERNS are the packed keys bits(d2) << 32 | index: for
// finite non-negative d2 the IEEE-754 double order of those patterns equals their unsigned integer order
// (exponent+mantissa are compared like an integer; a pattern is NaN/inf only when the float d2 is), so one
// compare-exchange of (d2, index) pairs is v_min_f64 + v_max_f64 (key_minmax, gfx950_ops.h) instead of a 64-bit compare
// and four selects.  d2 == 0 gives a denormal double, which f64 min/max preserve (f64 denormals are never flushed on
// gfx950).
template <int K>
__device__ __forceinline__ void topk_insert(double (&best)[K], double key) {
    if (key < best[K - 1]) {
        best[K - 1] = key;
#pragma unroll
        for (int j = K - 1; j > 0; --j) key_minmax(best[j - 1], best[j], best[j - 1], best[j]);
    }
}

#ifndef KNN_GROUP
#define KNN_GROUP 3
#endif

template <int K>
__device__ __forceinline__ void scan_run(const GridView& G, int cell_a, int cell_b, float qx, float qy,
                                         float qz, double (&best)[K]) {
    int p0 = G.cell_start[cell_a], p1 = G.cell_start[cell_b + 1];
    // candidates in groups of KNN_GROUP: the 16-byte loads of a group are in flight together (one exposed memory latency
    // per group instead of one per candidate -- the loop is latency-bound: lane-per-query gathers, ~5 waves per SIMD);
    // indices past the run are clamped to its last point and skipped
    for (int p = p0; p < p1; p += KNN_GROUP) {
        float4 c[KNN_GROUP];
#pragma unroll
        for (int j = 0; j < KNN_GROUP; ++j) c[j] = G.sorted[min(p + j, p1 - 1)];
#pragma unroll
        for (int j = 0; j < KNN_GROUP; ++j) {
            if (p + j < p1) {
                float d2 = dist2_canon(qx, qy, qz, c[j].x, c[j].y, c[j].z);
                u64 key = ((u64)__float_as_uint(d2) << 32) | (u64)(unsigned)__float_as_int(c[j].w);
                topk_insert<K>(best, __longlong_as_double((long long)key));
            }
        }
    }
}

// lane-per-query shell search of ONE query on grid segment g: the 3x3x3 block of cells first, then shells, until the k-th
// distance is strictly inside the scanned box.  Exact for any input; used for k > 16 and as the tile kernel's way out
// of tiles whose queries are too far apart to share candidates.
template <int K>
__device__ __forceinline__ void shell_search(const GridView& G, const GridSeg& g, float qx, float qy, float qz, int k,
                                             double (&best)[K]) {
    int cx = cell_coord(qx, g.lo[0], g.inv_c, g.dims[0]);
    int cy = cell_coord(qy, g.lo[1], g.inv_c, g.dims[1]);
    int cz = cell_coord(qz, g.lo[2], g.inv_c, g.dims[2]);
    int dxm = g.dims[0] - 1, dym = g.dims[1] - 1, dzm = g.dims[2] - 1;
    for (int r = 1;; ++r) {
        int xa = max(cx - r, 0), xb = min(cx + r, dxm);
        int ya = max(cy - r, 0), yb = min(cy + r, dym);
        int za = max(cz - r, 0), zb = min(cz + r, dzm);
        for (int z = za; z <= zb; ++z) {
            int az = z > cz ? z - cz : cz - z;
            for (int y = ya; y <= yb; ++y) {
                int ay = y > cy ? y - cy : cy - y;
                int row = g.cell_base + g.dims[0] * (y + g.dims[1] * z);
                if (r == 1 || az == r || ay == r) {
                    scan_run<K>(G, row + xa, row + xb, qx, qy, qz, best);
                } else {
                    if (cx - r >= 0) scan_run<K>(G, row + cx - r, row + cx - r, qx, qy, qz, best);
                    if (cx + r <= dxm) scan_run<K>(G, row + cx + r, row + cx + r, qx, qy, qz, best);
                }
            }
        }
        // every point outside the scanned box is at least `gd` away (inf when the box face is
        // past the grid).  Stop once the k-th best is strictly inside that radius.
        bool all = (cx - r <= 0) && (cx + r >= dxm) && (cy - r <= 0) && (cy + r >= dym) &&
                   (cz - r <= 0) && (cz + r >= dzm);
        if (all) break;
        float gd = 3.0e38f;
        if (cx - r > 0) gd = fminf(gd, qx - (g.lo[0] + (float)(cx - r) * g.c));
        if (cx + r < dxm) gd = fminf(gd, (g.lo[0] + (float)(cx + r + 1) * g.c) - qx);
        if (cy - r > 0) gd = fminf(gd, qy - (g.lo[1] + (float)(cy - r) * g.c));
        if (cy + r < dym) gd = fminf(gd, (g.lo[1] + (float)(cy + r + 1) * g.c) - qy);
        if (cz - r > 0) gd = fminf(gd, qz - (g.lo[2] + (float)(cz - r) * g.c));
        if (cz + r < dzm) gd = fminf(gd, (g.lo[2] + (float)(cz + r + 1) * g.c) - qz);
        gd -= g.margin;
        u64 kth = (u64)__double_as_longlong(best[K - 1]);
        if (k < K) {
            // fewer than K requested: the k-th entry decides
#pragma unroll
            for (int j = 0; j < K; ++j) if (j == k - 1) kth = (u64)__double_as_longlong(best[j]);
        }
        if (kth != KEY_EMPTY && gd > 0.f) {
            float dk = __uint_as_float((unsigned)(kth >> 32));
            if (dk < gd * gd * 0.999999f) break;
        }
    }
}

template <int K>
__device__ __forceinline__ void knn_one(const GridView& G, const QuerySrc& Q, int k, int index_local,
                                        const Segs& support_segs, int32_t* __restrict__ out_idx,
                                        float* __restrict__ out_d2, int64_t t) {
    if (t >= Q.n_total) return;
    int s; int64_t local;
    float qx, qy, qz;
    seg_locate(Q.segs, t, s, local);
    if (Q.sorted_q) {
        float4 q = Q.sorted_q[t];
        qx = q.x; qy = q.y; qz = q.z;
        local = __float_as_int(q.w);
    } else {
        const float* p = Q.raw + 3 * (seg_begin_global(Q.segs, s) + local);
        qx = p[0]; qy = p[1]; qz = p[2];
    }
    int64_t out_row = seg_begin_packed(Q.segs, s) + local;

    const GridSeg g = G.segs[s];
    double best[K];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = __longlong_as_double((long long)KEY_EMPTY);
    if (g.n > 0) shell_search<K>(G, g, qx, qy, qz, k, best);
    int64_t base = index_local ? 0 : seg_begin_global(support_segs, s);
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < k) {
            u64 key = (u64)__double_as_longlong(best[j]);
            bool ok = key != KEY_EMPTY;
            out_idx[out_row * k + j] = ok ? (int32_t)((int64_t)(unsigned)(key & 0xffffffffull) + base) : -1;
            if (out_d2) out_d2[out_row * k + j] = ok ? __uint_as_float((unsigned)(key >> 32)) : __uint_as_float(0x7f800000u);
        }
    }
}

template <int K>
__global__ void __launch_bounds__(256)
knn_query(GridView G, QuerySrc Q, int k, int index_local, Segs support_segs, int32_t* __restrict__ out_idx,
          float* __restrict__ out_d2) {
    knn_one<K>(G, Q, k, index_local, support_segs, out_idx, out_d2, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Several independent searches share ONE launch (the levels of the RandLA pyramid): the small levels alone cannot fill
// the chip; side by side they hide under the largest level.  Jobs are ordered largest first.
constexpr int KNN_MAX_JOBS = 8;

static int launch_query(const GridView& G, const QuerySrc& Q, int k, int index_local, Segs support_segs,
                        int32_t* out_idx, float* out_d2, hipStream_t stream) {
    if (Q.n_total <= 0) return 0;
    dim3 grid((unsigned)((Q.n_total + 255) / 256)), block(256);
    if (k <= 32)
        hipLaunchKernelGGL(knn_query<32>, grid, block, 0, stream, G, Q, k, index_local, support_segs, out_idx, out_d2);
    else if (k <= 64)
        hipLaunchKernelGGL(knn_query<64>, grid, block, 0, stream, G, Q, k, index_local, support_segs, out_idx, out_d2);
    else
        return ML3D_E_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

// =====================================================================================================================
// Tile kernel (k <= 16): ONE WAVE = 64 queries that are close in space (a "tile": 64 consecutive entries of the brick
// order, grid.h).  The wave stages the support points of the cells around the tile's bounding box into LDS once
// (coalesced float4 loads, a cell row = one contiguous run of the sorted array) and EVERY lane scans the same candidate
// list from LDS (broadcast reads): no per-lane cell walk, no divergent trip counts, no exposed global latency in the
// scan.  Candidates are consumed 16 at a time: 16 keys -> 60-comparator sorting network -> half-cleaner against the
// sorted best-16 (min(best[i], new[15 - i])) -> 4-stage bitonic merge; every compare-exchange is v_min_f64 + v_max_f64
// on the packed keys.  ~200 instructions per 16 candidates whatever the data (the per-lane insertion it replaces cost
// ~34 per candidate as soon as ONE lane of the wave accepted it).
//
// Exactness: pass 1 scans bbox(tile) + a small halo; a lane is final once its k-th distance is strictly inside the
// scanned box (same face-distance test as the shell search).  The k-th distances found so far bound the true ones from
// above, so pass 2 scans exactly the cells of the hull of the still-open lanes' balls minus what was scanned -- and is
// final for every lane that had k candidates.  Lanes with fewer than k (tiny / far-away supports) double the box.
// =====================================================================================================================
#ifndef KNN_TILE_WAVES
#define KNN_TILE_WAVES 3          // register budget: 512 / 3 = 170 VGPRs (two sorted 16-key lists + 8 staged candidates)
#endif
constexpr int TILE_CH = 256;                // candidates staged per chunk and wave (4 KB of LDS)
constexpr int PAD_IDX = 0x7fffffff;         // index of the (+inf, +inf, +inf) filler candidates
#ifndef KNN_TILE_MAX_CAND
#define KNN_TILE_MAX_CAND 1024
#endif
constexpr int TILE_MAX_CAND = KNN_TILE_MAX_CAND;         // candidates one pass of one tile may stage (typical: 250 + 100)
constexpr int TILE_MAX_ROWS = 512;          // cell rows (x 2 in later passes) one pass may look up
constexpr int TILE_MAX_PASSES = 3;

struct TileJob {
    GridView G;                 // support grid
    const float4* qorder;       // queries in tile order (x, y, z, bits(local index)), segment-contiguous
    const int* tile_splits;     // [batch + 1] first tile of every segment (used when qsegs.splits != nullptr)
    Segs qsegs;                 // layout of the queries / output rows
    Segs support;               // layout of the support items (global index base)
    int32_t* out_idx;
    float* out_d2;
    unsigned tile_begin;        // first tile of this job in the launch
    unsigned n_tiles;           // upper bound of this job's tile count
};
struct TileJobs {
    TileJob j[KNN_MAX_JOBS];
    int n;
};

#ifdef ML3D_KNN_STATS
// emulator-only instrumentation (tests/hipemu build with -DML3D_KNN_STATS): tiles, candidates, blocks merged / skipped, passes
unsigned long long g_knn_stats[8];
extern "C" unsigned long long* ml3d_knn_stats() { return g_knn_stats; }
#define KNN_STAT(i, v) do { if (lane == 0) atomicAdd(&g_knn_stats[i], (unsigned long long)(v)); } while (0)
#else
#define KNN_STAT(i, v) do { } while (0)
#endif

#ifdef KNN_PROF
// developer instrumentation (variant builds only): shader-clock cycles per phase of the tile kernel, one slot per wave
constexpr int PROF_WAVES = 1 << 17;
__device__ unsigned g_knn_prof[PROF_WAVES][8];
extern "C" int ml3d_knn_prof_read(unsigned* host, int waves) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_knn_prof), sizeof(unsigned) * 8 * (size_t)waves);
}
extern "C" int ml3d_knn_prof_reset() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_knn_prof)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(unsigned) * 8 * (size_t)PROF_WAVES);
}
#define PROF_DECL unsigned prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PROF_T(v) const long long v = clock64()
#define PROF_ADD(i, a, b) prof_acc[i] += (unsigned)((b) - (a))
#define PROF_FLUSH(w) do { if (lane == 0 && (w) < PROF_WAVES) { for (int i_ = 0; i_ < 8; ++i_) g_knn_prof[w][i_] = prof_acc[i_]; } } while (0)
#else
#define PROF_DECL do { } while (0)
#define PROF_T(v) do { } while (0)
#define PROF_ADD(i, a, b) do { } while (0)
#define PROF_FLUSH(w) do { } while (0)
#endif

struct CellBox { int xa, xb, ya, yb, za, zb; };

// wave reductions; the result is handed back through readfirstlane so the compiler KNOWS it is wave-uniform (SGPR):
// everything derived from it -- the cell box, loop bounds, branches -- then runs on the scalar unit
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
// min and max of three values each in ONE sweep: the six shuffle chains are independent, so every step issues six
// cross-lane moves back to back instead of paying the cross-lane latency 36 times in a row
__device__ __forceinline__ void wave_minmax3(const float (&v)[3], bool take, float (&lo)[3], float (&hi)[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = take ? v[a] : 3.0e38f; hi[a] = take ? v[a] : -3.0e38f; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float l2[3], h2[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { l2[a] = __shfl_xor(lo[a], o); h2[a] = __shfl_xor(hi[a], o); }
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], l2[a]); hi[a] = fmaxf(hi[a], h2[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = uni(lo[a]); hi[a] = uni(hi[a]); }
}
__device__ __forceinline__ void wave_minmax3_i(int (&lo)[3], int (&hi)[3]) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int l2[3], h2[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { l2[a] = __shfl_xor(lo[a], o); h2[a] = __shfl_xor(hi[a], o); }
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], l2[a]); hi[a] = max(hi[a], h2[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = uni(lo[a]); hi[a] = uni(hi[a]); }
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return uni(v);
}
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}

#define KCE(i, j) key_minmax(v[i], v[j], v[i], v[j])
// 60-comparator, 10-layer sorting network for 16 keys (ascending); verified exhaustively (0-1 principle) by
// tests/test_host_logic.py::test_sort16_network
__device__ __forceinline__ void sort16(double (&v)[16]) {
    KCE(0, 13); KCE(1, 12); KCE(2, 15); KCE(3, 14); KCE(4, 8); KCE(5, 6); KCE(7, 11); KCE(9, 10);
    KCE(0, 5); KCE(1, 7); KCE(2, 9); KCE(3, 4); KCE(6, 13); KCE(8, 14); KCE(10, 15); KCE(11, 12);
    KCE(0, 1); KCE(2, 3); KCE(4, 5); KCE(6, 8); KCE(7, 9); KCE(10, 11); KCE(12, 13); KCE(14, 15);
    KCE(0, 2); KCE(1, 3); KCE(4, 10); KCE(5, 11); KCE(6, 7); KCE(8, 9); KCE(12, 14); KCE(13, 15);
    KCE(1, 2); KCE(3, 12); KCE(4, 6); KCE(5, 7); KCE(8, 10); KCE(9, 11); KCE(13, 14);
    KCE(1, 4); KCE(2, 6); KCE(5, 8); KCE(7, 10); KCE(9, 13); KCE(11, 14);
    KCE(2, 4); KCE(3, 6); KCE(9, 12); KCE(11, 13);
    KCE(3, 5); KCE(6, 8); KCE(7, 9); KCE(10, 12);
    KCE(3, 4); KCE(5, 6); KCE(7, 8); KCE(9, 10); KCE(11, 12);
    KCE(6, 7); KCE(8, 9);
}
// best (sorted) <- the 16 smallest of best U fresh (fresh sorted): half-cleaner + bitonic merge
__device__ __forceinline__ void merge16(double (&v)[16], const double (&fresh)[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = key_min(v[i], fresh[15 - i]);
#pragma unroll
    for (int d = 8; d > 0; d >>= 1)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if ((i & d) == 0) KCE(i, i + d);
}
#undef KCE

__device__ __forceinline__ double pack_key(float d2, float idx_bits) {
    return __longlong_as_double((long long)(((u64)__float_as_uint(d2) << 32) | (u64)__float_as_uint(idx_bits)));
}

// ---------------------------------------------------------------------------------------------------------------------
// Straggler search: ONE query at a time, the whole wave on it.  Used for the lanes a tile cannot finish cheaply -- an
// isolated point (a return off a box top, metres above a dense ground sheet) whose k-th neighbour is so far that the box
// around its ball holds thousands of points.  A lane-per-query search would keep 63 lanes waiting for it; here the roles
// flip: lane = CANDIDATE.  The query's sorted best-k list lives one key per lane (lanes 0..k-1); the wave streams the
// cells of the ball that were not scanned yet -- row by row, each row clipped to the ball's chord (sphere, not box) --
// 64 candidates per step, ballots the ones below the current k-th key and inserts those few (readlane + one shifted
// copy of the list).  ~20 instructions per 64 candidates.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (long long)(unsigned)lo);
}

template <int K>
__device__ __forceinline__ void straggler_search(const TileJob& jb, const GridSeg& g, CellBox S, unsigned long long todo, int lane,
                                              float qx, float qy, float qz, int k, double (&best)[K], float4* cand,
                                              int* rstart, int* roff) {
    const int dxm = g.dims[0] - 1, dym = g.dims[1] - 1, dzm = g.dims[2] - 1;
    const CellBox S0 = S;
    while (todo) {
        const int L = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const float ux = uni(__shfl(qx, L)), uy = uni(__shfl(qy, L)), uz = uni(__shfl(qz, L));
        double lk = __longlong_as_double((long long)KEY_EMPTY);          // lane j < K: j-th best key of query L
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const double v = readlane_f64(best[j], L);
            if (lane == j) lk = v;
        }
        S = S0;
        for (;;) {
            const u64 kth = (u64)__double_as_longlong(readlane_f64(lk, k - 1));
            const bool found = kth != KEY_EMPTY && (unsigned)(kth >> 32) < 0x7f800000u;
            const float dk = __uint_as_float((unsigned)(kth >> 32));
            const bool all = S.xa <= 0 && S.xb >= dxm && S.ya <= 0 && S.yb >= dym && S.za <= 0 && S.zb >= dzm;
            if (all) break;
            float gd = 3.0e38f;
            if (S.xa > 0) gd = fminf(gd, ux - (g.lo[0] + (float)S.xa * g.c));
            if (S.xb < dxm) gd = fminf(gd, (g.lo[0] + (float)(S.xb + 1) * g.c) - ux);
            if (S.ya > 0) gd = fminf(gd, uy - (g.lo[1] + (float)S.ya * g.c));
            if (S.yb < dym) gd = fminf(gd, (g.lo[1] + (float)(S.yb + 1) * g.c) - uy);
            if (S.za > 0) gd = fminf(gd, uz - (g.lo[2] + (float)S.za * g.c));
            if (S.zb < dzm) gd = fminf(gd, (g.lo[2] + (float)(S.zb + 1) * g.c) - uz);
            gd -= g.margin;
            if (found && gd > 0.f && dk < gd * gd * 0.999999f) break;
            // next box: the ball of the current k-th distance (an upper bound of the final one), or twice the box
            CellBox N = S;
            float rad = 0.f;
            if (found) {
                rad = sqrtf(dk) * 1.000001f + 2.0f * g.margin;
                N.xa = min(N.xa, cell_coord(ux - rad, g.lo[0], g.inv_c, g.dims[0]));
                N.xb = max(N.xb, cell_coord(ux + rad, g.lo[0], g.inv_c, g.dims[0]));
                N.ya = min(N.ya, cell_coord(uy - rad, g.lo[1], g.inv_c, g.dims[1]));
                N.yb = max(N.yb, cell_coord(uy + rad, g.lo[1], g.inv_c, g.dims[1]));
                N.za = min(N.za, cell_coord(uz - rad, g.lo[2], g.inv_c, g.dims[2]));
                N.zb = max(N.zb, cell_coord(uz + rad, g.lo[2], g.inv_c, g.dims[2]));
            } else {
                const int wx = S.xb - S.xa + 1, wy = S.yb - S.ya + 1, wz = S.zb - S.za + 1;
                N.xa = max(S.xa - wx, 0); N.xb = min(S.xb + wx, dxm);
                N.ya = max(S.ya - wy, 0); N.yb = min(S.yb + wy, dym);
                N.za = max(S.za - wz, 0); N.zb = min(S.zb + wz, dzm);
            }
            bool clip = found;      // rows are clipped to the ball's chord; not when the box had to be grown artificially
            if (N.xa == S.xa && N.xb == S.xb && N.ya == S.ya && N.yb == S.yb && N.za == S.za && N.zb == S.zb) {
                N.xa = max(S.xa - 1, 0); N.xb = min(S.xb + 1, dxm);
                N.ya = max(S.ya - 1, 0); N.yb = min(S.yb + 1, dym);
                N.za = max(S.za - 1, 0); N.zb = min(S.zb + 1, dzm);
                clip = false;
            }
            // ---- stream the cells of N \ S -------------------------------------------------------------------------
            const int ny = N.yb - N.ya + 1, nz = N.zb - N.za + 1;
            const int64_t items64 = (int64_t)ny * nz * 2;
            const int items = (int)min(items64, (int64_t)0x7ffffff0);
            const float rad2 = rad * rad;
            for (int item_base = 0; item_base < items; item_base += 64) {
                const int it = item_base + lane;
                int start = 0, len = 0;
                if (it < items) {
                    const int part = it & 1, r = it >> 1;
                    const int y = N.ya + r % ny, z = N.za + r / ny;
                    int x0 = N.xa, x1 = N.xb;
                    const bool inside = y >= S.ya && y <= S.yb && z >= S.za && z <= S.zb;
                    if (inside) {
                        if (part == 0) x1 = S.xa - 1; else x0 = S.xb + 1;
                    } else if (part == 1) {
                        x1 = x0 - 1;
                    }
                    if (clip && x0 <= x1) {
                        // distance from the query to the row's cell column in y and z (0 when it is inside the column)
                        const float ylo = g.lo[1] + (float)y * g.c, zlo = g.lo[2] + (float)z * g.c;
                        const float dy = fmaxf(fmaxf(ylo - uy, uy - (ylo + g.c)), 0.f) - g.margin;
                        const float dz = fmaxf(fmaxf(zlo - uz, uz - (zlo + g.c)), 0.f) - g.margin;
                        const float dyz2 = fmaxf(dy, 0.f) * fmaxf(dy, 0.f) + fmaxf(dz, 0.f) * fmaxf(dz, 0.f);
                        if (dyz2 > rad2) {
                            x1 = x0 - 1;
                        } else {
                            const float xw = sqrtf(rad2 - dyz2) * 1.000001f + 2.0f * g.margin;
                            x0 = max(x0, cell_coord(ux - xw, g.lo[0], g.inv_c, g.dims[0]));
                            x1 = min(x1, cell_coord(ux + xw, g.lo[0], g.inv_c, g.dims[0]));
                        }
                    }
                    if (x0 <= x1) {
                        const int row = g.cell_base + g.dims[0] * (y + g.dims[1] * z);
                        start = jb.G.cell_start[row + x0];
                        len = jb.G.cell_start[row + x1 + 1] - start;
                    }
                }
                const int incl = wave_incl_scan_i(len, lane);
                const int total = __builtin_amdgcn_readlane(incl, 63);
                if (total == 0) continue;
                rstart[lane] = start;
                roff[lane] = incl - len;
                wave_lds_sync();
                for (int e = 0; e < total; e += 64) {
                    const int gi = e + lane;
                    const bool ok = gi < total;
                    int lo = 0;
#pragma unroll
                    for (int st = 32; st > 0; st >>= 1)
                        if (roff[lo + st] <= min(gi, total - 1)) lo += st;
                    const float4 c = jb.G.sorted[rstart[lo] + (min(gi, total - 1) - roff[lo])];
                    const double key = pack_key(dist2_canon(ux, uy, uz, c.x, c.y, c.z), c.w);
                    double kk = readlane_f64(lk, K - 1);        // a slot of the list is free or beaten: candidate enters
                    unsigned long long m = __ballot(ok && key < kk);
                    while (m) {
                        const int j = __ffsll((long long)m) - 1;
                        m &= m - 1;
                        const double x = readlane_f64(key, j);
                        // position = number of list keys below x; lanes at / above it take their lower neighbour's key
                        const int pos = __popcll(__ballot(lane < K && lk < x));
                        const double up = __shfl_up(lk, 1);
                        if (lane < K && lane >= pos) lk = lane == pos ? x : up;
                    }
                }
                wave_lds_sync();
            }
            S = N;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const double v = readlane_f64(lk, j);
            if (lane == L) best[j] = v;
        }
    }
}

template <int K>
__global__ void __launch_bounds__(256) ML3D_WAVES_PER_SIMD(KNN_TILE_WAVES) knn_tile(TileJobs J, int k, int index_local) {
    static_assert(K == 1 || K == 16, "tile kernel: k = 1 or k <= 16");
    __shared__ float4 s_cand[4][TILE_CH];
    __shared__ int s_rstart[4][64];
    __shared__ int s_roff[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float4* cand = s_cand[wv];
    int* rstart = s_rstart[wv];
    int* roff = s_roff[wv];

    PROF_DECL;
    PROF_T(t_begin);
    const unsigned tile_g = blockIdx.x * 4 + wv;
    int ji = 0;
#pragma unroll
    for (int i = 1; i < KNN_MAX_JOBS; ++i)
        if (i < J.n && tile_g >= J.j[i].tile_begin) ji = i;
    const TileJob& jb = J.j[ji];
    const unsigned t = tile_g - jb.tile_begin;
    if (t >= jb.n_tiles) return;

    // ---- which queries (wave-uniform) ----------------------------------------------------------------------------
    int s, cnt;
    int64_t first;
    if (!jb.qsegs.splits) {
        const int64_t nu = jb.qsegs.n_uniform;
        const unsigned tps = (unsigned)((nu + 63) / 64);
        if (tps == 0) return;
        s = (int)(t / tps);
        if (s >= jb.qsegs.batch) return;
        const int64_t tl = (int64_t)(t - (unsigned)s * tps) * 64;
        first = (int64_t)s * nu + tl;
        cnt = (int)min((int64_t)64, nu - tl);
    } else {
        const int B = jb.qsegs.batch;
        if ((int)t >= jb.tile_splits[B]) return;
        int lo = 0, hi = B;                       // largest s with tile_splits[s] <= t (never an empty item)
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (jb.tile_splits[mid] <= (int)t) lo = mid; else hi = mid;
        }
        s = lo;
        const int64_t tl = (int64_t)((int)t - jb.tile_splits[s]) * 64;
        first = jb.qsegs.splits[s] + tl;
        cnt = (int)min((int64_t)64, jb.qsegs.splits[s + 1] - jb.qsegs.splits[s] - tl);
    }
    const bool valid = lane < cnt;
    const float4 q4 = jb.qorder[first + (valid ? lane : cnt - 1)];   // idle lanes shadow the tile's last query
    const float qx = q4.x, qy = q4.y, qz = q4.z;
    const GridSeg g = jb.G.segs[s];

    double best[K];
    const int64_t out_base = index_local ? 0 : seg_begin_global(jb.support, s);
    const int64_t out_row = seg_begin_packed(jb.qsegs, s) + (int64_t)__float_as_int(q4.w);

    // A tile is normally ONE group of 64 lanes.  When its queries are too far apart to share candidates (the 64 points
    // straddle two bricks that are not neighbours: the box between them holds thousands of points) the group is halved --
    // lanes [gbeg, gbeg + gsize) search alone, the rest of the wave rides along masked -- down to 16 lanes; what is still
    // too big then goes to the lane-per-query shell search.
    int gbeg = 0, gsize = 64;
    while (gbeg < 64) {
    const bool active = lane >= gbeg && lane < gbeg + gsize;
    bool split = false;
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = __longlong_as_double((long long)KEY_EMPTY);

    if (g.n > 0) {
        const int dxm = g.dims[0] - 1, dym = g.dims[1] - 1, dzm = g.dims[2] - 1;
        // ---- pass-1 box: the group's bounding box plus a small halo, in cells --------------------------------------
        const float h0 = g.c * (K == 1 ? 0.25f : 0.5f);
        CellBox R, Rold = {0, -1, 0, -1, 0, -1};
        {
            const float qv[3] = {qx, qy, qz};
            float blo[3], bhi[3];
            wave_minmax3(qv, active, blo, bhi);
            R.xa = uni(cell_coord(blo[0] - h0, g.lo[0], g.inv_c, g.dims[0]));
            R.xb = uni(cell_coord(bhi[0] + h0, g.lo[0], g.inv_c, g.dims[0]));
            R.ya = uni(cell_coord(blo[1] - h0, g.lo[1], g.inv_c, g.dims[1]));
            R.yb = uni(cell_coord(bhi[1] + h0, g.lo[1], g.inv_c, g.dims[1]));
            R.za = uni(cell_coord(blo[2] - h0, g.lo[2], g.inv_c, g.dims[2]));
            R.zb = uni(cell_coord(bhi[2] + h0, g.lo[2], g.inv_c, g.dims[2]));
        }
        bool has_old = false;
        bool open = true;                 // this lane's list is not final yet
        KNN_STAT(0, 1);
        PROF_T(t_pro);
        PROF_ADD(0, t_begin, t_pro);

        for (int pass = 0;; ++pass) {
            KNN_STAT(5, 1);
            PROF_T(t_p0);
            // ---- scan the cells of R that are not in Rold --------------------------------------------------------
            // work items: one x-run of cells per (y, z) row of R -- two when the row also crosses Rold (left / right part)
            const int ny = R.yb - R.ya + 1, nz = R.zb - R.za + 1;
            const int parts = has_old ? 2 : 1;
            const int64_t items64 = (int64_t)ny * nz * parts;
            const int items = (int)min(items64, (int64_t)0x7fffffff);
            // run `it` of the scan: the cells [x0, x1] of row (y, z) -> (first slot, length) in the sorted array
            auto run_of = [&](int it, int& start, int& len) {
                start = 0; len = 0;
                if (it >= items) return;
                const int part = has_old ? (it & 1) : 0;
                const int r = has_old ? (it >> 1) : it;
                const int y = R.ya + r % ny, z = R.za + r / ny;
                int x0 = R.xa, x1 = R.xb;
                if (has_old) {
                    const bool inside = y >= Rold.ya && y <= Rold.yb && z >= Rold.za && z <= Rold.zb;
                    if (inside) {
                        if (part == 0) x1 = Rold.xa - 1; else x0 = Rold.xb + 1;
                    } else if (part == 1) {
                        x1 = x0 - 1;
                    }
                }
                if (x0 <= x1) {
                    const int row = g.cell_base + g.dims[0] * (y + g.dims[1] * z);
                    start = jb.G.cell_start[row + x0];
                    len = jb.G.cell_start[row + x1 + 1] - start;
                }
            };
            // ---- bounded work per tile: queries too far apart to share candidates (stragglers strung along a brick row,
            // sparse outliers still short of k) would make ONE wave scan thousands of candidates while the chip drains;
            // such tiles -- and passes beyond the third -- hand their open lanes to the lane-per-query shell search
            bool too_big = items64 > TILE_MAX_ROWS || pass >= TILE_MAX_PASSES;
            int start0 = 0, len0 = 0;         // this lane's run of the first round (looked up once, used twice)
            if (!too_big) {
                run_of(lane, start0, len0);
                int cnt_total = len0;
                for (int ib = 64; ib < items; ib += 64) {
                    int st_, ln_;
                    run_of(ib + lane, st_, ln_);
                    cnt_total += ln_;
                }
                too_big = wave_sum_i(cnt_total) > TILE_MAX_CAND;
            }
            PROF_T(t_p1);
            PROF_ADD(1, t_p0, t_p1);
            if (too_big && pass == 0 && gsize > 16) {
                split = true;         // retry with the first half of this group
                break;
            }
            if (too_big) {
                KNN_STAT(4, 1);
                KNN_STAT(6, (items64 > TILE_MAX_ROWS) ? 1 : 0);
                KNN_STAT(7, (pass >= TILE_MAX_PASSES) ? 1 : 0);
                if (pass == 0) {
                    // nothing scanned yet (a 16-lane group whose own box is already too big): lane-per-query shells
                    if (active) shell_search<K>(jb.G, g, qx, qy, qz, k, best);
                } else {
                    // every lane has seen Rold; the open ones finish one at a time with the whole wave on each
                    straggler_search<K>(jb, g, Rold, __ballot(open && active), lane, qx, qy, qz, k, best, cand, rstart, roff);
                }
                PROF_T(t_fb);
                PROF_ADD(5, t_p1, t_fb);
                break;
            }
            int item_base = 0, total = 0, e = 0, fill = 0;
            bool input_done = false;
            for (;;) {
                if (!input_done && e == total) {
                    if (item_base >= items) {
                        input_done = true;
                    } else {
                        // next 64 runs: (start, length) per lane, exclusive offsets by a wave scan
                        int start = start0, len = len0;
                        if (item_base > 0) run_of(item_base + lane, start, len);
                        const int incl = wave_incl_scan_i(len, lane);
                        total = __builtin_amdgcn_readlane(incl, 63);
                        rstart[lane] = start;
                        roff[lane] = incl - len;
                        wave_lds_sync();
                        e = 0;
                        item_base += 64;
                    }
                }
                if (!input_done) {
                    // copy elements [e, e + take) of this round's concatenated runs to cand[fill ..)
                    const int take = min(total - e, TILE_CH - fill);
                    // four elements per lane and trip: the binary searches and the 16-byte gathers of a trip are independent,
                    // so one trip exposes ONE global latency for up to 256 candidates
                    for (int i0 = 0; i0 < take; i0 += 256) {
                        int src[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int i = i0 + 64 * j + lane;
                            const int gi = e + min(i, take - 1);
                            int lo = 0;                  // largest r with roff[r] <= gi (zero-length runs never win)
#pragma unroll
                            for (int st = 32; st > 0; st >>= 1)
                                if (roff[lo + st] <= gi) lo += st;
                            src[j] = rstart[lo] + (gi - roff[lo]);
                        }
                        float4 v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = jb.G.sorted[src[j]];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int i = i0 + 64 * j + lane;
                            if (i < take) cand[fill + i] = v[j];
                        }
                    }
                    fill += take;
                    e += take;
                    if (fill < TILE_CH) continue;
                } else if (fill == 0) {
                    break;
                }
                // ---- consume the staged chunk -------------------------------------------------------------------
                PROF_T(t_c0);
                const int nc = (fill + 15) & ~15;
                if (lane < nc - fill) cand[fill + lane] = make_float4(__uint_as_float(0x7f800000u), __uint_as_float(0x7f800000u),
                                                                       __uint_as_float(0x7f800000u), __int_as_float(PAD_IDX));
                wave_lds_sync();
                KNN_STAT(1, fill);
                if constexpr (K == 1) {
                    for (int c0 = 0; c0 < nc; c0 += 4) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float4 c = cand[c0 + j];
                            best[0] = key_min(best[0], pack_key(dist2_canon(qx, qy, qz, c.x, c.y, c.z), c.w));
                        }
                    }
                } else {
                    for (int c0 = 0; c0 < nc; c0 += 16) {
                        double fresh[16];
                        unsigned dmin = 0xffffffffu;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            const float4 c = cand[c0 + j];
                            const float d2 = dist2_canon(qx, qy, qz, c.x, c.y, c.z);
                            dmin = min(dmin, __float_as_uint(d2));
                            fresh[j] = pack_key(d2, c.w);
                        }
                        // nothing in this block can enter any lane's list: skip the networks (d2 >= 0: uint order)
                        const unsigned kth_hi = (unsigned)((u64)__double_as_longlong(best[K - 1]) >> 32);
                        if (!__any(dmin <= kth_hi)) { KNN_STAT(3, 1); continue; }
                        KNN_STAT(2, 1);
#ifndef KNN_ABL_NONET
                        sort16(fresh);
                        merge16(best, fresh);
#else
#pragma unroll
                        for (int j = 0; j < 16; ++j) best[j] = key_min(best[j], fresh[j]);
#endif
                    }
                }
                wave_lds_sync();      // every lane is done with the chunk before it is refilled
                PROF_T(t_c1);
                PROF_ADD(3, t_c0, t_c1);
                fill = 0;
                if (input_done) break;
            }

            // ---- which lanes are final? --------------------------------------------------------------------------
            PROF_T(t_e0);
            PROF_ADD(2, t_p1, t_e0);          // staging + consumption of this pass (consumption also counted in [3])
            u64 kth = (u64)__double_as_longlong(best[K - 1]);
            if (K > 1 && k < K) {
#pragma unroll
                for (int j = 0; j < K; ++j) if (j == k - 1) kth = (u64)__double_as_longlong(best[j]);
            }
            const bool all = R.xa <= 0 && R.xb >= dxm && R.ya <= 0 && R.yb >= dym && R.za <= 0 && R.zb >= dzm;
            if (all) break;

            float gd = 3.0e38f;
            if (R.xa > 0) gd = fminf(gd, qx - (g.lo[0] + (float)R.xa * g.c));
            if (R.xb < dxm) gd = fminf(gd, (g.lo[0] + (float)(R.xb + 1) * g.c) - qx);
            if (R.ya > 0) gd = fminf(gd, qy - (g.lo[1] + (float)R.ya * g.c));
            if (R.yb < dym) gd = fminf(gd, (g.lo[1] + (float)(R.yb + 1) * g.c) - qy);
            if (R.za > 0) gd = fminf(gd, qz - (g.lo[2] + (float)R.za * g.c));
            if (R.zb < dzm) gd = fminf(gd, (g.lo[2] + (float)(R.zb + 1) * g.c) - qz);
            gd -= g.margin;
            const float dk = __uint_as_float((unsigned)(kth >> 32));       // NaN pattern when the slot is empty
            const bool found = kth != KEY_EMPTY && (unsigned)(kth >> 32) < 0x7f800000u;
            const bool exact = !active || (found && gd > 0.f && dk < gd * gd * 0.999999f);
            open = !exact;
            if (__all(exact)) break;
            // ---- next box: hull of the open lanes' balls (upper bounds), or twice the box for lanes still short of k --
            CellBox W = R;
            if (!exact) {
                if (found) {
                    const float dn = sqrtf(dk) * 1.000001f + 2.0f * g.margin;
                    W.xa = min(W.xa, cell_coord(qx - dn, g.lo[0], g.inv_c, g.dims[0]));
                    W.xb = max(W.xb, cell_coord(qx + dn, g.lo[0], g.inv_c, g.dims[0]));
                    W.ya = min(W.ya, cell_coord(qy - dn, g.lo[1], g.inv_c, g.dims[1]));
                    W.yb = max(W.yb, cell_coord(qy + dn, g.lo[1], g.inv_c, g.dims[1]));
                    W.za = min(W.za, cell_coord(qz - dn, g.lo[2], g.inv_c, g.dims[2]));
                    W.zb = max(W.zb, cell_coord(qz + dn, g.lo[2], g.inv_c, g.dims[2]));
                } else {
                    const int wx = R.xb - R.xa + 1, wy = R.yb - R.ya + 1, wz = R.zb - R.za + 1;
                    W.xa = R.xa - wx; W.xb = R.xb + wx;
                    W.ya = R.ya - wy; W.yb = R.yb + wy;
                    W.za = R.za - wz; W.zb = R.zb + wz;
                }
            }
            CellBox Nx;
            {
                int wl[3] = {W.xa, W.ya, W.za}, wh[3] = {W.xb, W.yb, W.zb};
                wave_minmax3_i(wl, wh);
                Nx.xa = max(wl[0], 0); Nx.xb = min(wh[0], dxm);
                Nx.ya = max(wl[1], 0); Nx.yb = min(wh[1], dym);
                Nx.za = max(wl[2], 0); Nx.zb = min(wh[2], dzm);
            }
            if (Nx.xa == R.xa && Nx.xb == R.xb && Nx.ya == R.ya && Nx.yb == R.yb && Nx.za == R.za && Nx.zb == R.zb) {
                // rounding left the hull where it was: grow by one cell (the box is not the whole grid here)
                Nx.xa = max(R.xa - 1, 0); Nx.xb = min(R.xb + 1, dxm);
                Nx.ya = max(R.ya - 1, 0); Nx.yb = min(R.yb + 1, dym);
                Nx.za = max(R.za - 1, 0); Nx.zb = min(R.zb + 1, dzm);
            }
            Rold = R;
            R = Nx;
            has_old = true;
            PROF_T(t_e1);
            PROF_ADD(4, t_e0, t_e1);
        }
    }
    if (split) {
        gsize >>= 1;
        KNN_STAT(6, 0);
        continue;
    }
    if (valid && active) {
        if (K == 16 && k == 16 && !jb.out_d2 && (((uintptr_t)jb.out_idx) & 15) == 0) {
            // the common case (RandLA pyramid): one 64-byte row per query, four 16-byte stores
            int32_t o[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const u64 key = (u64)__double_as_longlong(best[j < K ? j : 0]);
                const bool ok = key != KEY_EMPTY && (unsigned)(key & 0xffffffffull) != (unsigned)PAD_IDX;
                o[j] = ok ? (int32_t)((int64_t)(unsigned)(key & 0xffffffffull) + out_base) : -1;
            }
            int4* dst = reinterpret_cast<int4*>(jb.out_idx + out_row * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) dst[j] = make_int4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (j < k) {
                    const u64 key = (u64)__double_as_longlong(best[j]);
                    const bool ok = key != KEY_EMPTY && (unsigned)(key & 0xffffffffull) != (unsigned)PAD_IDX;
                    jb.out_idx[out_row * k + j] = ok ? (int32_t)((int64_t)(unsigned)(key & 0xffffffffull) + out_base) : -1;
                    if (jb.out_d2)
                        jb.out_d2[out_row * k + j] = ok ? __uint_as_float((unsigned)(key >> 32)) : __uint_as_float(0x7f800000u);
                }
            }
        }
    }
    gbeg += gsize;
    gsize = min(64 - gbeg, gbeg & -gbeg);        // the sibling of what was just finished (64 -> 32 | 32 -> 16 16 | 32 ...)
    }   // groups
    PROF_T(t_out);
    PROF_ADD(6, t_begin, t_out);
    PROF_ADD(7, t_out - 1, t_out);
    if (K == 16) PROF_FLUSH(tile_g);
}

static int launch_tiles(TileJobs& J, int k, int index_local, hipStream_t stream) {
    unsigned tiles = 0;
    for (int i = 0; i < J.n; ++i) {
        J.j[i].tile_begin = tiles;
        tiles += J.j[i].n_tiles;
    }
    if (tiles == 0) return 0;
    const unsigned blocks = (tiles + 3) / 4;
    if (k == 1) hipLaunchKernelGGL(knn_tile<1>, dim3(blocks), dim3(256), 0, stream, J, k, index_local);
    else if (k <= 16) hipLaunchKernelGGL(knn_tile<16>, dim3(blocks), dim3(256), 0, stream, J, k, index_local);
    else return ML3D_E_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

// tile order for the forward's attention kernels: the brick-sorted (packed, cloud-major) sequence of point rows
__global__ void order_from_qorder(const float4* __restrict__ qorder, int64_t n_total, int64_t n_per_item,
                                  int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    out[i] = (int32_t)((i / n_per_item) * n_per_item + (int64_t)__float_as_int(qorder[i].w));
}

static float tuning_occ() {
    // tuning knob only (speed, never results); read once, at the first call
    static const float v = [] { const char* e = getenv("ML3D_KNN_OCC"); return e ? (float)atof(e) : 0.f; }();
    return v;
}

}  // namespace ml3d

using namespace ml3d;

extern "C" int ml3d_abi_version(void) { return 1; }

extern "C" size_t ml3d_knn_workspace_bytes(int64_t n_points, int64_t n_queries, int64_t batch) {
    return grid_ws_bytes(n_points, batch) + tile_order_ws_bytes(n_queries, batch) + 256;
}

extern "C" int ml3d_knn_search(const float* points, const int64_t* points_row_splits, const float* queries,
                               const int64_t* queries_row_splits, int64_t batch, int64_t n_points,
                               int64_t n_queries, int k, int index_local, int32_t* out_index,
                               float* out_dist2, void* workspace, size_t workspace_bytes, void* stream) {
    if (!points_row_splits || !queries_row_splits || batch <= 0 || k <= 0 || n_points < 0 || n_queries < 0 ||
        n_points > 0x7fffffffll / GRID_CAP - 4096 || n_queries > 0x7fffffffll / GRID_CAP - 4096)
        return ML3D_E_INVALID;
    if (k > 64) return ML3D_E_UNSUPPORTED;
    if (n_queries == 0) return 0;
    if (!out_index || (n_points > 0 && !points) || !queries) return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_knn_workspace_bytes(n_points, n_queries, batch)) return ML3D_E_WORKSPACE;
    GridWs ws;
    const size_t gbytes = grid_ws_bytes(n_points, batch);
    if (!grid_ws_carve(workspace, gbytes, n_points, batch, &ws)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Segs ps = {points_row_splits, 0, 0, (int)batch};
    Segs qs = {queries_row_splits, 0, 0, (int)batch};
    int rc = grid_build(points, ps, ws, tuning_occ(), st);
    if (rc) return ML3D_E_LAUNCH;
    GridView G = grid_view(ws);
    if (k > 16) {
        // long lists: lane-per-query shell search (the register-resident sorting networks of the tile kernel stop at 16)
        QuerySrc Q;
        bool self = (queries == points) && (queries_row_splits == points_row_splits) && (n_queries == n_points);
        Q.sorted_q = self ? ws.sorted : nullptr;
        Q.qsegs = ws.segs;
        Q.raw = queries;
        Q.segs = qs;
        Q.n_total = n_queries;
        return launch_query(G, Q, k, index_local, ps, out_index, out_dist2, st);
    }
    TileOrderWs tw;
    if (!tile_order_ws_carve((char*)workspace + gbytes, workspace_bytes - gbytes, n_queries, batch, &tw))
        return ML3D_E_WORKSPACE;
    const bool self = (queries == points) && (queries_row_splits == points_row_splits) && (n_queries == n_points);
    if (self ? tile_order_build_sorted(ws, qs, tw, st) : tile_order_build(queries, qs, ws.segs, tw, st)) return ML3D_E_LAUNCH;
    TileJobs J;
    J.n = 1;
    TileJob& a = J.j[0];
    a.G = G; a.qorder = tw.qorder; a.tile_splits = tw.tile_splits; a.qsegs = qs; a.support = ps;
    a.out_idx = out_index; a.out_d2 = out_dist2; a.tile_begin = 0;
    a.n_tiles = (unsigned)tile_count_bound(n_queries, batch);
    return launch_tiles(J, k, index_local, st);
}

static int pyramid_sizes(int64_t n0, int num_layers, const int32_t* ratios, int64_t* n /* L+1 */) {
    n[0] = n0;
    for (int l = 0; l < num_layers; ++l) {
        if (ratios[l] <= 0) return -1;
        n[l + 1] = n[l] / ratios[l];
    }
    return 0;
}

extern "C" size_t ml3d_randla_pyramid_workspace_bytes(int64_t batch, int64_t n0, int num_layers,
                                                      const int32_t* ratios_host) {
    if (num_layers <= 0 || num_layers > 15 || !ratios_host) return 0;
    int64_t n[17];
    if (pyramid_sizes(n0, num_layers, ratios_host, n)) return 0;
    size_t b = 0;
    for (int l = 0; l <= num_layers; ++l) b += grid_ws_bytes(n[l] * batch, batch) + 256;
    for (int l = 0; l < num_layers; ++l) b += tile_order_ws_bytes(n[l] * batch, batch) + 256;
    return b;
}

extern "C" int ml3d_randla_knn_pyramid(const float* points, int64_t batch, int64_t n0, int num_layers,
                                       const int32_t* ratios_host, int k, int32_t* const* neighbor_idx_host,
                                       int32_t* const* interp_idx_host, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    return ml3d_randla_knn_pyramid_traced(points, batch, n0, num_layers, ratios_host, k, neighbor_idx_host,
                                          interp_idx_host, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int ml3d_randla_knn_pyramid_traced(const float* points, int64_t batch, int64_t n0, int num_layers,
                                              const int32_t* ratios_host, int k,
                                              int32_t* const* neighbor_idx_host, int32_t* const* interp_idx_host,
                                              void* workspace, size_t workspace_bytes, void* stream,
                                              const ml3d_trace* tr) {
    return ml3d_randla_knn_pyramid_ordered(points, batch, n0, num_layers, ratios_host, k, neighbor_idx_host,
                                           interp_idx_host, nullptr, workspace, workspace_bytes, stream, tr);
}

extern "C" int ml3d_randla_knn_pyramid_ordered(const float* points, int64_t batch, int64_t n0, int num_layers,
                                               const int32_t* ratios_host, int k,
                                               int32_t* const* neighbor_idx_host, int32_t* const* interp_idx_host,
                                               int32_t* const* tile_order_host, void* workspace,
                                               size_t workspace_bytes, void* stream, const ml3d_trace* tr) {
    auto tb = [&](int tag) { if (tr && tr->tag == tag && tr->ev_start) (void)hipEventRecord((hipEvent_t)tr->ev_start, (hipStream_t)stream); };
    auto te = [&](int tag) { if (tr && tr->tag == tag && tr->ev_stop) (void)hipEventRecord((hipEvent_t)tr->ev_stop, (hipStream_t)stream); };
    if (!points || batch <= 0 || n0 <= 0 || num_layers <= 0 || num_layers > 15 || !ratios_host || k <= 0 ||
        !neighbor_idx_host || !interp_idx_host)
        return ML3D_E_INVALID;
    if (k > 64) return ML3D_E_UNSUPPORTED;
    int64_t n[17];
    if (pyramid_sizes(n0, num_layers, ratios_host, n)) return ML3D_E_INVALID;
    if (n0 * batch > 0x7fffffffll / GRID_CAP - 4096) return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_randla_pyramid_workspace_bytes(batch, n0, num_layers, ratios_host))
        return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    GridWs ws[17];
    TileOrderWs tw[16];
    char* p = (char*)workspace;
    for (int l = 0; l <= num_layers; ++l) {
        size_t bytes = grid_ws_bytes(n[l] * batch, batch) + 256;
        if (!grid_ws_carve(p, bytes, n[l] * batch, batch, &ws[l])) return ML3D_E_WORKSPACE;
        p += bytes;
    }
    for (int l = 0; l < num_layers; ++l) {
        size_t bytes = tile_order_ws_bytes(n[l] * batch, batch) + 256;
        if (!tile_order_ws_carve(p, bytes, n[l] * batch, batch, &tw[l])) return ML3D_E_WORKSPACE;
        p += bytes;
    }
    float occ = tuning_occ();
    // grids of every level: level l = prefix [:n_l] of each cloud (randlanet.py:222); plus the level's tile order
    for (int l = 0; l <= num_layers; ++l) {
        if (n[l] == 0) continue;
        Segs S = {nullptr, n0, n[l], (int)batch};
        tb(100 + l);
        // level 0 probes the cloud; the thinner prefix levels reuse its box and dimension estimate
        if (l == 0 ? grid_build(points, S, ws[l], occ, st) : grid_build_derived(points, S, ws[l], ws[0], st))
            return ML3D_E_LAUNCH;
        if (l < num_layers) {
            if (tile_order_build_sorted(ws[l], S, tw[l], st)) return ML3D_E_LAUNCH;
            if (tile_order_host && tile_order_host[l]) {
                const int64_t nt = n[l] * batch;
                hipLaunchKernelGGL(order_from_qorder, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, tw[l].qorder, nt,
                                   n[l], tile_order_host[l]);
                if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
            }
        }
        te(100 + l);
    }
    // all k-NN searches (level l onto itself, randlanet.py:220) in one launch, all 1-NN interpolation
    // searches (level l in level l+1, randlanet.py:224) in a second one
    if (num_layers > KNN_MAX_JOBS) return ML3D_E_UNSUPPORTED;
    TileJobs Jk, J1;
    Jk.n = 0; J1.n = 0;
    for (int l = 0; l < num_layers; ++l) {
        if (n[l] == 0) continue;
        Segs S = {nullptr, n0, n[l], (int)batch};
        const unsigned nt = (unsigned)(((n[l] + 63) / 64) * batch);
        TileJob& a = Jk.j[Jk.n++];
        a.G = grid_view(ws[l]); a.qorder = tw[l].qorder; a.tile_splits = nullptr; a.qsegs = S; a.support = S;
        a.out_idx = neighbor_idx_host[l]; a.out_d2 = nullptr; a.tile_begin = 0; a.n_tiles = nt;
        if (n[l + 1] > 0) {
            Segs S1 = {nullptr, n0, n[l + 1], (int)batch};
            TileJob& c = J1.j[J1.n++];
            c.G = grid_view(ws[l + 1]); c.qorder = tw[l].qorder; c.tile_splits = nullptr; c.qsegs = S; c.support = S1;
            c.out_idx = interp_idx_host[l]; c.out_d2 = nullptr; c.tile_begin = 0; c.n_tiles = nt;
        } else {
            (void)hipMemsetAsync(interp_idx_host[l], 0xff, sizeof(int32_t) * (size_t)(n[l] * batch), st);
        }
    }
    tb(0);
    int rc = launch_tiles(Jk, k, 1, st);
    te(0);
    if (rc) return rc;
    tb(1);
    rc = launch_tiles(J1, 1, 1, st);
    te(1);
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// Patch sampler support: the num_points nearest points to a centre
//   search_tree.query(center_point, k=num_points)   ml3d/datasets/samplers/semseg_spatially_regular.py:90-91
// and the test-time vote accumulation
//   test_probs[inds] = smooth * test_probs[inds] + (1 - smooth) * softmax(logits)
//   ml3d/torch/models/randlanet.py:420-421, 457-462 (float16 accumulator, numpy promotion rules).
// k is the patch size (45 056), far beyond a register-resident best-k list: the whole cloud is keyed by
// (d2, index) and radix-sorted (sort.hip), the first k entries are the answer in canonical order.
// ---------------------------------------------------------------------------------------------------
#include <hip/hip_fp16.h>

#include "sort.h"

namespace ml3d {

__global__ void center_keys(const float* __restrict__ pts, int64_t n, float cx, float cy, float cz, u64* __restrict__ keys,
                            uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d2 = dist2_canon(cx, cy, cz, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    keys[i] = ((u64)__float_as_uint(d2) << 32) | (u64)(uint32_t)i;
    vals[i] = (uint32_t)i;
}

__global__ void center_take(const u64* __restrict__ keys, int64_t k, int32_t* __restrict__ out_idx, float* __restrict__ out_d2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const u64 key = keys[i];
    out_idx[i] = (int32_t)(key & 0xffffffffull);
    if (out_d2) out_d2[i] = __uint_as_float((unsigned)(key >> 32));
}

// one wave per patch point: softmax over the C classes (lanes stride the classes), then the float16 vote update
__global__ void __launch_bounds__(256)
vote_update(const float* __restrict__ logits, const int32_t* __restrict__ inds, int64_t n, int C, float smooth,
            __half* __restrict__ probs, int64_t n_cloud) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* row = logits + i * C;
    float mx = -3.0e38f;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(row[c] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const int64_t dst = inds[i];
    if (dst < 0 || dst >= n_cloud) return;
    const __half hs = __float2half(smooth);                     // numpy: python float * float16 array -> float16
    const float w_new = 1.0f - smooth;
    for (int c = lane; c < C; c += 64) {
        const float p = expf(row[c] - mx) / sum;
        const __half old = probs[dst * C + c];
        const __half keep = __float2half(__half2float(hs) * __half2float(old));     // float16 product, one rounding
        probs[dst * C + c] = __float2half(__half2float(keep) + w_new * p);          // float32 sum -> float16 store
    }
}

}  // namespace ml3d

extern "C" size_t ml3d_nearest_to_center_workspace_bytes(int64_t n_points) {
    if (n_points < 0) return 0;
    const int64_t m = n_points > 0 ? n_points : 1;
    return ((sizeof(u64) * (size_t)m + 255) & ~(size_t)255) + ((sizeof(uint32_t) * (size_t)m + 255) & ~(size_t)255) +
           sort_ws_bytes(m) + 512;
}

extern "C" int ml3d_nearest_to_center(const float* points, int64_t n_points, const float* center_host, int64_t k,
                                      int32_t* out_index, float* out_dist2, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    if (n_points < 0 || k < 0 || k > n_points || !center_host || n_points > 0x7ffffff0ll) return ML3D_E_INVALID;
    if (k == 0) return 0;
    if (!points || !out_index) return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_nearest_to_center_workspace_bytes(n_points)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    u64* keys = (u64*)p;            p += (sizeof(u64) * (size_t)n_points + 255) & ~(size_t)255;
    uint32_t* vals = (uint32_t*)p;  p += (sizeof(uint32_t) * (size_t)n_points + 255) & ~(size_t)255;
    SortWs sw;
    if (!sort_ws_carve(p, sort_ws_bytes(n_points), n_points, &sw)) return ML3D_E_WORKSPACE;
    hipLaunchKernelGGL(center_keys, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, st, points, n_points,
                       center_host[0], center_host[1], center_host[2], keys, vals);
    if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    if (sort_pairs_u64(keys, vals, n_points, 64, sw, st)) return ML3D_E_LAUNCH;
    hipLaunchKernelGGL(center_take, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, keys, k, out_index, out_dist2);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_vote_update(const float* logits, const int32_t* point_inds, int64_t n, int num_classes, float smooth,
                                void* probs_f16, int64_t n_cloud, void* stream) {
    if (n < 0 || num_classes <= 0 || n_cloud < 0) return ML3D_E_INVALID;
    if (n == 0) return 0;
    if (!logits || !point_inds || !probs_f16) return ML3D_E_INVALID;
    hipLaunchKernelGGL(vote_update, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, point_inds, n,
                       num_classes, smooth, (__half*)probs_f16, n_cloud);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}
