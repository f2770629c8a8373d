// voxel.hip — voxelize (PointPillars) and grid subsample (KPConv / RandLA-Net preprocessing), gfx950.
//
// voxelize  replaces open3d.ml.torch.ops.voxelize as called from PointPillarsVoxelization.forward
//           (ml3d/torch/models/point_pillars.py:354-357);
// subsample replaces open3d.ml.contrib.subsample / subsample_batch as called from
//           DataProcessing.grid_subsampling (ml3d/datasets/utils/dataprocessing.py:14-49) and
//           batch_grid_subsampling (ml3d/torch/models/kpconv.py:2037-2164).
//
// Both group points by an integer voxel key.  A hash table would make the order of voxels and of
// the points inside a voxel depend on atomics; the oracle's canonical order is ascending voxel key
// with ORIGINAL point order inside a voxel, and the subsample barycentres are float32 sums in that
// order, so the grouping is a stable LSD radix sort of (key, point index) pairs (sort.hip) followed
// by head flags, an int32 scan and one thread per voxel walking its run.  Everything is an HBM stream:
// algorithmic bytes per point 12 (xyz read) + 12 * passes (sort) + 4..16 (outputs).
// Two-phase ragged results (count -> caller allocates -> fill); the workspace carries the sorted pairs
// between the two calls.  No allocation, no global state.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "grid.h"
#include "ml3d_hip.h"
#include "sort.h"

namespace ml3d {

static inline size_t vx_align(size_t x) { return (x + 255) & ~(size_t)255; }

// workspace shared by voxelize and subsample
struct GroupWs {
    u64* keys;          // [n]
    uint32_t* vals;     // [n]
    int* flags;         // [n + 1] head flags -> inclusive scan (flags[0] = 0)
    int* hp;            // [n + 1] first sorted slot of voxel ordinal v; hp[n_vox] = number of valid pairs
    int* cnt;           // [n + 1] kept points per voxel ordinal -> inclusive scan (cnt[0] = 0)
    int* fv;            // [batch + 1] first voxel ordinal of a batch item
    int* block_sums;    // scan scratch
    unsigned* bbox;     // [batch][6]  (subsample)
    unsigned* occ;      // [batch][GRID_LEVELS] scratch of bbox_compute
    float* seg;         // [batch][8]  org[3], then G[3] as int bits, pad (subsample)
    SortWs sort;
    FusedWs fused;      // hand-off tables of the fused voxelize (sort.h)
    int64_t n;
    int batch;
};

static size_t group_ws_bytes(int64_t n, int64_t batch) {
    int64_t m = n > 0 ? n : 1;
    size_t b = 0;
    b += vx_align(sizeof(u64) * (size_t)m);
    b += vx_align(sizeof(uint32_t) * (size_t)m);
    b += 3 * vx_align(sizeof(int) * (size_t)(m + 2));
    b += vx_align(sizeof(int) * (size_t)(batch + 2));
    b += vx_align(sizeof(int) * (size_t)((m + 1 + 1023) / 1024 + 2));
    b += vx_align(sizeof(unsigned) * 6 * (size_t)batch);
    b += vx_align(sizeof(unsigned) * GRID_LEVELS * (size_t)batch);
    b += vx_align(sizeof(float) * 8 * (size_t)batch);
    b += vx_align(sort_ws_bytes(m));
    b += fused_ws_bytes(m, batch);
    return b + 256;
}

static bool group_ws_carve(void* ws, size_t bytes, int64_t n, int64_t batch, GroupWs* o) {
    if (bytes < group_ws_bytes(n, batch)) return false;
    int64_t m = n > 0 ? n : 1;
    char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    o->keys = (u64*)p;          p += vx_align(sizeof(u64) * (size_t)m);
    o->vals = (uint32_t*)p;     p += vx_align(sizeof(uint32_t) * (size_t)m);
    o->flags = (int*)p;         p += vx_align(sizeof(int) * (size_t)(m + 2));
    o->hp = (int*)p;            p += vx_align(sizeof(int) * (size_t)(m + 2));
    o->cnt = (int*)p;           p += vx_align(sizeof(int) * (size_t)(m + 2));
    o->fv = (int*)p;            p += vx_align(sizeof(int) * (size_t)(batch + 2));
    o->block_sums = (int*)p;    p += vx_align(sizeof(int) * (size_t)((m + 1 + 1023) / 1024 + 2));
    o->bbox = (unsigned*)p;     p += vx_align(sizeof(unsigned) * 6 * (size_t)batch);
    o->occ = (unsigned*)p;      p += vx_align(sizeof(unsigned) * GRID_LEVELS * (size_t)batch);
    o->seg = (float*)p;         p += vx_align(sizeof(float) * 8 * (size_t)batch);
    size_t sb = sort_ws_bytes(m);
    if (!sort_ws_carve(p, sb, m, &o->sort)) return false;
    p += vx_align(sb);
    if (!fused_ws_carve(p, fused_ws_bytes(m, batch), m, batch, &o->fused)) return false;
    o->n = n;
    o->batch = (int)batch;
    return true;
}

static int bits_for(unsigned long long v) {   // bits needed to represent every value in [0, v]
    int b = 1;
    while (b < 64 && (v >> b) != 0ull) ++b;
    return b;
}

// ---- shared grouping kernels -----------------------------------------------------------------------
// flags[i + 1] = 1 iff sorted pair i opens a voxel (valid key differing from its predecessor)
__global__ void group_heads(const u64* __restrict__ keys, int64_t n, u64 key_invalid, int* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) flags[0] = 0;
    if (i >= n) return;
    const u64 k = keys[i];
    flags[i + 1] = (k != key_invalid && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// after the scan: hp[v] = slot of the head of voxel ordinal v; hp[n_vox] = number of valid pairs
__global__ void group_headpos(const u64* __restrict__ keys, int64_t n, u64 key_invalid, const int* __restrict__ flags,
                              int* __restrict__ hp) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = keys[i];
    const bool valid = k != key_invalid;
    if (valid && (i == 0 || keys[i - 1] != k)) hp[flags[i + 1] - 1] = (int)i;
    const int nv = flags[n];
    if (!valid && (i == 0 || keys[i - 1] != key_invalid)) hp[nv] = (int)i;   // first invalid pair
    if (i == n - 1 && valid) hp[nv] = (int)n;
}

__device__ __forceinline__ int64_t lower_bound_u64(const u64* a, int64_t n, u64 v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// fv[b] = number of voxels whose key is below the first key of batch item b
__global__ void group_first_voxel(const u64* __restrict__ keys, int64_t n, const int* __restrict__ flags, int batch,
                                  u64 item_stride, int item_shift, int* __restrict__ fv) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > batch) return;
    if (b == batch) { fv[b] = n > 0 ? flags[n] : 0; return; }
    const u64 first = item_shift >= 0 ? ((u64)b << item_shift) : (u64)b * item_stride;
    fv[b] = n > 0 ? flags[lower_bound_u64(keys, n, first)] : 0;
}

// ---- voxelize ------------------------------------------------------------------------------------------
struct VoxParams {
    float vs[3], rmin[3], rmax[3];
    long long G[3];
    long long cells;        // G0 * G1 * G2
    long long max_points, max_voxels;
};

__global__ void vox_keys(const float* __restrict__ pts, int64_t stride, Segs S, int64_t n, VoxParams P,
                         u64* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s; int64_t local;
    seg_locate(S, i, s, local);
    const float* p = pts + stride * i;
    bool ok = true;
    long long c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = p[a];
        ok = ok && (v >= P.rmin[a]) && (v <= P.rmax[a]);
        c[a] = (long long)__fdiv_rn(__fsub_rn(v, P.rmin[a]), P.vs[a]);
    }
    const u64 inval = (u64)S.batch * (u64)P.cells;
    keys[i] = ok ? (u64)s * (u64)P.cells + (u64)(c[0] + P.G[0] * (c[1] + P.G[1] * c[2])) : inval;
    vals[i] = (uint32_t)i;
}

// kept points per voxel ordinal (0 for voxels beyond max_voxels of their batch item)
__global__ void vox_counts(const u64* __restrict__ keys, const int* __restrict__ flags, int64_t n,
                           const int* __restrict__ hp, const int* __restrict__ fv, VoxParams P, int* __restrict__ cnt) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v == 0) cnt[0] = 0;
    if (v >= n) return;
    const int nv = flags[n];
    if (v >= nv) { cnt[v + 1] = 0; return; }
    const int h = hp[v];
    const int b = (int)(keys[h] / (u64)P.cells);
    const long long rank = v - fv[b];
    long long c = hp[v + 1] - h;
    if (c > P.max_points) c = P.max_points;
    cnt[v + 1] = rank < P.max_voxels ? (int)c : 0;
}

__global__ void vox_batch_splits(const int* __restrict__ fv, int batch, long long max_voxels, const int* __restrict__ cnt,
                                 const int* __restrict__ flags, int64_t n, int64_t* __restrict__ batch_splits,
                                 int64_t* __restrict__ stats) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    int64_t acc = 0;
    batch_splits[0] = 0;
    for (int b = 0; b < batch; ++b) {
        long long nvb = fv[b + 1] - fv[b];
        acc += nvb < max_voxels ? nvb : max_voxels;
        batch_splits[b + 1] = acc;
    }
    stats[0] = acc;
    stats[1] = n > 0 ? (int64_t)cnt[flags[n]] : 0;
}

__global__ void vox_fill(const u64* __restrict__ keys, const uint32_t* __restrict__ vals, const int* __restrict__ flags,
                         int64_t n, const int* __restrict__ hp, const int* __restrict__ fv, const int* __restrict__ cnt,
                         const int64_t* __restrict__ batch_splits, VoxParams P, int32_t* __restrict__ coords,
                         int64_t* __restrict__ pidx, int64_t* __restrict__ prs) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v == 0) prs[0] = 0;
    if (v >= n || v >= flags[n]) return;
    const int h = hp[v];
    const u64 key = keys[h];
    const int b = (int)(key / (u64)P.cells);
    const long long rank = v - fv[b];
    if (rank >= P.max_voxels) return;
    const int64_t ov = batch_splits[b] + rank;
    const long long lin = (long long)(key - (u64)b * (u64)P.cells);
    coords[3 * ov + 0] = (int32_t)(lin % P.G[0]);
    coords[3 * ov + 1] = (int32_t)((lin / P.G[0]) % P.G[1]);
    coords[3 * ov + 2] = (int32_t)(lin / (P.G[0] * P.G[1]));
    const int64_t o = cnt[v];
    const int c = cnt[v + 1] - cnt[v];
    prs[ov + 1] = cnt[v + 1];
    for (int j = 0; j < c; ++j) pidx[o + j] = (int64_t)vals[h + j];
}

// ---- voxelize, fused form (round 6): 7 launches for the count phase instead of 27 -----------------------------------------------------
// keys (32 bits: item * cells + cell fits for every PointPillars grid: 16 KITTI sweeps are 3.4 M cells) + the digit histograms of all
// passes -> one launch per radix pass (sort.h: fs_pass) -> ONE grouping launch that does what group_heads / scan / group_headpos /
// group_first_voxel / vox_counts / scan / vox_batch_splits did in eleven.  Results are the same arrays (hp, cnt, fv, batch_splits,
// stats), bit for bit.
__global__ void __launch_bounds__(256)
vox_keys32(const float* __restrict__ pts, int64_t stride, Segs S, int64_t n, VoxParams P, uint32_t inval, int passes,
           uint32_t* __restrict__ keys, int* __restrict__ ghist) {
    __shared__ int h[4][256];
    const int t = threadIdx.x, lane = t & 63;
    h[0][t] = 0; h[1][t] = 0; h[2][t] = 0; h[3][t] = 0;
    __syncthreads();
    for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < n; b0 += (int64_t)gridDim.x * 256) {    // (block-uniform trip count: ballots below)
        const int64_t i = b0 + t;
        const bool act = i < n;
        uint32_t key = inval;
        if (act) {
            int s; int64_t local;
            seg_locate(S, i, s, local);
            const float* p = pts + stride * i;
            bool ok = true;
            long long c[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float v = p[a];
                ok = ok && (v >= P.rmin[a]) && (v <= P.rmax[a]);
                c[a] = (long long)__fdiv_rn(__fsub_rn(v, P.rmin[a]), P.vs[a]);
            }
            if (ok) key = (uint32_t)((u64)s * (u64)P.cells + (u64)(c[0] + P.G[0] * (c[1] + P.G[1] * c[2])));
            keys[i] = key;
        }
        // out-of-range points (half of a KITTI sweep) all carry `inval`: one LDS add per wave and pass instead of a 64-way conflict
        const bool inv = act && key == inval;
        const u64 m = __ballot(inv);
        const int lead = m ? __ffsll((long long)m) - 1 : -1;
        for (int q = 0; q < passes; ++q) {
            if (act && !inv) atomicAdd(&h[q][(key >> (8 * q)) & 255u], 1);
            else if (lane == lead) atomicAdd(&h[q][(inval >> (8 * q)) & 255u], __popcll(m));
        }
    }
    __syncthreads();
    for (int q = 0; q < passes; ++q)
        if (h[q][t]) atomicAdd(&ghist[256 * q + t], h[q][t]);
}

// what the scans of the grouping kernel carry along the sorted keys: h = voxel heads so far, r = heads since (and with) the last
// head that opened a new batch item (f: there was one), p = 1 + position of the last head.  Associative, not commutative.
struct GrpRec { int h, r, f, p; };
__device__ __forceinline__ GrpRec grp_combine(const GrpRec& a, const GrpRec& b) {     // a in front of b
    GrpRec o;
    o.h = a.h + b.h;
    o.r = b.f ? b.r : a.r + b.r;
    o.f = a.f | b.f;
    o.p = a.p > b.p ? a.p : b.p;
    return o;
}
__device__ __forceinline__ GrpRec grp_wave_inclusive(GrpRec v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        GrpRec u;
        u.h = __shfl_up(v.h, o); u.r = __shfl_up(v.r, o); u.f = __shfl_up(v.f, o); u.p = __shfl_up(v.p, o);
        if (lane >= o) v = grp_combine(u, v);
    }
    return v;
}
__device__ __forceinline__ void grp_store(uint32_t* p, const GrpRec& v) {
    st_agent(p + 0, fs_word(1u, (uint32_t)v.h));
    st_agent(p + 1, fs_word(1u, (uint32_t)v.r | ((uint32_t)v.f << 28)));
    st_agent(p + 2, fs_word(1u, (uint32_t)v.p));
}
__device__ __forceinline__ GrpRec grp_load_wait(const uint32_t* p) {
    GrpRec v;
    v.h = (int)fs_wait(p + 0, 1u);
    const uint32_t w = fs_wait(p + 1, 1u);
    v.r = (int)(w & ((1u << 28) - 1u));
    v.f = (int)(w >> 28) & 1;
    v.p = (int)fs_wait(p + 2, 1u);
    return v;
}
// exclusive prefix of `mine` (the tile's aggregate, valid in thread 0) over the tiles in front: the flat two-level hand-off of sort.h
// with records instead of digit counts, evaluated by wave 0 (lane = tile of the group / group), handed to the block through LDS
__device__ __forceinline__ GrpRec grp_tile_prefix(int tile, int tiles, const GrpRec& mine, uint32_t* tA, uint32_t* gA, uint32_t* gcnt,
                                                  int* s_flag, GrpRec* s_rec) {
    const int t = threadIdx.x, lane = t & 63;
    const int g = tile / FS_GROUP, gfirst = g * FS_GROUP;
    const int members = min(FS_GROUP, tiles - gfirst);
    if (t == 0) {
        grp_store(tA + 4 * (size_t)tile, mine);
        *s_flag = atomicAdd(&gcnt[g], 1u) == (uint32_t)(members - 1) ? 1 : 0;
    }
    __syncthreads();
    if (t < 64) {
        const GrpRec zero = {0, 0, 0, 0};
        if (*s_flag) {                                  // this tile arrived last in its group: it publishes the group's record
            GrpRec v = zero;
            if (lane < members) v = grp_load_wait(tA + 4 * (size_t)(gfirst + lane));
            v = grp_wave_inclusive(v, lane);
            if (lane == 63) grp_store(gA + 4 * (size_t)g, v);
        }
        GrpRec a = zero;
        if (lane < g) a = grp_load_wait(gA + 4 * (size_t)lane);
        a = grp_wave_inclusive(a, lane);
        GrpRec b = zero;
        if (lane < tile - gfirst) b = grp_load_wait(tA + 4 * (size_t)(gfirst + lane));
        b = grp_wave_inclusive(b, lane);
        if (lane == 63) *s_rec = grp_combine(a, b);
    }
    __syncthreads();
    const GrpRec out = *s_rec;
    __syncthreads();
    return out;
}

struct GrpArgs {
    const uint32_t* keys;      // sorted ascending, `inval` (= batch * cells) behind every real key
    int64_t n;
    uint32_t inval, cells;
    int batch;
    long long max_points, max_voxels;
    int *hp, *cnt, *fv;
    int64_t *batch_splits, *stats;
    uint32_t *ticket, *gcnt1, *gcnt2, *tA, *gA, *tC, *gC, *fvf;
};

// Thread t of a tile owns the 8 consecutive sorted positions [2048 tile + 8 t, + 8).  head: a real key differing from its predecessor;
// end: a real key differing from its successor; a head whose batch item differs from its predecessor's opens that item (and every
// empty item in between).  At a head: hp[ordinal] = position.  At an end: the voxel's kept points = min(length, max_points), 0 beyond
// max_voxels of its item -> a second prefix gives cnt[ordinal + 1].  The thread that sees the LAST real key closes the open items,
// and its tile -- which by then has waited for nothing but tiles in front of it -- turns the first-voxel table into batch_splits / stats.
__global__ void __launch_bounds__(256) vox_group32(GrpArgs A) {
    __shared__ GrpRec s_wave[4];
    __shared__ GrpRec s_rec;
    __shared__ int s_flag, s_tile, s_fin, s_nv, s_total;
    __shared__ long long s_part[256];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int tiles = (int)gridDim.x;
    if (t == 0) { s_tile = (int)atomicAdd(A.ticket, 1u); s_fin = 0; }
    __syncthreads();
    const int tile = s_tile;
    const int64_t p0 = (int64_t)tile * FS_TILE + 8 * t;
    uint32_t k[8];
    if (p0 + 8 <= A.n) {
        const uint4 a = *reinterpret_cast<const uint4*>(A.keys + p0), b = *reinterpret_cast<const uint4*>(A.keys + p0 + 4);
        k[0] = a.x; k[1] = a.y; k[2] = a.z; k[3] = a.w; k[4] = b.x; k[5] = b.y; k[6] = b.z; k[7] = b.w;
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) k[e] = p0 + e < A.n ? A.keys[p0 + e] : A.inval;
    }
    const uint32_t kprev = p0 > 0 && p0 - 1 < A.n ? A.keys[p0 - 1] : A.inval;          // (position 0: no predecessor, see `head` below)
    const uint32_t knext = p0 + 8 < A.n ? A.keys[p0 + 8] : A.inval;
    unsigned heads = 0u, iheads = 0u, ends = 0u;
    GrpRec mine = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t prev = e ? k[e - 1] : kprev, next = e < 7 ? k[e + 1] : knext;
        const bool valid = k[e] != A.inval;
        const bool first = p0 + e == 0;
        const bool head = valid && (first || prev != k[e]);
        bool ihead = false;
        if (head) ihead = first || prev / A.cells != k[e] / A.cells;
        heads |= (head ? 1u : 0u) << e;
        iheads |= (ihead ? 1u : 0u) << e;
        ends |= (valid && next != k[e] ? 1u : 0u) << e;
        if (head) { mine.h += 1; mine.r = ihead ? 1 : mine.r + 1; mine.f |= ihead ? 1 : 0; mine.p = (int)(p0 + e) + 1; }
    }
    // ---- stage 1: (h, r, p) in front of every thread ----
    const GrpRec zero = {0, 0, 0, 0};
    GrpRec inc = grp_wave_inclusive(mine, lane);
    if (lane == 63) s_wave[w] = inc;
    GrpRec exc;
    exc.h = __shfl_up(inc.h, 1); exc.r = __shfl_up(inc.r, 1); exc.f = __shfl_up(inc.f, 1); exc.p = __shfl_up(inc.p, 1);
    if (lane == 0) exc = zero;
    __syncthreads();
    GrpRec front = zero;
    for (int w2 = 0; w2 < w; ++w2) front = grp_combine(front, s_wave[w2]);
    GrpRec whole = grp_combine(grp_combine(s_wave[0], s_wave[1]), grp_combine(s_wave[2], s_wave[3]));
    const GrpRec tile_front = grp_tile_prefix(tile, tiles, whole, A.tA, A.gA, A.gcnt1, &s_flag, &s_rec);
    GrpRec cur = grp_combine(tile_front, grp_combine(front, exc));
    int ce[8], ve[8];
    int csum = 0;
    bool fin = false;
    int fin_v = -1, fin_item = -1, fin_valid = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int64_t i = p0 + e;
        ce[e] = 0; ve[e] = 0;
        if ((heads >> e) & 1u) {
            const bool ih = (iheads >> e) & 1u;
            cur.h += 1; cur.r = ih ? 1 : cur.r + 1; cur.p = (int)i + 1;
            A.hp[cur.h - 1] = (int)i;
            if (ih) {
                const uint32_t prev = e ? k[e - 1] : kprev;
                const int b1 = (int)(k[e] / A.cells), b0 = i == 0 ? -1 : (int)(prev / A.cells);
                for (int b = b0 + 1; b <= b1; ++b) st_agent(&A.fvf[b], fs_word(1u, (uint32_t)(cur.h - 1)));
            }
        }
        if ((ends >> e) & 1u) {
            const int v = cur.h - 1, rank = cur.r - 1;
            const long long len = (long long)i - (long long)cur.p + 2;
            const int c = (long long)rank < A.max_voxels ? (int)(len < A.max_points ? len : A.max_points) : 0;
            ce[e] = c; ve[e] = v; csum += c;
            const uint32_t next = e < 7 ? k[e + 1] : knext;
            if (next == A.inval) { fin = true; fin_v = v; fin_item = (int)(k[e] / A.cells); fin_valid = (int)i + 1; }
        }
    }
    if (p0 == 0 && k[0] == A.inval) fin = true;                                      // no real key at all: nv = 0
    // ---- stage 2: kept points in front of every thread ----
    const int cinc = wave_inclusive_scan(csum);
    if (lane == 63) s_wave[w].h = cinc;
    __syncthreads();
    int cfront = cinc - csum;
    for (int w2 = 0; w2 < w; ++w2) cfront += s_wave[w2].h;
    GrpRec ctile = zero;
    ctile.h = s_wave[0].h + s_wave[1].h + s_wave[2].h + s_wave[3].h;
    __syncthreads();
    const GrpRec ctile_front = grp_tile_prefix(tile, tiles, ctile, A.tC, A.gC, A.gcnt2, &s_flag, &s_rec);
    int C = ctile_front.h + cfront;
    if (p0 == 0) A.cnt[0] = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if ((ends >> e) & 1u) { C += ce[e]; A.cnt[ve[e] + 1] = C; }
    if (fin) {
        const int nv = fin_v + 1;
        A.hp[nv] = fin_valid;                                                         // number of real keys
        for (int b = fin_item + 1; b <= A.batch; ++b) st_agent(&A.fvf[b], fs_word(1u, (uint32_t)nv));
        s_fin = 1; s_nv = nv; s_total = C;
    }
    __syncthreads();
    if (!s_fin) return;
    // ---- the tile of the last real key: first-voxel table -> fv, batch_splits, stats (every entry was written by a tile in front
    //      of this one or by this one: the waits below cannot block) ----
    const int per = (A.batch + 255) / 256;
    const int b0 = min(A.batch, t * per), b1 = min(A.batch, b0 + per);
    long long local = 0;
    {
        uint32_t lo = b0 < b1 ? fs_wait(&A.fvf[b0], 1u) : 0u;
        for (int b = b0; b < b1; ++b) {
            const uint32_t hi = fs_wait(&A.fvf[b + 1], 1u);
            const long long nvb = (long long)hi - (long long)lo;
            local += nvb < A.max_voxels ? nvb : A.max_voxels;
            A.fv[b] = (int)lo;
            lo = hi;
        }
    }
    if (t == 0) A.fv[A.batch] = s_nv;
    s_part[t] = local;
    __syncthreads();
    if (t == 0) {
        long long run = 0;
        for (int i = 0; i < 256; ++i) { const long long v = s_part[i]; s_part[i] = run; run += v; }
        A.batch_splits[0] = 0;
        A.stats[0] = run;
        A.stats[1] = s_total;
    }
    __syncthreads();
    {
        long long run = s_part[t];
        uint32_t lo = b0 < b1 ? fs_wait(&A.fvf[b0], 1u) : 0u;
        for (int b = b0; b < b1; ++b) {
            const uint32_t hi = fs_wait(&A.fvf[b + 1], 1u);
            const long long nvb = (long long)hi - (long long)lo;
            run += nvb < A.max_voxels ? nvb : A.max_voxels;
            A.batch_splits[b + 1] = run;
            lo = hi;
        }
    }
}

__global__ void vox_fill32(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n, const int* __restrict__ hp,
                           const int* __restrict__ fv, const int* __restrict__ cnt, const int64_t* __restrict__ batch_splits,
                           VoxParams P, int batch, int32_t* __restrict__ coords, int64_t* __restrict__ pidx, int64_t* __restrict__ prs) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v == 0) prs[0] = 0;
    if (v >= n || v >= fv[batch]) return;
    const int h = hp[v];
    const uint32_t key = keys[h];
    const int b = (int)(key / (uint32_t)P.cells);
    const long long rank = v - fv[b];
    if (rank >= P.max_voxels) return;
    const int64_t ov = batch_splits[b] + rank;
    const long long lin = (long long)(key - (uint32_t)b * (uint32_t)P.cells);
    coords[3 * ov + 0] = (int32_t)(lin % P.G[0]);
    coords[3 * ov + 1] = (int32_t)((lin / P.G[0]) % P.G[1]);
    coords[3 * ov + 2] = (int32_t)(lin / (P.G[0] * P.G[1]));
    const int64_t o = cnt[v];
    const int c = cnt[v + 1] - cnt[v];
    prs[ov + 1] = cnt[v + 1];
    for (int j = 0; j < c; ++j) pidx[o + j] = (int64_t)vals[h + j];
}

// ---- grid subsample ------------------------------------------------------------------------------------
// key = item << 40 | linear voxel id.  40 bits per item (a 10 000^3 grid: 600 m of extent at the 0.06 m of the raw-sweep front
// end) instead of round 1's 48: the stable radix sort of the keys runs 8 bits per pass in an even number of passes, so a 64-item batch
// takes 6 passes (47 key bits) instead of 8 (55) -- a quarter of the sort's launches, which are what the batch build's subsampling costs
constexpr int SUB_ITEM_SHIFT = 40;

__global__ void sub_setup(const unsigned* __restrict__ bbox, Segs S, float dl, float* __restrict__ seg, int64_t* err) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S.batch) return;
    float* o = seg + 8 * s;
    if (seg_len(S, s) <= 0) {
        for (int a = 0; a < 3; ++a) { o[a] = 0.f; o[3 + a] = __int_as_float(1); }
        return;
    }
    double cells = 1.0;
    for (int a = 0; a < 3; ++a) {
        const float mn = ord2f(bbox[6 * s + a]), mx = ord2f(bbox[6 * s + 3 + a]);
        const float org = __fmul_rn(floorf(__fdiv_rn(mn, dl)), dl);
        const float gf = floorf(__fdiv_rn(__fsub_rn(mx, org), dl));
        int G = gf < 2.0e9f ? (int)gf + 1 : 0x7fffffff;
        if (G < 1) G = 1;
        o[a] = org;
        o[3 + a] = __int_as_float(G);
        cells *= (double)G;
    }
    if (cells >= 1099511627776.0) *err = 1;     // 2^40 voxels per item
}

__global__ void sub_keys(const float* __restrict__ pts, Segs S, int64_t n, float dl, const float* __restrict__ seg,
                         u64* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s; int64_t local;
    seg_locate(S, i, s, local);
    const float* o = seg + 8 * s;
    const float* p = pts + 3 * i;
    long long c[3], G[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c[a] = (long long)floorf(__fdiv_rn(__fsub_rn(p[a], o[a]), dl));
        G[a] = (long long)__float_as_int(o[3 + a]);
    }
    const u64 lin = (u64)(c[0] + G[0] * (c[1] + G[1] * c[2]));
    keys[i] = ((u64)s << SUB_ITEM_SHIFT) | (lin & ((1ull << SUB_ITEM_SHIFT) - 1ull));
    vals[i] = (uint32_t)i;
}

__global__ void sub_lengths(const int* __restrict__ fv, int batch, int64_t* __restrict__ lengths, int64_t* __restrict__ stats) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) lengths[b] = fv[b + 1] - fv[b];
    if (b == 0) stats[0] = fv[batch];
}

// one thread per output voxel: float32 sums in original point order, then / count
__global__ void sub_fill(const uint32_t* __restrict__ vals, const int* __restrict__ flags, int64_t n,
                         const int* __restrict__ hp, const float* __restrict__ pts, const float* __restrict__ feats,
                         int64_t fdim, const int32_t* __restrict__ labels, float* __restrict__ out_pts,
                         float* __restrict__ out_feats, int32_t* __restrict__ out_labels) {
    const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n || v >= flags[n]) return;
    const int h0 = hp[v], h1 = hp[v + 1];
    const float cnt = (float)(h1 - h0);
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int h = h0; h < h1; ++h) {
        const float* p = pts + 3 * (int64_t)vals[h];
        sx = __fadd_rn(sx, p[0]); sy = __fadd_rn(sy, p[1]); sz = __fadd_rn(sz, p[2]);
    }
    out_pts[3 * v + 0] = __fdiv_rn(sx, cnt);
    out_pts[3 * v + 1] = __fdiv_rn(sy, cnt);
    out_pts[3 * v + 2] = __fdiv_rn(sz, cnt);
    if (feats && out_feats)
        for (int64_t f = 0; f < fdim; ++f) {
            float fs = 0.f;
            for (int h = h0; h < h1; ++h) fs = __fadd_rn(fs, feats[(int64_t)vals[h] * fdim + f]);
            out_feats[v * fdim + f] = __fdiv_rn(fs, cnt);
        }
    if (labels && out_labels) {     // majority vote, ties -> smallest label
        int32_t bestl = 0; int bestc = -1;
        for (int h = h0; h < h1; ++h) {
            const int32_t l = labels[vals[h]];
            int cc = 0;
            for (int e = h0; e < h1; ++e) cc += (labels[vals[e]] == l) ? 1 : 0;
            if (cc > bestc || (cc == bestc && l < bestl)) { bestc = cc; bestl = l; }
        }
        out_labels[v] = bestl;
    }
}

// ---- grid subsample, ONE WORKGROUP PER BATCH ITEM (round 6) -----------------------------------------------------------------------
// The KPConv batch build subsamples ~100 spheres of <= 10 000 points four times per batch; through the stable 64-bit radix sort above
// that is ~30 dependent launches per call over 12-byte pairs.  Here a 1024-thread workgroup owns one item END TO END in LDS, the
// whole call is ONE launch (count) + ONE launch (fill):
//   bounding box (the item's own min / max: the same origin and grid as sub_setup) -> occupancy BITMAP of the item's grid (1 bit per
//   cell, atomicOr) -> popcount prefix of the bitmap words: the voxel ordinal of a cell = words before + bits below, i.e. voxels
//   come out in ascending key order without a sort -> per-voxel counters (16-bit pairs, atomicAdd returns the arrival slot) ->
//   exclusive scan -> scatter of the local point indices -> every run is put back into ORIGINAL point order (insertion sort by the
//   voxel's thread; runs longer than SI_LONG by the whole workgroup: rank = number of smaller indices) -> float32 sums in that order.
// The fill launch repeats the grouping (a few microseconds in LDS) instead of carrying 6 bytes per point through HBM, so the op
// needs NO workspace.  Limits: SI_NMAX points and SI_CMAX grid cells per item (a 4 m sphere at dl = 0.16 m has 125 000); anything
// larger -- whole clouds of the preprocessing front end -- keeps the sort path (the caller retries on stats[1] == 2).
// Three size classes (points, grid cells, threads, LDS): {12 288, 262 144, 1024, 145 KB} for the input spheres, {4096, 65 536, 512,
// 49 KB} and {1024, 16 384, 256, 13 KB} for the pooled levels -- a workgroup that needs a whole CU's LDS waits for one to drain
// when the forward of another batch co-runs (KPConvPipelineN), so the small levels must not ask for it.
constexpr int SI_NMAX = 12288;       // (largest class: ml3d_subsample_items_max_points)
constexpr int SI_LONG = 48;

template <int NMAX, int CMAX, int THREADS>
struct SiSmemT {
    static constexpr int WORDS = CMAX / 32;
    uint32_t bitmap[WORDS];             // occupancy; later the staging buffer of long runs (3 x THREADS floats)
    uint16_t wpre[WORDS];               // voxels before this word
    uint16_t vox[NMAX];                 // voxel ordinal of local point i
    uint32_t cnt[NMAX / 2 + 8];         // 16-bit pairs: per-voxel counters, then IN PLACE hp[v] (first slot; hp[M] = n)
    uint16_t slot[NMAX];                // arrival slot of point i; later the ordered copy of a long run
    uint16_t run[NMAX];                 // local point indices grouped by voxel
    uint16_t longv[NMAX / SI_LONG + 8];
    float red[16][6];
    int scan[16];
    int misc[8];                        // [1] number of long runs
};

// exclusive prefix of one int per thread over the workgroup; `total` = the sum (all threads)
template <int THREADS>
__device__ __forceinline__ int si_block_exscan(int v, int* scan16, int& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int incl = wave_inclusive_scan(v);
    __syncthreads();                                   // (scan16 may still be read from the previous use)
    if (lane == 63) scan16[wv] = incl;
    __syncthreads();
    int carry = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) { const int s = scan16[w]; if (w < wv) carry += s; tot += s; }
    total = tot;
    return carry + incl - v;
}

template <bool FILL, int SI_NMAXC, int SI_CMAX, int SI_THREADS>
__global__ void __launch_bounds__(SI_THREADS)
sub_items_k(const float* __restrict__ pts, const int64_t* __restrict__ splits, int batch, float dl, int64_t* __restrict__ lengths,
            int64_t* __restrict__ stats, float* __restrict__ out_pts) {
    typedef SiSmemT<SI_NMAXC, SI_CMAX, SI_THREADS> SiSmem;
    constexpr int SI_WORDS = SI_CMAX / 32;
    constexpr int SI_NMAX = SI_NMAXC;                  // (shadows the largest class's constant inside the kernel)
    static_assert(SI_WORDS % SI_THREADS == 0 && SI_NMAX % SI_THREADS == 0 && SI_THREADS <= 1024, "size class");
    HIP_DYNAMIC_SHARED(unsigned char, si_raw)
    SiSmem& L = *reinterpret_cast<SiSmem*>(si_raw);
    uint16_t* hp = reinterpret_cast<uint16_t*>(L.cnt);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, b = blockIdx.x;
    const int64_t s0 = splits[b];
    const int n = (int)(splits[b + 1] - s0);
    if (n <= 0) { if (!FILL && t == 0) lengths[b] = 0; return; }
    if (n > SI_NMAX) { if (!FILL && t == 0) { lengths[b] = 0; stats[1] = 2; } return; }
    const float* P = pts + 3 * s0;
    // ---- the item's bounding box -> origin and grid (sub_setup's arithmetic) -----------------------------------------------------------
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = t; i < n; i += SI_THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = P[3 * i + a]; mn[a] = fminf(mn[a], v); mx[a] = fmaxf(mx[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], o)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o)); }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { L.red[wv][a] = mn[a]; L.red[wv][3 + a] = mx[a]; }
    }
    for (int w = t; w < SI_WORDS; w += SI_THREADS) L.bitmap[w] = 0u;
    for (int w = t; w < SI_NMAX / 2 + 8; w += SI_THREADS) L.cnt[w] = 0u;
    if (t == 0) { L.misc[1] = 0; L.misc[2] = 0; }
    __syncthreads();
    float org[3];
    int G[3];
    double cells = 1.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo = L.red[0][a], hi = L.red[0][3 + a];
        for (int w = 1; w < SI_THREADS / 64; ++w) { lo = fminf(lo, L.red[w][a]); hi = fmaxf(hi, L.red[w][3 + a]); }
        org[a] = __fmul_rn(floorf(__fdiv_rn(lo, dl)), dl);
        const float gf = floorf(__fdiv_rn(__fsub_rn(hi, org[a]), dl));
        G[a] = gf < 2.0e9f ? (int)gf + 1 : 0x7fffffff;
        if (G[a] < 1) G[a] = 1;
        cells *= (double)G[a];
    }
    if (cells > (double)SI_CMAX) { if (!FILL && t == 0) { lengths[b] = 0; stats[1] = 2; } return; }
    auto cell_of = [&](int i) -> uint32_t {
        const float* p = P + 3 * i;
        const int c0 = (int)floorf(__fdiv_rn(__fsub_rn(p[0], org[0]), dl));
        const int c1 = (int)floorf(__fdiv_rn(__fsub_rn(p[1], org[1]), dl));
        const int c2 = (int)floorf(__fdiv_rn(__fsub_rn(p[2], org[2]), dl));
        return (uint32_t)(c0 + G[0] * (c1 + G[1] * c2));
    };
    // ---- occupancy bitmap -------------------------------------------------------------------------------------------------------
    for (int i = t; i < n; i += SI_THREADS) {
        const uint32_t c = cell_of(i);
        atomicOr(&L.bitmap[c >> 5], 1u << (c & 31u));
    }
    __syncthreads();
    // ---- voxels before every word (each thread: 8 consecutive words) ------------------------------------------------------------------
    int M;
    {
        int pc[SI_WORDS / SI_THREADS], sum = 0;
#pragma unroll
        for (int j = 0; j < SI_WORDS / SI_THREADS; ++j) { pc[j] = __popc(L.bitmap[t * (SI_WORDS / SI_THREADS) + j]); sum += pc[j]; }
        int run0 = si_block_exscan<SI_THREADS>(sum, L.scan, M);
#pragma unroll
        for (int j = 0; j < SI_WORDS / SI_THREADS; ++j) { L.wpre[t * (SI_WORDS / SI_THREADS) + j] = (uint16_t)run0; run0 += pc[j]; }
    }
    __syncthreads();
    if (!FILL) {
        if (t == 0) { lengths[b] = M; atomicAdd(reinterpret_cast<unsigned long long*>(stats), (unsigned long long)M); }
        return;
    }
    // ---- voxel ordinal + arrival slot of every point ------------------------------------------------------------------------------------
    for (int i = t; i < n; i += SI_THREADS) {
        const uint32_t c = cell_of(i), w = c >> 5;
        const int v = (int)L.wpre[w] + __popc(L.bitmap[w] & ((1u << (c & 31u)) - 1u));
        L.vox[i] = (uint16_t)v;
        const uint32_t old = atomicAdd(&L.cnt[v >> 1], (v & 1) ? 0x10000u : 1u);
        L.slot[i] = (uint16_t)((v & 1) ? (old >> 16) : (old & 0xffffu));
    }
    __syncthreads();
    // ---- counters -> first slots, in place (each thread: 12 consecutive voxels; entry M becomes n) ---------------------------------------
    {
        constexpr int PER = SI_NMAX / SI_THREADS;
        int c[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { c[j] = hp[t * PER + j]; sum += c[j]; }
        int tot, run0 = si_block_exscan<SI_THREADS>(sum, L.scan, tot);
#pragma unroll
        for (int j = 0; j < PER; ++j) { hp[t * PER + j] = (uint16_t)run0; run0 += c[j]; }
        if (t == SI_THREADS - 1) hp[SI_NMAX] = (uint16_t)run0;        // (M == SI_NMAX: every point its own voxel)
    }
    __syncthreads();
    for (int i = t; i < n; i += SI_THREADS) L.run[(int)hp[L.vox[i]] + (int)L.slot[i]] = (uint16_t)i;
    __syncthreads();
    // ---- first output row of this item ---------------------------------------------------------------------------------------------------
    int64_t off = 0;
    {
        long long part = 0;
        for (int q = t; q < b; q += SI_THREADS) part += (long long)lengths[q];
        int lo32 = (int)(part & 0x7fffffffll), hi32 = (int)(part >> 31);      // (two 31-bit halves through the int scan)
        int tl, th;
        (void)si_block_exscan<SI_THREADS>(lo32, L.scan, tl);
        (void)si_block_exscan<SI_THREADS>(hi32, L.scan, th);
        off = ((int64_t)th << 31) + (int64_t)tl;
    }
    // ---- barycentres: float32 sums in original point order ----------------------------------------------------------------------------
    for (int v = t; v < M; v += SI_THREADS) {
        const int h0 = hp[v], h1 = hp[v + 1], c = h1 - h0;
        if (c > SI_LONG) { L.longv[atomicAdd(&L.misc[1], 1)] = (uint16_t)v; continue; }
        for (int i = h0 + 1; i < h1; ++i) {                         // insertion sort of the run (ascending local index)
            const uint16_t x = L.run[i];
            int j = i - 1;
            while (j >= h0 && L.run[j] > x) { L.run[j + 1] = L.run[j]; --j; }
            L.run[j + 1] = x;
        }
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int h = h0; h < h1; ++h) {
            const float* p = P + 3 * (int)L.run[h];
            sx = __fadd_rn(sx, p[0]); sy = __fadd_rn(sy, p[1]); sz = __fadd_rn(sz, p[2]);
        }
        const float fc = (float)c;
        float* o = out_pts + 3 * (off + v);
        o[0] = __fdiv_rn(sx, fc); o[1] = __fdiv_rn(sy, fc); o[2] = __fdiv_rn(sz, fc);
    }
    __syncthreads();
    const int n_long = L.misc[1];
    float* stage = reinterpret_cast<float*>(L.bitmap);             // (the bitmap is no longer needed)
    for (int q = 0; q < n_long; ++q) {
        const int v = L.longv[q], h0 = hp[v], h1 = hp[v + 1], c = h1 - h0;
        for (int e = t; e < c; e += SI_THREADS) {                   // ordered copy: rank = number of smaller indices in the run
            const uint16_t x = L.run[h0 + e];
            int r = 0;
            for (int j = 0; j < c; ++j) r += (L.run[h0 + j] < x) ? 1 : 0;
            L.slot[h0 + r] = x;
        }
        __syncthreads();
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int base = 0; base < c; base += SI_THREADS) {          // 1024 points at a time through LDS, summed in order by thread 0
            const int e = base + t;
            if (e < c) {
                const float* p = P + 3 * (int)L.slot[h0 + e];
                stage[3 * t + 0] = p[0]; stage[3 * t + 1] = p[1]; stage[3 * t + 2] = p[2];
            }
            __syncthreads();
            if (t == 0) {
                const int m = c - base < SI_THREADS ? c - base : SI_THREADS;
                for (int j = 0; j < m; ++j) {
                    sx = __fadd_rn(sx, stage[3 * j]); sy = __fadd_rn(sy, stage[3 * j + 1]); sz = __fadd_rn(sz, stage[3 * j + 2]);
                }
            }
            __syncthreads();
        }
        if (t == 0) {
            const float fc = (float)c;
            float* o = out_pts + 3 * (off + v);
            o[0] = __fdiv_rn(sx, fc); o[1] = __fdiv_rn(sy, fc); o[2] = __fdiv_rn(sz, fc);
        }
    }
}

// out[i, j] = (p[i,0] * R[b][0][j] + p[i,1] * R[b][1][j]) + p[i,2] * R[b][2][j]   (transpose: R[b][j][.])
// — the row-vector rotation of batch_grid_subsampling (kpconv.py:2086-2092, 2105-2110), every product and
// sum rounded separately like numpy's float32 `np.sum(expand_dims(p, 2) * R, axis=1)`.
__global__ void rotate_rows_k(const float* __restrict__ pts, Segs S, int64_t n, const float* __restrict__ R,
                              int transpose, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int s; int64_t local;
    seg_locate(S, i, s, local);
    const float* M = R + 9 * s;
    const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float m0 = transpose ? M[3 * j + 0] : M[0 + j];
        const float m1 = transpose ? M[3 * j + 1] : M[3 + j];
        const float m2 = transpose ? M[3 * j + 2] : M[6 + j];
        out[3 * i + j] = __fadd_rn(__fadd_rn(__fmul_rn(x, m0), __fmul_rn(y, m1)), __fmul_rn(z, m2));
    }
}

#define VX_CHECK()                                                   \
    do {                                                             \
        if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;   \
    } while (0)

// sort the pairs, flag the heads, scan, record head positions, first voxel per item
// (the sort may stop after an ODD number of passes -- voxelize: 22 key bits at 16 KITTI sweeps = 3 passes instead of 4 -- and leave its
//  result in the alternate buffers: group_swap_if_alt points W.keys / W.vals there, in the count AND in the fill call)
static void group_swap_if_alt(GroupWs& W, int64_t n, int key_bits) {
    if (sort_result_in_alt(n, key_bits)) {
        u64* k = W.keys; W.keys = W.sort.keys_alt; W.sort.keys_alt = k;
        uint32_t* v = W.vals; W.vals = W.sort.vals_alt; W.sort.vals_alt = v;
    }
}

static int group_pairs(GroupWs& W, int64_t n, int key_bits, u64 key_invalid, u64 item_stride, int item_shift,
                       hipStream_t st) {
    const unsigned nb = (unsigned)((n + 255) / 256);
    if (n > 0) {
        if (sort_pairs_u64(W.keys, W.vals, n, key_bits, W.sort, st, true)) return ML3D_E_LAUNCH;
        group_swap_if_alt(W, n, key_bits);
        hipLaunchKernelGGL(group_heads, dim3(nb), dim3(256), 0, st, W.keys, n, key_invalid, W.flags);
        VX_CHECK();
        if (scan_inclusive_i32(W.flags + 1, n, W.block_sums, st)) return ML3D_E_LAUNCH;
        hipLaunchKernelGGL(group_headpos, dim3(nb), dim3(256), 0, st, W.keys, n, key_invalid, W.flags, W.hp);
        VX_CHECK();
    }
    hipLaunchKernelGGL(group_first_voxel, dim3((unsigned)(W.batch + 1 + 63) / 64), dim3(64), 0, st, W.keys, n, W.flags,
                       W.batch, item_stride, item_shift, W.fv);
    VX_CHECK();
    return 0;
}

static int vox_params(const float* vs, const float* rmin, const float* rmax, int64_t max_points, int64_t max_voxels,
                      int64_t batch, VoxParams* P) {
    if (!vs || !rmin || !rmax || max_points <= 0 || max_voxels <= 0) return ML3D_E_INVALID;
    double cells = 1.0;
    for (int a = 0; a < 3; ++a) {
        if (!(vs[a] > 0.f) || !(rmax[a] >= rmin[a])) return ML3D_E_INVALID;
        P->vs[a] = vs[a]; P->rmin[a] = rmin[a]; P->rmax[a] = rmax[a];
        // largest coordinate a kept point can take is int((max - min) / vs), all in float32
        float q = (float)(rmax[a] - rmin[a]) / vs[a];
        if (!(q < 2.0e9f)) return ML3D_E_UNSUPPORTED;
        P->G[a] = (long long)q + 1;
        if (P->G[a] < 1) P->G[a] = 1;
        cells *= (double)P->G[a];
    }
    if (cells * (double)(batch + 1) >= 9.0e18) return ML3D_E_UNSUPPORTED;
    P->cells = P->G[0] * P->G[1] * P->G[2];
    P->max_points = max_points;
    P->max_voxels = max_voxels;
    return 0;
}

// the fused form (one launch per pass, one grouping launch) when the keys fit 32 bits and the input the hand-off tables;
// ML3D_VOX_FUSED=0 keeps the launch chain (A/B: profiles/r06_voxelize_fused_ab.log).  A pure function of the call's arguments, so the
// fill call finds the count call's choice again.
static bool vox_fused(int64_t n, int64_t batch, const VoxParams& P) {
    static const bool off = [] { const char* e = getenv("ML3D_VOX_FUSED"); return e && atoi(e) == 0; }();
    const double inval = (double)batch * (double)P.cells;
    return !off && inval <= 4294967295.0 && fused_sort_fits(n, bits_for((u64)batch * (u64)P.cells));
}

struct VoxFusedBufs { uint32_t *ka, *va, *kb, *vb; };
static VoxFusedBufs vox_fused_bufs(const GroupWs& W) {
    return {(uint32_t*)W.keys, W.vals, (uint32_t*)W.sort.keys_alt, W.sort.vals_alt};
}

}  // namespace ml3d

using namespace ml3d;

extern "C" size_t ml3d_voxelize_workspace_bytes(int64_t n_points, int64_t batch) {
    if (n_points < 0 || batch <= 0) return 0;
    return group_ws_bytes(n_points, batch);
}

extern "C" int ml3d_voxelize_count(const float* points, int64_t point_stride, const int64_t* row_splits, int64_t batch,
                                   int64_t n_points, const float* voxel_size_host, const float* range_min_host,
                                   const float* range_max_host, int64_t max_points_per_voxel, int64_t max_voxels,
                                   int64_t* out_batch_splits, int64_t* out_stats, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (!row_splits || batch <= 0 || n_points < 0 || n_points > 0x7ffffff0ll || point_stride < 3 ||
        !out_batch_splits || !out_stats || (n_points > 0 && !points))
        return ML3D_E_INVALID;
    VoxParams P;
    int rc = vox_params(voxel_size_host, range_min_host, range_max_host, max_points_per_voxel, max_voxels, batch, &P);
    if (rc) return rc;
    GroupWs W;
    if (!group_ws_carve(workspace, workspace_bytes, n_points, batch, &W)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Segs S = {row_splits, 0, 0, (int)batch};
    const int64_t n = n_points;
    const unsigned nb = (unsigned)((n + 255) / 256);
    const u64 inval = (u64)batch * (u64)P.cells;
    if (vox_fused(n, batch, P)) {
        const FusedWs& F = W.fused;
        const VoxFusedBufs B = vox_fused_bufs(W);
        const int bits = bits_for(inval), passes = fused_passes(bits);
        zero_async(F.base, F.bytes, st);
        const unsigned kb = nb < 512u ? nb : 512u;
        hipLaunchKernelGGL(vox_keys32, dim3(kb), dim3(256), 0, st, points, point_stride, S, n, P, (uint32_t)inval, passes, B.ka, F.ghist);
        VX_CHECK();
        if (sort_u32_fused(B.ka, B.va, B.kb, B.vb, n, bits, F, st) != passes) return ML3D_E_LAUNCH;
        GrpArgs A;
        A.keys = (passes & 1) ? B.kb : B.ka;
        A.n = n; A.inval = (uint32_t)inval; A.cells = (uint32_t)P.cells; A.batch = (int)batch;
        A.max_points = P.max_points; A.max_voxels = P.max_voxels;
        A.hp = W.hp; A.cnt = W.cnt; A.fv = W.fv; A.batch_splits = out_batch_splits; A.stats = out_stats;
        A.ticket = F.ticket + 4; A.gcnt1 = F.gcnt + 4 * FS_MAX_GROUPS; A.gcnt2 = F.gcnt + 5 * FS_MAX_GROUPS;
        A.tA = F.tA; A.gA = F.gA; A.tC = F.tC; A.gC = F.gC; A.fvf = F.fvf;
        hipLaunchKernelGGL(vox_group32, dim3(F.tiles), dim3(256), 0, st, A);
        VX_CHECK();
        return 0;
    }
    if (n > 0) {
        hipLaunchKernelGGL(vox_keys, dim3(nb), dim3(256), 0, st, points, point_stride, S, n, P, W.keys, W.vals);
        VX_CHECK();
    }
    rc = group_pairs(W, n, bits_for(inval), inval, (u64)P.cells, -1, st);
    if (rc) return rc;
    if (n > 0) {
        hipLaunchKernelGGL(vox_counts, dim3(nb), dim3(256), 0, st, W.keys, W.flags, n, W.hp, W.fv, P, W.cnt);
        VX_CHECK();
        if (scan_inclusive_i32(W.cnt + 1, n, W.block_sums, st)) return ML3D_E_LAUNCH;
    }
    hipLaunchKernelGGL(vox_batch_splits, dim3(1), dim3(64), 0, st, W.fv, (int)batch, P.max_voxels, W.cnt, W.flags, n,
                       out_batch_splits, out_stats);
    VX_CHECK();
    return 0;
}

extern "C" int ml3d_voxelize_fill(int64_t batch, int64_t n_points, const float* voxel_size_host,
                                  const float* range_min_host, const float* range_max_host,
                                  int64_t max_points_per_voxel, int64_t max_voxels, const int64_t* batch_splits,
                                  int32_t* out_voxel_coords, int64_t* out_point_indices, int64_t* out_point_row_splits,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (batch <= 0 || n_points < 0 || !batch_splits || !out_point_row_splits) return ML3D_E_INVALID;
    VoxParams P;
    int rc = vox_params(voxel_size_host, range_min_host, range_max_host, max_points_per_voxel, max_voxels, batch, &P);
    if (rc) return rc;
    GroupWs W;
    if (!group_ws_carve(workspace, workspace_bytes, n_points, batch, &W)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (n_points == 0) {
        (void)hipMemsetAsync(out_point_row_splits, 0, sizeof(int64_t), st);
        return 0;
    }
    if (vox_fused(n_points, batch, P)) {
        const VoxFusedBufs B = vox_fused_bufs(W);
        const bool in_b = fused_passes(bits_for((u64)batch * (u64)P.cells)) & 1;
        hipLaunchKernelGGL(vox_fill32, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, st, in_b ? B.kb : B.ka, in_b ? B.vb : B.va,
                           n_points, W.hp, W.fv, W.cnt, batch_splits, P, (int)batch, out_voxel_coords, out_point_indices,
                           out_point_row_splits);
        VX_CHECK();
        return 0;
    }
    group_swap_if_alt(W, n_points, bits_for((u64)batch * (u64)P.cells));
    hipLaunchKernelGGL(vox_fill, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, st, W.keys, W.vals, W.flags,
                       n_points, W.hp, W.fv, W.cnt, batch_splits, P, out_voxel_coords, out_point_indices,
                       out_point_row_splits);
    VX_CHECK();
    return 0;
}

extern "C" size_t ml3d_subsample_workspace_bytes(int64_t n_points, int64_t batch) {
    if (n_points < 0 || batch <= 0) return 0;
    return group_ws_bytes(n_points, batch);
}

extern "C" int ml3d_subsample_count(const float* points, const int64_t* row_splits, int64_t batch, int64_t n_points,
                                    float sample_dl, int64_t* out_lengths, int64_t* out_stats, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (!row_splits || batch <= 0 || batch > 65535 || n_points < 0 || n_points > 0x7ffffff0ll || !(sample_dl > 0.f) ||
        !out_lengths || !out_stats || (n_points > 0 && !points))
        return ML3D_E_INVALID;
    GroupWs W;
    if (!group_ws_carve(workspace, workspace_bytes, n_points, batch, &W)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Segs S = {row_splits, 0, 0, (int)batch};
    const int64_t n = n_points;
    (void)hipMemsetAsync(out_stats, 0, 2 * sizeof(int64_t), st);
    if (bbox_compute(points, S, n, W.bbox, W.occ, st)) return ML3D_E_LAUNCH;
    hipLaunchKernelGGL(sub_setup, dim3((unsigned)(batch + 63) / 64), dim3(64), 0, st, W.bbox, S, sample_dl, W.seg,
                       out_stats + 1);
    VX_CHECK();
    if (n > 0) {
        hipLaunchKernelGGL(sub_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, points, S, n, sample_dl, W.seg,
                           W.keys, W.vals);
        VX_CHECK();
    }
    int rc = group_pairs(W, n, SUB_ITEM_SHIFT + bits_for((u64)batch), ~0ull, 0ull, SUB_ITEM_SHIFT, st);
    if (rc) return rc;
    hipLaunchKernelGGL(sub_lengths, dim3((unsigned)(batch + 63) / 64), dim3(64), 0, st, W.fv, (int)batch, out_lengths,
                       out_stats);
    VX_CHECK();
    return 0;
}

extern "C" int ml3d_subsample_fill(const float* points, const float* features, int64_t feature_dim,
                                   const int32_t* labels, int64_t batch, int64_t n_points, float* out_points,
                                   float* out_features, int32_t* out_labels, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (batch <= 0 || n_points < 0 || feature_dim < 0) return ML3D_E_INVALID;
    if (n_points == 0) return 0;
    if (!points || !out_points) return ML3D_E_INVALID;
    GroupWs W;
    if (!group_ws_carve(workspace, workspace_bytes, n_points, batch, &W)) return ML3D_E_WORKSPACE;
    group_swap_if_alt(W, n_points, SUB_ITEM_SHIFT + bits_for((u64)batch));
    hipLaunchKernelGGL(sub_fill, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W.vals,
                       W.flags, n_points, W.hp, points, features, feature_dim, labels, out_points, out_features,
                       out_labels);
    VX_CHECK();
    return 0;
}

extern "C" int64_t ml3d_subsample_items_max_points(void) { return SI_NMAX; }

namespace {
template <bool FILL, int NMAX, int CMAX, int THREADS>
int si_launch(const float* points, const int64_t* row_splits, int64_t batch, float dl, int64_t* lengths, int64_t* stats, float* out,
              hipStream_t st) {
    const auto kern = sub_items_k<FILL, NMAX, CMAX, THREADS>;
    const size_t sm = sizeof(SiSmemT<NMAX, CMAX, THREADS>);
    if (sm > 48 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)
        return ML3D_E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(THREADS), sm, st, points, row_splits, (int)batch, dl, lengths, stats, out);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}
// the size class is a function of the largest item (host value) alone, so count and fill agree on it
template <bool FILL>
int si_dispatch(int64_t max_item_points, const float* points, const int64_t* row_splits, int64_t batch, float dl, int64_t* lengths,
                int64_t* stats, float* out, hipStream_t st) {
    if (max_item_points <= 1024) return si_launch<FILL, 1024, 16384, 256>(points, row_splits, batch, dl, lengths, stats, out, st);
    if (max_item_points <= 4096) return si_launch<FILL, 4096, 65536, 512>(points, row_splits, batch, dl, lengths, stats, out, st);
    return si_launch<FILL, SI_NMAX, 262144, 1024>(points, row_splits, batch, dl, lengths, stats, out, st);
}
}  // namespace

extern "C" int ml3d_subsample_items_count(const float* points, const int64_t* row_splits, int64_t batch, int64_t n_points,
                                          float sample_dl, int64_t max_item_points, int64_t* out_lengths, int64_t* out_stats,
                                          void* stream) {
    if (!row_splits || batch <= 0 || batch > 65535 || n_points < 0 || !(sample_dl > 0.f) || !out_lengths || !out_stats ||
        (n_points > 0 && !points))
        return ML3D_E_INVALID;
    if (max_item_points < 0 || max_item_points > SI_NMAX) return ML3D_E_UNSUPPORTED;      // (the caller keeps ml3d_subsample_count)
    hipStream_t st = (hipStream_t)stream;
    zero_async(out_stats, 2 * sizeof(int64_t), st);
    return si_dispatch<false>(max_item_points, points, row_splits, batch, sample_dl, out_lengths, out_stats, nullptr, st);
}

extern "C" int ml3d_subsample_items_fill(const float* points, const int64_t* row_splits, int64_t batch, int64_t n_points,
                                         float sample_dl, int64_t max_item_points, const int64_t* lengths, float* out_points,
                                         void* stream) {
    if (!row_splits || batch <= 0 || batch > 65535 || n_points < 0 || !(sample_dl > 0.f) || !lengths) return ML3D_E_INVALID;
    if (max_item_points < 0 || max_item_points > SI_NMAX) return ML3D_E_UNSUPPORTED;
    if (n_points == 0) return 0;
    if (!points || !out_points) return ML3D_E_INVALID;
    return si_dispatch<true>(max_item_points, points, row_splits, batch, sample_dl, const_cast<int64_t*>(lengths), nullptr, out_points,
                             (hipStream_t)stream);
}

extern "C" int ml3d_rotate_points(const float* points, const int64_t* row_splits, int64_t batch, int64_t n_points,
                                  const float* rotations, int transpose, float* out, void* stream) {
    if (batch <= 0 || n_points < 0 || !row_splits || !rotations) return ML3D_E_INVALID;
    if (n_points == 0) return 0;
    if (!points || !out) return ML3D_E_INVALID;
    Segs S = {row_splits, 0, 0, (int)batch};
    hipLaunchKernelGGL(rotate_rows_k, dim3((unsigned)((n_points + 255) / 256)), dim3(256), 0, (hipStream_t)stream, points,
                       S, n_points, rotations, transpose, out);
    VX_CHECK();
    return 0;
}
