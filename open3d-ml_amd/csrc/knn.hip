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
// gfx950).  Measured (tools/micro/valu_rates.hip, profiles/r02_micro_valu_rates.log): v_min_f64 / v_max_f64 issue at the
// rate of v_min_u32; the integer form (v_cmp_lt_u64 + 4 v_cndmask) is 7.6x slower.
// (Measured and removed, numbers in profiles/DESIGN_rounds_1_to_4.md §3.2: a branchless insertion, a pending queue merged by sorting networks, an
// LDS-transposed index store.)
template <int K>
__device__ __forceinline__ void topk_insert(double (&best)[K], double key) {
    if (key < best[K - 1]) {
        // front to back, every slot in place: slot j keeps min(slot, x) and hands max(slot, x) on -- the key settles at its rank,
        // everything behind it moves down one, the old last entry falls off the end (31 instructions for K = 16; no register
        // rotation: key_insert_step, gfx950_ops.h)
        double x = key;
#pragma unroll
        for (int j = 0; j < K; ++j) key_insert_step(best[j], x);
    }
}

constexpr int KNN_GROUP = 3;

// SUB: also track the nearest candidate whose index is below n_sub -- the RandLA pyramid's 1-NN interpolation target
// (level l + 1 is the prefix [:n_sub] of level l, randlanet.py:222-224), found in the same scan as the k-NN
template <int K, bool SUB>
__device__ __forceinline__ void scan_run(const GridView& G, int p0, int p1, float qx, float qy,
                                         float qz, double (&best)[K], int n_sub, double& best1) {
    // [p0, p1): the run's slice of the cell-sorted array (the caller read the two cell_start entries, one row ahead).
    // candidates in groups of KNN_GROUP: the 16-byte loads of a group are in flight together (one exposed memory latency
    // per group instead of one per candidate); indices past the run are clamped to its last point and skipped
    // (addressing: the array base is wave-uniform (SGPR pair), the lane's position a 32-bit BYTE offset -- `global_load ... v_off,
    //  s[base]` with the group's three loads as immediate offsets -- so a group costs one add and one compare.  The last group of
    //  a run may read up to KNN_GROUP - 1 entries PAST the run -- the next cell's points, or the slack grid_ws_bytes keeps behind
    //  the array -- which the `< end` test below discards: clamping every index to the run cost three VALU instructions per
    //  candidate in a loop that is VALU-issue bound (n_total < 2^28: the launcher's GRID_CAP check, so byte offsets fit 32 bits).
    //  The squared distance pairs x with y -- an even-aligned register pair of the 16-byte load, one v_pk_add_f32 / v_pk_mul_f32
    //  without moves -- in the canonical order ((dx dx) + (dy dy)) + (dz dz), no fma.)
    typedef float v2f __attribute__((ext_vector_type(2)));
    const char* __restrict__ base = reinterpret_cast<const char*>(G.sorted);
    const uint32_t end = (uint32_t)p1 * 16u;
    const v2f qxy = {qx, qy};
    for (uint32_t off = (uint32_t)p0 * 16u; off < end; off += 16u * KNN_GROUP) {
        float4 c[KNN_GROUP];
#pragma unroll
        for (int j = 0; j < KNN_GROUP; ++j) c[j] = *reinterpret_cast<const float4*>(base + off + 16u * j);
#pragma unroll
        for (int j = 0; j < KNN_GROUP; ++j) {
            if (!(off + 16u * j < end)) continue;
            v2f dxy = qxy - (v2f){c[j].x, c[j].y};
            dxy = dxy * dxy;
            const float dz = qz - c[j].z;
            const float d2 = (dxy.x + dxy.y) + dz * dz;
            const u64 key = ((u64)__float_as_uint(d2) << 32) | (u64)(unsigned)__float_as_int(c[j].w);
            const double kd = __longlong_as_double((long long)key);
            if (SUB) key_min_if(__float_as_int(c[j].w) < n_sub, best1, kd);      // (prefix candidates only: level l + 1 = [:n_sub])
            topk_insert<K>(best, kd);
        }
    }
}

// every point outside the box of shell r around cell (cx, cy, cz) is at least this far from the query (inf when the box
// reaches past the grid on every side: `all`)
__device__ __forceinline__ float shell_guard(const GridSeg& g, float qx, float qy, float qz, int cx, int cy, int cz, int r,
                                             bool& all) {
    const int dxm = g.dims[0] - 1, dym = g.dims[1] - 1, dzm = g.dims[2] - 1;
    all = (cx - r <= 0) && (cx + r >= dxm) && (cy - r <= 0) && (cy + r >= dym) && (cz - r <= 0) && (cz + r >= dzm);
    float gd = 3.0e38f;
    if (cx - r > 0) gd = fminf(gd, qx - (g.lo[0] + (float)(cx - r) * g.c));
    if (cx + r < dxm) gd = fminf(gd, (g.lo[0] + (float)(cx + r + 1) * g.c) - qx);
    if (cy - r > 0) gd = fminf(gd, qy - (g.lo[1] + (float)(cy - r) * g.c));
    if (cy + r < dym) gd = fminf(gd, (g.lo[1] + (float)(cy + r + 1) * g.c) - qy);
    if (cz - r > 0) gd = fminf(gd, qz - (g.lo[2] + (float)(cz - r) * g.c));
    if (cz + r < dzm) gd = fminf(gd, (g.lo[2] + (float)(cz + r + 1) * g.c) - qz);
    return gd - g.margin;
}

__device__ __forceinline__ float axis_gap(float q, float lo, float c, int i, float margin) {
    // distance along one axis from q to the slab of cell i: [lo + i c, lo + (i + 1) c); 0 inside
    const float a = (lo + (float)i * c) - q, b = q - (lo + (float)(i + 1) * c);
    return fmaxf(fmaxf(a, b) - margin, 0.f);
}

// The shell between box radius r_in (already read) and r_out: every cell with r_in < max(|dx|, |dy|, |dz|) <= r_out, one run of
// cells per row (rows that cross the inner box: the two ends).  r_out = r_in + 1 is the classic unit shell.
// An isolated point (a lone return metres from anything: 1-7 % of the waves hold one) walks tens of shells before it has k
// neighbours; one shell at a time that is (2 r + 1)^2 row visits per shell, O(R^3) in all, and ONE such lane sets the time of a
// small launch (a single 45 056-point frame: 1.2 ms on average, up to 6.9 ms, against 1.7 ms for 64 frames: round-4 profile of
// the model-class patch loop).  Beyond the third shell the box therefore grows GEOMETRICALLY (and jumps straight to the box
// that holds the k-th neighbour's ball once k points are known): O(R^2) row visits.  `bound`: squared distance beyond which a
// row cannot matter (inf while the list is short); a superset of what unit shells would read, so the result is unchanged.
template <int K, bool SUB>
__device__ __forceinline__ void scan_shell(const GridView& G, const GridSeg& g, float qx, float qy, float qz, int cx, int cy, int cz,
                                           int r_in, int r_out, float bound, double (&best)[K], int n_sub, double& best1) {
    const int dxm = g.dims[0] - 1, dym = g.dims[1] - 1, dzm = g.dims[2] - 1;
    const int xa = max(cx - r_out, 0), xb = min(cx + r_out, dxm);
    const int ya = max(cy - r_out, 0), yb = min(cy + r_out, dym);
    const int za = max(cz - r_out, 0), zb = min(cz + r_out, dzm);
    // a row's cells [xa, xb] are one run, or -- where the row crosses the box already read -- the two ends [xa, lb] and [ra, xb];
    // either way four cell_start entries describe it (an empty end: two equal entries).  They are requested ONE ROW AHEAD: a row
    // costs two dependent round trips (bounds, then candidates), and a lane in a sparse region walks hundreds of near-empty rows
    // in which the bounds are all there is -- the time of a small launch is its slowest lane's.
    const int lb = min(cx - r_in - 1, xb), ra = max(cx + r_in + 1, xa);      // (meaningful for rows inside the inner box's y, z range)
    const int i1 = max(lb + 1, xa), i2 = min(ra, xb + 1);
    auto request = [&](int y, int z, int (&b)[4]) {
        const int32_t* cs = G.cell_start + g.cell_base + g.dims[0] * (y + g.dims[1] * z);
        b[0] = cs[xa]; b[1] = cs[i1]; b[2] = cs[i2]; b[3] = cs[xb + 1];
    };
    int nb[4];
    request(ya, za, nb);
    for (int z = za; z <= zb; ++z) {
        const int az = z > cz ? z - cz : cz - z;
        const float ez = axis_gap(qz, g.lo[2], g.c, z, g.margin);
        for (int y = ya; y <= yb; ++y) {
            const int cur[4] = {nb[0], nb[1], nb[2], nb[3]};
            // the next row of the walk (past the end: this one again)
            const bool last_y = y == yb;
            const int yn = last_y ? (z < zb ? ya : y) : y + 1, zn = last_y && z < zb ? z + 1 : z;
            request(yn, zn, nb);
            const int ay = y > cy ? y - cy : cy - y;
            const float ey = axis_gap(qy, g.lo[1], g.c, y, g.margin);
            if ((ez * ez + ey * ey) * 0.999999f > bound) continue;
            if (r_in == 0 || az > r_in || ay > r_in) {
                scan_run<K, SUB>(G, cur[0], cur[3], qx, qy, qz, best, n_sub, best1);
            } else {
                // (inlined copies of the scan: a one-copy loop over the two ends costs 9 % -- 1.73 against 1.59 ms alone)
                scan_run<K, SUB>(G, cur[0], cur[1], qx, qy, qz, best, n_sub, best1);
                scan_run<K, SUB>(G, cur[2], cur[3], qx, qy, qz, best, n_sub, best1);
            }
        }
    }
}

// Hand-off of the queries the first shell (the 3 x 3 x 3 block) does not settle, from the launch that reads that block for EVERY query
// to a second launch that walks the further shells for those queries ONLY (round 6).  In one launch the ~44 % of the lanes that go on
// drag their waves through the shell walk at that lane utilisation -- every candidate step of a wave costs the full insertion network
// as soon as ONE lane inserts; compacted, the same walks run in full waves.  State per query: the 16 + 1 best keys (136 bytes) and the
// query's (job, row); everything else is recomputed.  A full list (cap) is not an error: the lane simply continues inline.
struct Handoff {
    unsigned* count;        // number of appended queries (may exceed cap: the surplus continued inline)
    uint32_t* who;          // [cap] job << 28 | query row of the job (rows < 2^28: the launcher's GRID_CAP check)
    double* keys;           // [17][cap] best[0 .. 15], best1
    unsigned cap;
};

// MODE 0: the whole search.  1: first shell, then hand-off (H).  2: continuation of hand-off entry `hslot` from the second shell on.
template <int K, bool SUB, int MODE = 0>
__device__ __forceinline__ void knn_one(const GridView& G, const QuerySrc& Q, int k, int index_local,
                                        const Segs& support_segs, int32_t* __restrict__ out_idx,
                                        float* __restrict__ out_d2, int64_t t, int n_sub = 0,
                                        int32_t* __restrict__ out_sub = nullptr, bool vec_store = false,
                                        const Handoff* H = nullptr, unsigned job = 0u, unsigned hslot = 0u) {
    if (t >= Q.n_total) return;
    int s = 0; int64_t local = 0;
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
    const int64_t out_row = seg_begin_packed(Q.segs, s) + local;

    const GridSeg g = G.segs[s];
    double best[K];
#pragma unroll
    for (int j = 0; j < K; ++j) best[j] = __longlong_as_double((long long)KEY_EMPTY);
    double best1 = __longlong_as_double((long long)KEY_EMPTY);

    if (g.n > 0) {
        int cx = cell_coord(qx, g.lo[0], g.inv_c, g.dims[0]);
        int cy = cell_coord(qy, g.lo[1], g.inv_c, g.dims[1]);
        int cz = cell_coord(qz, g.lo[2], g.inv_c, g.dims[2]);
        constexpr int UNIT_SHELLS = 3;        // shells read one at a time before the box starts to grow geometrically
        // squared distance both answers are known within (inf: not yet) -- a function of the lists alone
        auto known_within = [&](u64& kth) -> float {
            kth = (u64)__double_as_longlong(best[K - 1]);
            if (k < K) {
                // fewer than K requested: the k-th entry decides
#pragma unroll
                for (int j = 0; j < K; ++j) if (j == k - 1) kth = (u64)__double_as_longlong(best[j]);
            }
            float d = 3.0e38f;
            if (kth != KEY_EMPTY) {
                d = __uint_as_float((unsigned)(kth >> 32));
                if (SUB && n_sub > 0) {       // (n_sub == 0: no coarser level, the interpolation index is -1 whatever is scanned)
                    // both answers must be final: the nearest prefix point may lie beyond the k-th neighbour
                    const u64 k1 = (u64)__double_as_longlong(best1);
                    if (k1 != KEY_EMPTY) d = fmaxf(d, __uint_as_float((unsigned)(k1 >> 32))); else d = 3.0e38f;
                }
            }
            return d;
        };
        float dk = 3.0e38f;
        int r_in = 0, r = 1;
        if constexpr (MODE == 2) {
            // the state the first launch left: the lists after the 3 x 3 x 3 block; the walk resumes with the second shell
#pragma unroll
            for (int j = 0; j < K; ++j) best[j] = H->keys[(size_t)j * H->cap + hslot];
            best1 = H->keys[(size_t)K * H->cap + hslot];
            u64 kth0;
            dk = known_within(kth0);
            r_in = 1; r = 2;
        }
        for (;;) {
            scan_shell<K, SUB>(G, g, qx, qy, qz, cx, cy, cz, r_in, r, dk, best, n_sub, best1);
            // every point outside the scanned box is at least `gd` away (inf when the box face is
            // past the grid).  Stop once the k-th best is strictly inside that radius.
            bool all;
            const float gd = shell_guard(g, qx, qy, qz, cx, cy, cz, r, all);
            if (all) break;
            u64 kth;
            dk = known_within(kth);
            if (kth != KEY_EMPTY && gd > 0.f && dk < gd * gd * 0.999999f) break;
            if constexpr (MODE == 1) {
                if (r == 1) {
                    const unsigned slot = wave_append(H->count);
                    if (slot < H->cap) {
#pragma unroll
                        for (int j = 0; j < K; ++j) H->keys[(size_t)j * H->cap + slot] = best[j];
                        H->keys[(size_t)K * H->cap + slot] = best1;
                        H->who[slot] = (job << 28) | (uint32_t)t;
                        return;
                    }
                }
            }
            r_in = r;
            if (r < UNIT_SHELLS) { ++r; continue; }
            int rn = r + max(1, r >> 1);
            if (dk < 3.0e38f) {
                // the box that holds the ball of the k-th neighbour: nothing beyond it can matter
                const float rad = sqrtf(dk) * 1.00001f + g.margin;
                const int need = max(max(cx - cell_coord(qx - rad, g.lo[0], g.inv_c, g.dims[0]), cell_coord(qx + rad, g.lo[0], g.inv_c, g.dims[0]) - cx),
                                     max(max(cy - cell_coord(qy - rad, g.lo[1], g.inv_c, g.dims[1]), cell_coord(qy + rad, g.lo[1], g.inv_c, g.dims[1]) - cy),
                                         max(cz - cell_coord(qz - rad, g.lo[2], g.inv_c, g.dims[2]), cell_coord(qz + rad, g.lo[2], g.inv_c, g.dims[2]) - cz)));
                rn = max(r + 1, need);
            }
            r = rn;
        }
    }
    int64_t base = index_local ? 0 : seg_begin_global(support_segs, s);
    if (SUB) {
        const u64 k1 = (u64)__double_as_longlong(best1);
        out_sub[out_row] = k1 != KEY_EMPTY ? (int32_t)(unsigned)(k1 & 0xffffffffull) : -1;
    }
    if (K == 16 && k == 16 && !out_d2 && vec_store) {
        // The queries of a wave sit in cell-sorted order, their result rows anywhere: 16 stores of 4 bytes per lane at a
        // 64-byte stride are 64 partial-line writes per instruction (WRITE_SIZE 2x the index bytes, profiles/r02_pmc_write).
        // Four 16-byte stores per lane instead: a lane's 64-byte row (the caller checked the base's 16-byte alignment) is
        // complete after four consecutive instructions, a quarter of the write requests.
        struct alignas(16) I4 { int32_t v[4]; };
        I4* dst = reinterpret_cast<I4*>(out_idx + out_row * 16);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            I4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u64 key = (u64)__double_as_longlong(best[4 * qd + e]);
                o.v[e] = key != KEY_EMPTY ? (int32_t)((int64_t)(unsigned)(key & 0xffffffffull) + base) : -1;
            }
            dst[qd] = o;
        }
        return;
    }
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
    knn_one<K, false>(G, Q, k, index_local, support_segs, out_idx, out_d2, (int64_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// Several independent searches in ONE launch (the levels of the RandLA pyramid): the small levels are
// latency-bound (one thread per query, a few thousand queries), so running them back to back leaves
// the chip idle; side by side they hide under the largest level.  Jobs are ordered largest first.
constexpr int KNN_MAX_JOBS = 8;
struct KnnJob {
    GridView G;
    QuerySrc Q;
    Segs support;
    int32_t* out_idx;
    int32_t* out_sub;     // 1-NN among the prefix [:n_sub] of the support (nullptr: not asked for)
    int n_sub;
    unsigned block_begin;
};
struct KnnJobs {
    KnnJob j[KNN_MAX_JOBS];
    int n;
};

#ifndef KNN_WAVES
#define KNN_WAVES 5        // register budget 512 / 5 = 96.  Round 2 ran six waves per SIMD (80 registers: 1.68 against 1.76 ms); with the
#endif                    // shell walk of round 4 six waves spill 19 registers to scratch (+70 % FETCH_SIZE, +40 % WRITE_SIZE:
                          // profiles/r04_pmc_fetch.csv) for no gain: 1.42 ms / 5992 frames/s at six, 1.38 ms / 6041 at five (gpurun r4j)
template <int K, bool SUB, bool HAND>
__global__ void __launch_bounds__(256)
#if KNN_WAVES > 0
ML3D_WAVES_PER_SIMD(KNN_WAVES)
#endif
knn_query_multi(KnnJobs J, int k, int index_local, int vec_store, Handoff H) {
    int ji = 0;
#pragma unroll
    for (int i = 1; i < KNN_MAX_JOBS; ++i)
        if (i < J.n && blockIdx.x >= J.j[i].block_begin) ji = i;
    const KnnJob& jb = J.j[ji];
    knn_one<K, SUB, HAND ? 1 : 0>(jb.G, jb.Q, k, index_local, jb.support, jb.out_idx, nullptr,
                                  (int64_t)(blockIdx.x - jb.block_begin) * blockDim.x + threadIdx.x, jb.n_sub, jb.out_sub, vec_store != 0,
                                  &H, (unsigned)ji);
}

#ifndef KNN_WAVES_HAND
#define KNN_WAVES_HAND 4   // register budget of the two hand-off launches: 128 (at 96 they spill 17 / 33 registers for the list addressing)
#endif
// the first launch of a hand-off (knn_query_multi<.., HAND = true> with its own register budget)
template <int K, bool SUB>
__global__ void __launch_bounds__(256)
ML3D_WAVES_PER_SIMD(KNN_WAVES_HAND)
knn_first(KnnJobs J, int k, int index_local, int vec_store, Handoff H) {
    int ji = 0;
#pragma unroll
    for (int i = 1; i < KNN_MAX_JOBS; ++i)
        if (i < J.n && blockIdx.x >= J.j[i].block_begin) ji = i;
    const KnnJob& jb = J.j[ji];
    knn_one<K, SUB, 1>(jb.G, jb.Q, k, index_local, jb.support, jb.out_idx, nullptr,
                       (int64_t)(blockIdx.x - jb.block_begin) * blockDim.x + threadIdx.x, jb.n_sub, jb.out_sub, vec_store != 0, &H, (unsigned)ji);
}

// the second launch of a hand-off: entry i of the list = one lane; blocks past the list's end leave at once
template <int K, bool SUB>
__global__ void __launch_bounds__(256)
ML3D_WAVES_PER_SIMD(KNN_WAVES_HAND)
knn_continue(KnnJobs J, int k, int index_local, int vec_store, Handoff H) {
    const unsigned have = *H.count;
    const unsigned n = have < H.cap ? have : H.cap;
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t who = H.who[i];
    const KnnJob& jb = J.j[who >> 28];
    knn_one<K, SUB, 2>(jb.G, jb.Q, k, index_local, jb.support, jb.out_idx, nullptr, (int64_t)(who & 0x0fffffffu), jb.n_sub, jb.out_sub,
                       vec_store != 0, &H, who >> 28, i);
}

// threads per workgroup of the multi-job launch.  No LDS, no barriers: ONE wave per workgroup, so a wave slot is refilled as soon
// as its wave retires instead of when four have (1.65 against 1.71 ms per 64-frame launch, +0.5-1 % frames/s: gpurun r4e).
static int knn_block() { return 64; }

// (16 or 32 queries per wave for launches that cannot fill the chip's wave slots -- the five levels of ONE frame are 940 full
//  waves on 5120 slots -- measured: 0.363 -> 0.342 ms at batch 1, i.e. a launch that small is not bound by what shares a wave
//  (the suspect: its slowest single query, an outlier walking hundreds of near-empty rows, two dependent loads each).  Removed.)
// hand-off scratch behind the grids of a pyramid workspace: room for 5/8 of the queries (measured: ~44 % go past the first shell)
static size_t handoff_bytes(int64_t total_queries) {
    const size_t cap = (size_t)(total_queries * 5 / 8 + 64);
    return 256 + ((cap * 4 + 255) & ~(size_t)255) + 17 * cap * 8 + 256;
}
static Handoff handoff_carve(char* p, int64_t total_queries) {
    Handoff H;
    const size_t cap = (size_t)(total_queries * 5 / 8 + 64);
    p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    H.count = (unsigned*)p;  p += 256;
    H.who = (uint32_t*)p;    p += (cap * 4 + 255) & ~(size_t)255;
    H.keys = (double*)p;
    H.cap = (unsigned)cap;
    return H;
}

#ifndef ML3D_KNN_HANDOFF
// A/B switch (build time).  Default 0 = the one-launch search: measured on the MI355X (profiles/r06_knn_handoff_ab.log, same box,
// alternating) the hand-off LOSES -- k-NN alone 2.47 ms (2.31 at five waves per SIMD) against 2.09-2.12 ms, 7350 / 7234 against 7332 / 7471
// frames/s: the 136 bytes of state per open query written and read back (~0.9 GB per 128-frame step), the second launch and the
// register budget of the list addressing cost more than the fuller waves of the shell walk return.  The code stays as the record
// of that experiment ("if it loses, commit the A/B log and stop").
#define ML3D_KNN_HANDOFF 0
#endif

template <bool SUB>
static int launch_query_multi(KnnJobs& J, int k, int index_local, hipStream_t stream, const Handoff* hand = nullptr) {
    const int T = knn_block();
    unsigned blocks = 0;
    int vec = 1;            // 16-byte index stores need every job's rows at a 16-byte aligned base (a view or a carved slab may not be)
    for (int i = 0; i < J.n; ++i) {
        J.j[i].block_begin = blocks;
        blocks += (unsigned)((J.j[i].Q.n_total + T - 1) / T);
        if ((uintptr_t)J.j[i].out_idx & 15) vec = 0;
    }
    if (blocks == 0) return 0;
    Handoff H = {};
    if ((ML3D_KNN_HANDOFF) && hand && k > 8 && k <= 16) {
        // two launches: the 3 x 3 x 3 block of every query, then the further shells of the queries it did not settle, compacted
        H = *hand;
        zero_async(H.count, 16, stream);
        hipLaunchKernelGGL((knn_first<16, SUB>), dim3(blocks), dim3(T), 0, stream, J, k, index_local, vec, H);
        if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
        hipLaunchKernelGGL((knn_continue<16, SUB>), dim3((H.cap + T - 1) / T), dim3(T), 0, stream, J, k, index_local, vec, H);
        return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
    }
    if (k == 1) hipLaunchKernelGGL((knn_query_multi<1, SUB, false>), dim3(blocks), dim3(T), 0, stream, J, k, index_local, vec, H);
    else if (k <= 8) hipLaunchKernelGGL((knn_query_multi<8, SUB, false>), dim3(blocks), dim3(T), 0, stream, J, k, index_local, vec, H);
    else if (k <= 16) hipLaunchKernelGGL((knn_query_multi<16, SUB, false>), dim3(blocks), dim3(T), 0, stream, J, k, index_local, vec, H);
    else if (k <= 32) hipLaunchKernelGGL((knn_query_multi<32, SUB, false>), dim3(blocks), dim3(T), 0, stream, J, k, index_local, vec, H);
    else if (k <= 64) hipLaunchKernelGGL((knn_query_multi<64, SUB, false>), dim3(blocks), dim3(T), 0, stream, J, k, index_local, vec, H);
    else return ML3D_E_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

static int launch_query(const GridView& G, const QuerySrc& Q, int k, int index_local, Segs support_segs,
                        int32_t* out_idx, float* out_d2, hipStream_t stream) {
    if (Q.n_total <= 0) return 0;
    dim3 grid((unsigned)((Q.n_total + 255) / 256)), block(256);
    if (k == 1)
        hipLaunchKernelGGL(knn_query<1>, grid, block, 0, stream, G, Q, k, index_local, support_segs, out_idx, out_d2);
    else if (k <= 8)
        hipLaunchKernelGGL(knn_query<8>, grid, block, 0, stream, G, Q, k, index_local, support_segs, out_idx, out_d2);
    else if (k <= 16)
        hipLaunchKernelGGL(knn_query<16>, grid, block, 0, stream, G, Q, k, index_local, support_segs, out_idx, out_d2);
    else if (k <= 32)
        hipLaunchKernelGGL(knn_query<32>, grid, block, 0, stream, G, Q, k, index_local, support_segs, out_idx, out_d2);
    else if (k <= 64)
        hipLaunchKernelGGL(knn_query<64>, grid, block, 0, stream, G, Q, k, index_local, support_segs, out_idx, out_d2);
    else
        return ML3D_E_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

// tile order for the forward's attention kernels: the grid's cell-sorted (packed, cloud-major) sequence of point rows
__global__ void order_from_grid(const float4* __restrict__ sorted, int64_t n_total, int64_t n_per_item,
                                int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    out[i] = (int32_t)((i / n_per_item) * n_per_item + (int64_t)__float_as_int(sorted[i].w));
}

// grid occupancy target handed to grid_build: 0 = the builder's own choice (mean occupancy of the non-empty cells ~4).  A constant:
// the library reads nothing from the environment and keeps no process-wide state.
static constexpr float tuning_occ() { return 0.f; }

}  // namespace ml3d

using namespace ml3d;

extern "C" int ml3d_abi_version(void) { return ML3D_ABI_VERSION; }

extern "C" size_t ml3d_knn_workspace_bytes(int64_t n_points, int64_t n_queries, int64_t batch) {
    (void)n_queries;
    return grid_ws_bytes(n_points, batch);
}

extern "C" int ml3d_knn_search(const float* points, const int64_t* points_row_splits, const float* queries,
                               const int64_t* queries_row_splits, int64_t batch, int64_t n_points,
                               int64_t n_queries, int k, int index_local, int32_t* out_index,
                               float* out_dist2, void* workspace, size_t workspace_bytes, void* stream) {
    if (!points_row_splits || !queries_row_splits || batch <= 0 || k <= 0 || n_points < 0 || n_queries < 0 ||
        n_points > 0x7fffffffll / GRID_CAP - 4096 || n_queries > 0x7fffffffll)
        return ML3D_E_INVALID;
    if (k > 64) return ML3D_E_UNSUPPORTED;
    if (n_queries == 0) return 0;
    if (!out_index || (n_points > 0 && !points) || !queries) return ML3D_E_INVALID;
    GridWs ws;
    if (!grid_ws_carve(workspace, workspace_bytes, n_points, batch, &ws)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Segs ps = {points_row_splits, 0, 0, (int)batch};
    Segs qs = {queries_row_splits, 0, 0, (int)batch};
    int rc = grid_build(points, ps, ws, tuning_occ(), st);
    if (rc) return ML3D_E_LAUNCH;
    GridView G = grid_view(ws);
    QuerySrc Q;
    bool self = (queries == points) && (queries_row_splits == points_row_splits) && (n_queries == n_points);
    Q.sorted_q = self ? ws.sorted : nullptr;
    Q.qsegs = ws.segs;
    Q.raw = queries;
    Q.segs = qs;
    Q.n_total = n_queries;
    return launch_query(G, Q, k, index_local, ps, out_index, out_dist2, st);
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
    if (num_layers <= 0 || num_layers > KNN_MAX_JOBS || !ratios_host) return 0;
    int64_t n[17];
    if (pyramid_sizes(n0, num_layers, ratios_host, n)) return 0;
    size_t b = 0;
    int64_t total = 0;
    for (int l = 0; l < num_layers; ++l) { b += grid_ws_bytes(n[l] * batch, batch) + 256; total += n[l] * batch; }
    return b + handoff_bytes(total);
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
    auto tb = [&](int tag) { for (const ml3d_trace* r = tr; r; r = r->next) if (r->tag == tag && r->ev_start) (void)hipEventRecord((hipEvent_t)r->ev_start, (hipStream_t)stream); };
    auto te = [&](int tag) { for (const ml3d_trace* r = tr; r; r = r->next) if (r->tag == tag && r->ev_stop) (void)hipEventRecord((hipEvent_t)r->ev_stop, (hipStream_t)stream); };
    if (!points || batch <= 0 || n0 <= 0 || num_layers <= 0 || num_layers > KNN_MAX_JOBS || !ratios_host || k <= 0 ||
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
    char* p = (char*)workspace;
    for (int l = 0; l < num_layers; ++l) {
        size_t bytes = grid_ws_bytes(n[l] * batch, batch) + 256;
        if (!grid_ws_carve(p, bytes, n[l] * batch, batch, &ws[l])) return ML3D_E_WORKSPACE;
        p += bytes;
    }
    float occ = tuning_occ();
    // grids of the searched levels: level l = prefix [:n_l] of each cloud (randlanet.py:222)
    for (int l = 0; l < num_layers; ++l) {
        if (n[l] == 0) continue;
        Segs S = {nullptr, n0, n[l], (int)batch};
        tb(100 + l);
        // level 0 probes the cloud; the thinner prefix levels reuse its box and dimension estimate
        if (l == 0 ? grid_build(points, S, ws[l], occ, st) : grid_build_derived(points, S, ws[l], ws[0], st))
            return ML3D_E_LAUNCH;
        if (tile_order_host && tile_order_host[l]) {
            const int64_t nt = n[l] * batch;
            hipLaunchKernelGGL(order_from_grid, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, ws[l].sorted, nt, n[l],
                               tile_order_host[l]);
            if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
        }
        te(100 + l);
    }
    // ONE launch for all levels and both searches of a level: the k-NN of level l onto itself (randlanet.py:220) and the
    // 1-NN of level l in level l + 1 (randlanet.py:224).  Level l + 1 is the prefix [:n_{l+1}] of level l, so the
    // interpolation target is simply the nearest scanned candidate with index < n_{l+1}: it rides along in the k-NN scan
    // (one extra key per lane) instead of eight searches on five grids.
    KnnJobs Jk;
    Jk.n = 0;
    for (int l = 0; l < num_layers; ++l) {
        if (n[l] == 0) continue;
        Segs S = {nullptr, n0, n[l], (int)batch};
        QuerySrc Q;
        Q.sorted_q = ws[l].sorted;
        Q.qsegs = ws[l].segs;
        Q.raw = points;
        Q.segs = S;
        Q.n_total = n[l] * batch;
        KnnJob& a = Jk.j[Jk.n++];
        a.G = grid_view(ws[l]); a.Q = Q; a.support = S; a.out_idx = neighbor_idx_host[l]; a.block_begin = 0;
        a.out_sub = interp_idx_host[l];
        a.n_sub = (int)n[l + 1];          // 0: every interpolation index comes out -1 (no coarser level)
    }
    int64_t total_q = 0;
    for (int l = 0; l < num_layers; ++l) total_q += n[l] * batch;
    const Handoff hand = handoff_carve(p, total_q);
    tb(0);
    int rc = launch_query_multi<true>(Jk, k, 1, st, &hand);
    te(0);
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

// sampler query (semseg_spatially_regular.py:90-91): the reference's search tree is sklearn's KDTree, which converts
// the float32 sub-cloud to float64 and orders neighbours by the float64 reduced distance ((dx*dx + dy*dy) + dz*dz,
// sklearn/metrics/_dist_metrics: sequential accumulation, no FMA).  The patch ORDER feeds random.shuffle and, through the
// prefix subsampling of RandLANet.transform, every coarser level -- so the key is that float64 value, bit for bit
// (positive doubles order like their bit patterns); ties keep ascending index (stable sort, values start as 0..n-1).
__global__ void center_keys(const float* __restrict__ pts, int64_t n, double cx, double cy, double cz,
                            const float* __restrict__ cdev, u64* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (cdev) { cx = (double)cdev[0]; cy = (double)cdev[1]; cz = (double)cdev[2]; }
    const double dx = (double)pts[3 * i] - cx, dy = (double)pts[3 * i + 1] - cy, dz = (double)pts[3 * i + 2] - cz;
    const double d2 = (dx * dx + dy * dy) + dz * dz;      // -ffp-contract=off: three products, two sums, no FMA
    keys[i] = (u64)__double_as_longlong(d2);
    vals[i] = (uint32_t)i;
}

__global__ void center_take(const u64* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t k,
                            int32_t* __restrict__ out_idx, double* __restrict__ out_d2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    out_idx[i] = (int32_t)vals[i];
    if (out_d2) out_d2[i] = __longlong_as_double((long long)keys[i]);
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

// labels = argmax over the classes (first maximum, NaN counts as the maximum -- torch.argmax), one thread per point
__global__ void __launch_bounds__(256)
argmax_rows(const float* __restrict__ scores, int64_t n, int C, uint8_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* row = scores + i * C;
    float best = row[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
        const float v = row[c];
        if ((v > best || v != v) && !(best != best)) { best = v; arg = c; }
    }
    out[i] = (uint8_t)arg;
}

// the same for C <= 32 with the 64 rows of a workgroup staged through LDS: the rows are contiguous in memory, so the workgroup reads
// them as one run of 16-byte loads (one thread per row reads 76-byte-strided scalars: 0.34 ms for the 438 MB of a 128-frame step,
// ~1.3 TB/s), then every thread walks its row in LDS (row pitch C: odd pitches are conflict-free, even ones two-way)
__global__ void __launch_bounds__(64)
argmax_rows_lds(const float* __restrict__ scores, int64_t n, int C, uint8_t* __restrict__ out) {
    // (64 rows per workgroup, <= 8 KB of LDS: the kernel runs on the post stream beside the forward's LDS-heavy kernels, and a 32 KB
    //  workgroup waited for room -- 1.3 ms per launch in step against 0.09 ms alone)
    __shared__ __attribute__((aligned(16))) float tile[64 * 32];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int rows = (int)((n - r0) < 64 ? (n - r0) : 64);
    const int64_t f0 = r0 * C;                                   // first float of the tile
    const int total = rows * C;
    const float* src = scores + f0;
    const int mis = (int)(f0 & 3);                               // floats before the first 16-byte boundary of the tile's run
    const int head = mis ? 4 - mis : 0;
    for (int e = threadIdx.x; e < head && e < total; e += 64) tile[e] = src[e];
    const int n4 = total > head ? (total - head) / 4 : 0;
    const float4* s4 = reinterpret_cast<const float4*>(src + head);
    for (int q = threadIdx.x; q < n4; q += 64) {
        const float4 v = s4[q];
        float* d = tile + head + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int e = head + 4 * n4 + threadIdx.x; e < total; e += 64) tile[e] = src[e];
    __syncthreads();
    if ((int)threadIdx.x >= rows) return;
    const float* row = tile + threadIdx.x * C;
    float best = row[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
        const float v = row[c];
        if ((v > best || v != v) && !(best != best)) { best = v; arg = c; }
    }
    out[r0 + threadIdx.x] = (uint8_t)arg;
}

}  // namespace ml3d

#ifndef ML3D_ARGMAX_LDS
// A/B switch (build time).  Default 0 = one thread per row reading its 76-byte-strided scalars: the LDS-staged form is 4x faster ALONE
// (0.087 against 0.34 ms for a 128-frame step) and SLOWER in the step -- 7476 / 7526 against 7624 / 7615 frames/s, same box, alternating
// (profiles/r06_argmax_ab.log): on the post stream, beside the forward's LDS-resident kernels, its workgroups wait for LDS room.
#define ML3D_ARGMAX_LDS 0
#endif

extern "C" int ml3d_argmax_labels(const float* scores, int64_t n, int num_classes, uint8_t* out_labels, void* stream) {
    if (n < 0 || num_classes <= 0 || num_classes > 256) return ML3D_E_INVALID;
    if (n == 0) return 0;
    if (!scores || !out_labels) return ML3D_E_INVALID;
    if ((ML3D_ARGMAX_LDS) && num_classes <= 32 && (((uintptr_t)scores) & 15) == 0)
        hipLaunchKernelGGL(argmax_rows_lds, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, scores, n, num_classes,
                           out_labels);
    else
        hipLaunchKernelGGL(argmax_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, n, num_classes,
                           out_labels);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" size_t ml3d_nearest_to_center_workspace_bytes(int64_t n_points) {
    if (n_points < 0) return 0;
    const int64_t m = n_points > 0 ? n_points : 1;
    return ((sizeof(u64) * (size_t)m + 255) & ~(size_t)255) + ((sizeof(uint32_t) * (size_t)m + 255) & ~(size_t)255) +
           sort_ws_bytes(m) + 512;
}

static int nearest_to_center_impl(const float* points, int64_t n_points, const float* center_host, const float* center_dev,
                                  int64_t k, int32_t* out_index, double* out_dist2, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    if (n_points < 0 || k < 0 || k > n_points || (!center_host && !center_dev) || n_points > 0x7ffffff0ll) return ML3D_E_INVALID;
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
                       center_host ? (double)center_host[0] : 0.0, center_host ? (double)center_host[1] : 0.0,
                       center_host ? (double)center_host[2] : 0.0, center_dev, keys, vals);
    if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    if (sort_pairs_u64(keys, vals, n_points, 64, sw, st)) return ML3D_E_LAUNCH;
    hipLaunchKernelGGL(center_take, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, keys, vals, k, out_index, out_dist2);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_nearest_to_center(const float* points, int64_t n_points, const float* center_host, int64_t k,
                                      int32_t* out_index, double* out_dist2, void* workspace, size_t workspace_bytes,
                                      void* stream) {
    if (!center_host) return ML3D_E_INVALID;
    return nearest_to_center_impl(points, n_points, center_host, nullptr, k, out_index, out_dist2, workspace, workspace_bytes, stream);
}

extern "C" int ml3d_nearest_to_center_dev(const float* points, int64_t n_points, const float* center_dev, int64_t k,
                                          int32_t* out_index, double* out_dist2, void* workspace, size_t workspace_bytes,
                                          void* stream) {
    if (!center_dev) return ML3D_E_INVALID;
    return nearest_to_center_impl(points, n_points, nullptr, center_dev, k, out_index, out_dist2, workspace, workspace_bytes, stream);
}

// ---- the patch loop on device buffers (ml3d_hip.h, ABI 5) ----------------------------------------------------------------
namespace ml3d {

__global__ void __launch_bounds__(256)
patch_gather(const float* __restrict__ pts, const int32_t* __restrict__ cand, const int32_t* __restrict__ perm,
             const float* __restrict__ center, int64_t k, float* __restrict__ out_pts, int32_t* __restrict__ out_sel,
             float* __restrict__ d2, unsigned* __restrict__ d2max_bits) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float d = 0.f;
    if (j < k) {
        const int32_t i = cand[perm[j]];
        const float x = pts[3 * (int64_t)i], y = pts[3 * (int64_t)i + 1], z = pts[3 * (int64_t)i + 2];
        out_pts[3 * j] = x; out_pts[3 * j + 1] = y; out_pts[3 * j + 2] = z;
        out_sel[j] = i;
        // np.sum(np.square((pc - center).astype(np.float32)), axis=1): three float32 squares added left to right
        const float dx = __fsub_rn(x, center[0]), dy = __fsub_rn(y, center[1]), dz = __fsub_rn(z, center[2]);
        d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        d2[j] = d;
    }
    // max of non-negative floats == max of their bit patterns
    unsigned b = __float_as_uint(d);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = max(b, (unsigned)__shfl_xor((int)b, o));
    if ((threadIdx.x & 63) == 0) atomicMax(d2max_bits, b);
}

__global__ void __launch_bounds__(256)
patch_bump(const int32_t* __restrict__ sel, const float* __restrict__ d2, const unsigned* __restrict__ d2max_bits, int64_t k,
           double* __restrict__ possibility) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const float mx = __uint_as_float(*d2max_bits);
    const float t = __fsub_rn(1.0f, __fdiv_rn(d2[j], mx));          // float32: 1 - dists / np.max(dists)
    possibility[sel[j]] += (double)__fmul_rn(t, t);                 // float64 += float32 square
}

// one workgroup, two LDS stages of 2 560 rows (column-major): waves 1..3 transpose stage i + 1 in while threads 0..2 of wave 0 add
// their column of stage i IN ROW ORDER -- the chain is the 45 056 dependent additions, so everything else is kept off it: the
// next stage's global reads and LDS writes run beside it, 16-byte LDS reads, eight of them (32 values) in flight under the 32
// adds of the previous block, two register blocks in ping-pong (no copies)
__global__ void __launch_bounds__(256)
patch_mean_seq(const float* __restrict__ pts, int64_t k, float* __restrict__ mean_out) {
    constexpr int ROWS = 2560;            // 30 KB of LDS per stage
    __shared__ __attribute__((aligned(16))) float buf[2][3 * ROWS];
    auto fill = [&](int64_t base, float* dst, int first, int step) {
        if (base >= k) return;
        const int rows = (int)min<int64_t>(ROWS, k - base);
        for (int e = first; e < rows * 3; e += step) {
            const int r = e / 3, c = e - 3 * r;
            dst[c * ROWS + r] = pts[3 * base + e];
        }
    };
    fill(0, buf[0], threadIdx.x, 256);
    __syncthreads();
    float s = 0.f;
    int stage = 0;
    for (int64_t base = 0; base < k; base += ROWS, stage ^= 1) {
        const int rows = (int)min<int64_t>(ROWS, k - base);
        if (threadIdx.x >= 64) {
            fill(base + ROWS, buf[stage ^ 1], threadIdx.x - 64, 192);
        } else if (threadIdx.x < 3) {
            const float* col = buf[stage] + threadIdx.x * ROWS;
            const float4* col4 = reinterpret_cast<const float4*>(col);
            int r = 0;
            // blocks of 256 rows as STRAIGHT-LINE code, eight groups of 32 rows: the 16-byte LDS reads of group g + 1 are issued,
            // then the 32 dependent adds of group g run under them -- two register sets in ping-pong with nothing carried
            // around a loop (the rolled two-block loop this replaces paid a v_mov per add for half of the rows, the phi copies
            // of its prefetch registers); only a block's first group is exposed, once per 256 adds.  The scheduling barriers keep
            // the compiler from hoisting all 64 reads to the top (it did: the first add then waited for 46 of them).
            for (; r + 256 <= rows; r += 256) {
                const float4* c4 = col4 + r / 4;
                float4 a[8], b[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = c4[i];
#pragma unroll
                for (int g = 0; g < 8; g += 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) b[i] = c4[8 * (g + 1) + i];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { s = __fadd_rn(s, a[i].x); s = __fadd_rn(s, a[i].y); s = __fadd_rn(s, a[i].z); s = __fadd_rn(s, a[i].w); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (g + 2 < 8) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) a[i] = c4[8 * (g + 2) + i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 8; ++i) { s = __fadd_rn(s, b[i].x); s = __fadd_rn(s, b[i].y); s = __fadd_rn(s, b[i].z); s = __fadd_rn(s, b[i].w); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            for (; r < rows; ++r) s = __fadd_rn(s, col[r]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) mean_out[threadIdx.x] = __fdiv_rn(s, (float)k);
}

__global__ void __launch_bounds__(256)
patch_apply(float* __restrict__ pts, int64_t k, int dims_mask, const float* __restrict__ mean, const float* __restrict__ extra,
            int n_extra, float bias, float scale, float* __restrict__ feats) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const int C = 3 + n_extra;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = pts[3 * j + d];
        if (dims_mask & (1 << d)) { v = __fsub_rn(v, mean[d]); pts[3 * j + d] = v; }
        if (feats) feats[j * C + d] = v;
    }
    if (feats)
        for (int c = 0; c < n_extra; ++c) feats[j * C + 3 + c] = __fdiv_rn(__fsub_rn(extra[j * n_extra + c], bias), scale);
}

}  // namespace ml3d

extern "C" int ml3d_patch_crop(const float* points, int64_t n_points, const int32_t* cand, const int32_t* perm,
                               const float* center_dev, int64_t k, float* out_pts, int32_t* out_sel, double* possibility,
                               void* scratch, size_t scratch_bytes, void* stream) {
    if (k < 0 || n_points < 0 || k > n_points) return ML3D_E_INVALID;
    if (k == 0) return 0;
    if (!points || !cand || !perm || !center_dev || !out_pts || !out_sel || !possibility || !scratch) return ML3D_E_INVALID;
    if (scratch_bytes < sizeof(float) * (size_t)k + 64) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    unsigned* mx = (unsigned*)scratch;
    float* d2 = (float*)((char*)scratch + 64);
    zero_async(mx, 16, st);       // (a fill kernel, not hipMemsetAsync: this call is replayed inside HIP graphs -- grid.h; 64 bytes reserved)
    const dim3 grid((unsigned)((k + 255) / 256)), block(256);
    hipLaunchKernelGGL(patch_gather, grid, block, 0, st, points, cand, perm, center_dev, k, out_pts, out_sel, d2, mx);
    hipLaunchKernelGGL(patch_bump, grid, block, 0, st, out_sel, d2, mx, k, possibility);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_patch_recenter(float* pts, int64_t k, int dims_mask, const float* extra, int n_extra, float feat_bias,
                                   float feat_scale, float* out_features, void* scratch, size_t scratch_bytes, void* stream) {
    if (k < 0 || n_extra < 0 || (n_extra > 0 && !extra) || (dims_mask & ~7)) return ML3D_E_INVALID;
    if (k == 0) return 0;
    if (!pts || !scratch) return ML3D_E_INVALID;
    if (scratch_bytes < 16) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float* mean = (float*)scratch;
    if (dims_mask) hipLaunchKernelGGL(patch_mean_seq, dim3(1), dim3(256), 0, st, pts, k, mean);
    hipLaunchKernelGGL(patch_apply, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, pts, k, dims_mask, mean, extra, n_extra,
                       feat_bias, feat_scale, out_features);
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
