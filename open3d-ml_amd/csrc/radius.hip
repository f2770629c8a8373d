// radius.hip — batched fixed-radius neighbour search + ragged_to_dense (gfx950).
//
// Replaces open3d.ml.torch.layers.FixedRadiusSearch()(supports, queries, r, s_splits, q_splits)
// and open3d.ml.torch.ops.ragged_to_dense as called from batch_neighbors
// (ml3d/torch/models/kpconv.py:2002-2034) — 3 calls per KPConv layer in
// KPConvBatch.segmentation_inputs (ml3d/torch/dataloaders/concat_batcher.py:186-305) — and the
// ragged_to_dense of PointPillarsVoxelization (ml3d/torch/models/point_pillars.py:364-366).
//
// neighbour iff d2 <= r*r, d2 = ((dx*dx)+(dy*dy))+(dz*dz) in f32 without fma; rows are emitted in
// the oracle's canonical order, ascending (d2, index) — so "first column = closest point"
// (closest_pool, kpconv.py:821-838) and "truncate = drop the furthest" (big_neighborhood_filter,
// concat_batcher.py:176-184) hold as in the original KPConv.
//
// One WAVE per query on a counting-sorted grid with cell ~ r: the query's box is <= 3x3x3 cells =
// 9 contiguous x-runs; the 64 lanes stream a run's float4 candidates (coalesced 1 KiB per step),
// ballot the hits and compact them with a popcount prefix into an LDS-staged row (<= 256
// neighbours; longer rows spill to the caller's workspace), then rank-sort the row in place.
// Two-phase ragged result: count -> caller allocates -> fill (the library never allocates).
// Roofline: HBM — algorithmic bytes 12 B/query + 12 B/support read, 4 B per neighbour written.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "grid.h"
#include "ml3d_hip.h"

namespace ml3d {

typedef unsigned long long u64;
constexpr int RAD_LDS_ROW = 256;   // neighbours staged in LDS per wave

struct RadArgs {
    GridView G;
    const float* queries;
    Segs qsegs;      // layout of queries / output rows
    Segs psegs;      // layout of the support points (for global index base)
    int64_t nq;
    float r, r2;
};

// the query of this wave and the cell box that covers its ball
struct RadQuery {
    float qx, qy, qz;
    int s;
    int xa, xb, ya, yb, za, zb;
    bool any;
};

__device__ __forceinline__ RadQuery rad_setup(const RadArgs& A, int64_t t) {
    // (t is wave-uniform and the whole wave is here: the segment lookup is one round trip, seg_locate_wave; with row splits the
    //  query's global row IS t, so its coordinates are requested before the lookup returns)
    RadQuery Q;
    int64_t local, begin;
    const float* p = A.queries + 3 * t;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (A.qsegs.splits) { qx = p[0]; qy = p[1]; qz = p[2]; }
    seg_locate_wave(A.qsegs, t, (int)(threadIdx.x & 63), Q.s, local, begin);
    if (!A.qsegs.splits) {
        p = A.queries + 3 * (seg_begin_global(A.qsegs, Q.s) + local);
        qx = p[0]; qy = p[1]; qz = p[2];
    }
    Q.qx = qx; Q.qy = qy; Q.qz = qz;
    Q.any = false;
    Q.xa = Q.xb = Q.ya = Q.yb = Q.za = Q.zb = 0;
    return Q;
}

__device__ __forceinline__ void rad_box(RadQuery& Q, const GridSeg& g, float r) {
    Q.any = g.n > 0;
    // a point accepted by the f32 test d2 <= r2 lies within r * (1 + 2^-22) + rounding of the
    // differences; pad the box so that cell_coord (monotone in its argument) cannot miss it.
    const float pad = r * 1.00001f + g.margin;
    Q.xa = cell_coord(Q.qx - pad, g.lo[0], g.inv_c, g.dims[0]);
    Q.xb = cell_coord(Q.qx + pad, g.lo[0], g.inv_c, g.dims[0]);
    Q.ya = cell_coord(Q.qy - pad, g.lo[1], g.inv_c, g.dims[1]);
    Q.yb = cell_coord(Q.qy + pad, g.lo[1], g.inv_c, g.dims[1]);
    Q.za = cell_coord(Q.qz - pad, g.lo[2], g.inv_c, g.dims[2]);
    Q.zb = cell_coord(Q.qz + pad, g.lo[2], g.inv_c, g.dims[2]);
}

// ---- the candidate scan shared by both phases ------------------------------------------------------
// The query's box is at most 4 x 4 rows of cells (cell >= r), each row one contiguous run [p0, p1) of the cell-sorted points.
// Walking the rows one by one costs two DEPENDENT round trips per row (cell_start -> sorted) with the wave mostly idle (a row
// holds ~10-60 points): 18-32 serial L2 latencies per query.  Here lanes 0 .. 2*rows-1 fetch all row bounds in ONE load, the
// runs are concatenated by a prefix sum in scalar registers, and the 64 lanes stream the concatenation: two round trips and
// ceil(T / 64) full-width steps per query.  body(active, candidate) is called in wave-uniform control flow.
constexpr int RAD_ROWS = 16;

// ROWS: compile-time bound of the row prefix held in scalar registers (9 = the 3 x 3 rows of the common box, 16 = up to 4 x 4):
// the per-candidate row select is ROWS - 1 compare + select pairs
template <int ROWS, class F>
__device__ __forceinline__ void rad_scan_rows(const RadArgs& A, const RadQuery& Q, const GridSeg& g, int lane, int ny, int nrows, F&& body) {
    int b = 0;
    if (lane < 2 * nrows) {
        const int r = lane >> 1;
        const int row = g.cell_base + g.dims[0] * ((Q.ya + r % ny) + g.dims[1] * (Q.za + r / ny));
        b = A.G.cell_start[row + ((lane & 1) ? Q.xb + 1 : Q.xa)];
    }
    int pre[ROWS + 1], D[ROWS];
    pre[0] = 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int p0 = __builtin_amdgcn_readlane(b, 2 * r), p1 = __builtin_amdgcn_readlane(b, 2 * r + 1);
        D[r] = p0 - pre[r];                      // candidate f of row r is sorted[f + D[r]]
        pre[r + 1] = pre[r] + (p1 - p0);         // (lanes >= 2 * nrows hold 0: empty rows)
    }
    const int T = pre[ROWS];
    for (int f0 = 0; f0 < T; f0 += 64) {
        const int f = f0 + lane;
        int off = D[0];
#pragma unroll
        for (int j = 1; j < ROWS; ++j) off = f >= pre[j] ? D[j] : off;
        const bool act = f < T;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) c = A.G.sorted[f + off];
        body(act, c);
    }
}

template <class F>
__device__ __forceinline__ void rad_scan(const RadArgs& A, const RadQuery& Q, const GridSeg& g, int lane, F&& body) {
    if (!Q.any) return;
    const int ny = Q.yb - Q.ya + 1, nz = Q.zb - Q.za + 1, nrows = ny * nz;
#ifndef ML3D_RAD_ROWS16_ONLY
    if (nrows <= 9) { rad_scan_rows<9>(A, Q, g, lane, ny, nrows, body); return; }
#endif
    if (nrows <= RAD_ROWS) { rad_scan_rows<RAD_ROWS>(A, Q, g, lane, ny, nrows, body); return; }
    for (int z = Q.za; z <= Q.zb; ++z)               // (unreachable with cell >= r; kept for any other grid)
        for (int y = Q.ya; y <= Q.yb; ++y) {
            const int row = g.cell_base + g.dims[0] * (y + g.dims[1] * z);
            const int p0 = A.G.cell_start[row + Q.xa], p1 = A.G.cell_start[row + Q.xb + 1];
            for (int pb = p0; pb < p1; pb += 64) {
                const int p = pb + lane;
                float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p < p1) c = A.G.sorted[p];
                body(p < p1, c);
            }
        }
}

// ---- phase 1: neighbours per query ---------------------------------------------------------------
__global__ void __launch_bounds__(256) radius_count(RadArgs A, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= A.nq) return;                       // wave-uniform
    RadQuery Q = rad_setup(A, t);
    const GridSeg g = A.G.segs[Q.s];
    rad_box(Q, g, A.r);
    int total = 0;
    rad_scan(A, Q, g, lane, [&](bool act, const float4& c) {
        const bool hit = act && dist2_canon(Q.qx, Q.qy, Q.qz, c.x, c.y, c.z) <= A.r2;
        total += __popcll(__ballot(hit));
    });
    if (lane == 0) counts[t] = total;
}

// counts were scanned in place (inclusive) in c[1 .. nq]; c[0] = 0.  Emit int64 row_splits and the
// (total, longest row) pair the host needs to size the ragged / dense result.
__global__ void radius_splits(const int* __restrict__ c, int64_t nq, int64_t* __restrict__ row_splits,
                              unsigned long long* __restrict__ stats) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long len = 0ull;
    if (i <= nq) row_splits[i] = (int64_t)c[i];
    if (i < nq) {
        // the scan runs in int32: 2^31 neighbours or more wrap it, which shows as a DECREASE somewhere along the scanned
        // counts (every per-query count is < 2^31).  Flag it -- stats[1] = 2^63: the host sees an impossible longest row and
        // raises instead of sizing its buffers from a wrapped total.
        if (c[i + 1] < c[i]) atomicMax(&stats[1], 1ull << 63);
        else len = (unsigned long long)(c[i + 1] - c[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {       // wave max, one atomic per wave
        unsigned long long other = __shfl_xor(len, o);
        len = other > len ? other : len;
    }
    if ((threadIdx.x & 63) == 0 && len > 0ull) atomicMax(&stats[1], len);
    if (i == nq) stats[0] = (unsigned long long)c[nq];
}

// ---- phase 2: fill ---------------------------------------------------------------------------------
// dense_cols == 0 : ragged — out_index[row_splits[t] + j]
// dense_cols  > 0 : dense  — out_index[t * dense_cols + j], truncated / padded with pad_value
__global__ void __launch_bounds__(256)
radius_fill(RadArgs A, const int64_t* __restrict__ row_splits, int index_local, int64_t dense_cols,
            int32_t pad_value, int32_t* __restrict__ out_index, float* __restrict__ out_d2, u64* spill) {
    __shared__ __attribute__((aligned(16))) u64 rows[4][RAD_LDS_ROW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 4 + w;
    if (t >= A.nq) return;                       // wave-uniform; no block barrier below
    RadQuery Q = rad_setup(A, t);
    const GridSeg g = A.G.segs[Q.s];
    rad_box(Q, g, A.r);
    const int64_t rs = row_splits[t];
    const int L = (int)(row_splits[t + 1] - rs);
    const bool in_lds = L <= RAD_LDS_ROW;
    u64* buf = in_lds ? rows[w] : (spill + rs);
    int total = 0;
    rad_scan(A, Q, g, lane, [&](bool act, const float4& c) {
        const float d2 = dist2_canon(Q.qx, Q.qy, Q.qz, c.x, c.y, c.z);
        const bool hit = act && d2 <= A.r2;
        const u64 key = ((u64)__float_as_uint(d2) << 32) | (u64)(unsigned)__float_as_int(c.w);
        const u64 m = __ballot(hit);
        if (hit) {
            const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < L) buf[pos] = key;
        }
        total += __popcll(m);
    });
    if (!in_lds) __threadfence();
    wave_sync();   // every lane's keys are in `buf` before any lane ranks
    // rank sort: keys are distinct (distinct indices), rank = number of smaller keys
    const int64_t base = index_local ? 0 : seg_begin_global(A.psegs, Q.s);
    const int n_out = dense_cols > 0 ? (int)(L < dense_cols ? L : dense_cols) : L;
    int32_t* orow = dense_cols > 0 ? out_index + t * dense_cols : out_index + rs;
    float* drow = out_d2 ? (dense_cols > 0 ? out_d2 + t * dense_cols : out_d2 + rs) : nullptr;
    if (in_lds) {
        // the keys are bit patterns of non-negative finite doubles (d2 <= r2 is a finite f32 in the high word), so their order
        // as doubles is their order as integers: v_cmp_lt_f64 issues at full rate where the 64-bit integer compare does not
        // (tools/micro/valu_rates.hip), and two keys arrive per ds_read_b128
        const double* db = reinterpret_cast<const double*>(rows[w]);
        for (int e = lane; e < L; e += 64) {
            const double key = db[e];
            int rank = 0;
            int j = 0;
            for (; j + 2 <= L; j += 2) {
                const double k0 = db[j], k1 = db[j + 1];
                rank += (k0 < key ? 1 : 0) + (k1 < key ? 1 : 0);
            }
            if (j < L) rank += db[j] < key ? 1 : 0;
            if (rank < n_out) {
                const u64 kb = (u64)__double_as_longlong(key);
                orow[rank] = (int32_t)((int64_t)(unsigned)(kb & 0xffffffffull) + base);
                if (drow) drow[rank] = __uint_as_float((unsigned)(kb >> 32));
            }
        }
    } else {
        const volatile u64* vb = buf;
        for (int e = lane; e < L; e += 64) {
            const u64 key = vb[e];
            int rank = 0;
            for (int j = 0; j < L; ++j) rank += (vb[j] < key) ? 1 : 0;
            if (rank < n_out) {
                orow[rank] = (int32_t)((int64_t)(unsigned)(key & 0xffffffffull) + base);
                if (drow) drow[rank] = __uint_as_float((unsigned)(key >> 32));
            }
        }
    }
    if (dense_cols > 0)
        for (int64_t e = n_out + lane; e < dense_cols; e += 64) {
            orow[e] = pad_value;
            if (drow) drow[e] = __uint_as_float(0x7f800000u);
        }
}

// ---- one traversal for the DENSE result (batch_neighbors of the KPConv batcher) ---------------------------------------
// The two-phase form walks the grid twice (count -> host reads the sizes -> fill): 1.4 + 1.9 ms of a 12 ms 64-sphere step
// (profiles/r03_kp_kernel_stats.csv).  For the dense [Nq, longest] matrix the only thing the host must know before it can
// allocate is the LONGEST row.  radius_gather does the whole search once -- scan, compaction, rank sort -- and parks every
// row, already in canonical order and with global indices, in a stash of `cap` entries per query next to its length;
// radius_expand then only copies rows into the [Nq, cols] matrix and pads (a coalesced stream).  A row longer than `cap`
// (or than the LDS staging row) raises a flag and the caller falls back to the two-phase entries for that search.
__global__ void __launch_bounds__(256)
radius_gather(RadArgs A, int cap, int32_t* __restrict__ stash, int* __restrict__ counts, unsigned long long* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) u64 rows[4][RAD_LDS_ROW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t t = (int64_t)blockIdx.x * 4 + w;
    if (t >= A.nq) return;                       // wave-uniform; no block barrier below
    RadQuery Q = rad_setup(A, t);
    const GridSeg g = A.G.segs[Q.s];
    const int64_t base = seg_begin_global(A.psegs, Q.s);      // (requested here, beside the grid record: it only depends on the segment)
    rad_box(Q, g, A.r);
    int total = 0;
    rad_scan(A, Q, g, lane, [&](bool act, const float4& c) {
        const float d2 = dist2_canon(Q.qx, Q.qy, Q.qz, c.x, c.y, c.z);
        const bool hit = act && d2 <= A.r2;
        const u64 key = ((u64)__float_as_uint(d2) << 32) | (u64)(unsigned)__float_as_int(c.w);
        const u64 m = __ballot(hit);
        if (hit) {
            const int pos = total + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < RAD_LDS_ROW) rows[w][pos] = key;
        }
        total += __popcll(m);
    });
    if (lane == 0) {
        counts[t] = total;
        // longest row / overflow flag: a plain read first -- after the first few waves the maximum stands and almost no
        // wave issues the atomic (one atomic per wave on one address would serialise 160 000 of them per 640 000 queries)
        if ((unsigned long long)total > *(volatile unsigned long long*)&stats[1]) atomicMax(&stats[1], (unsigned long long)total);
        if (total > cap || total > RAD_LDS_ROW) stats[0] = 1ull;
    }
    if (total > cap || total > RAD_LDS_ROW) return;
    wave_sync();
    const double* db = reinterpret_cast<const double*>(rows[w]);
    int32_t* orow = stash + t * cap;
    for (int e = lane; e < total; e += 64) {
        const double key = db[e];
        int rank = 0;
        int j = 0;
        for (; j + 2 <= total; j += 2) {
            const double k0 = db[j], k1 = db[j + 1];
            rank += (k0 < key ? 1 : 0) + (k1 < key ? 1 : 0);
        }
        if (j < total) rank += db[j] < key ? 1 : 0;
        const u64 kb = (u64)__double_as_longlong(key);
        orow[rank] = (int32_t)((int64_t)(unsigned)(kb & 0xffffffffull) + base);
    }
}

// (the 64-bit i / cols of the first version is a ~100-instruction software division per 4-byte element: below 2^31 elements the
//  index arithmetic is 32-bit, and a thread walks its elements with an incremental (row, column) instead of dividing again)
__global__ void __launch_bounds__(256)
radius_expand(const int32_t* __restrict__ stash, const int* __restrict__ counts, int64_t nq, int cap, int64_t cols,
              int32_t pad_value, int32_t* __restrict__ out) {
    const int64_t total = nq * cols;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int64_t t;
    int j;
    if (total < 0x7fffffffll) {
        const unsigned c = (unsigned)cols, q = (unsigned)i / c;
        t = q; j = (int)((unsigned)i - q * c);
    } else {
        t = i / cols; j = (int)(i - t * cols);
    }
    const int64_t dt = step / cols;                  // one division per thread, not per element
    const int dj = (int)(step - dt * cols);
    for (; i < total; i += step) {
        out[i] = j < counts[t] ? stash[t * cap + j] : pad_value;
        t += dt; j += dj;
        if (j >= (int)cols) { j -= (int)cols; ++t; }
    }
}

// ---- ragged_to_dense -----------------------------------------------------------------------------
// element = `elem` 4-byte words; out[r][c] = values[rs[r] + c] for c < min(len, cols), else default
__global__ void ragged_to_dense_k(const uint32_t* __restrict__ values, const int64_t* __restrict__ rs, int64_t rows,
                                  int64_t cols, int elem, const uint32_t* __restrict__ def, uint32_t* __restrict__ out) {
    const int64_t total = rows * cols * elem;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t w = i % elem, rc = i / elem;
        const int64_t c = rc % cols, r = rc / cols;
        const int64_t s = rs[r], len = rs[r + 1] - s;
        out[i] = c < len ? values[(s + c) * elem + w] : def[w];
    }
}

static inline size_t rad_align(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace ml3d

using namespace ml3d;

// workspace layout: [grid | counts int32 (nq + 1) | scan scratch | stats u64[2] | spill u64[total]]
static size_t rad_fixed_bytes(int64_t n_points, int64_t n_queries, int64_t batch) {
    size_t b = grid_ws_bytes(n_points, batch);
    b += rad_align(sizeof(int) * (size_t)(n_queries + 2));
    b += rad_align(sizeof(int) * (size_t)((n_queries + 1 + 1023) / 1024 + 2));
    b += rad_align(16);
    return b + 256;
}

struct RadWs {
    GridWs grid;
    int* counts;
    int* block_sums;
    unsigned long long* stats;
    u64* spill;
};

static bool rad_carve(void* ws, size_t bytes, int64_t n_points, int64_t n_queries, int64_t batch, RadWs* out) {
    if (bytes < rad_fixed_bytes(n_points, n_queries, batch)) return false;
    char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    size_t gb = grid_ws_bytes(n_points, batch);
    if (!grid_ws_carve(p, gb, n_points, batch, &out->grid)) return false;
    p += rad_align(gb);
    out->counts = (int*)p;       p += rad_align(sizeof(int) * (size_t)(n_queries + 2));
    out->block_sums = (int*)p;   p += rad_align(sizeof(int) * (size_t)((n_queries + 1 + 1023) / 1024 + 2));
    out->stats = (unsigned long long*)p; p += rad_align(16);
    out->spill = (u64*)p;
    return true;
}

extern "C" size_t ml3d_radius_workspace_bytes(int64_t n_points, int64_t n_queries, int64_t batch,
                                              int64_t total_neighbors) {
    if (n_points < 0 || n_queries < 0 || batch <= 0 || total_neighbors < 0) return 0;
    return rad_fixed_bytes(n_points, n_queries, batch) + sizeof(u64) * (size_t)total_neighbors + 256;
}

static int rad_args(const float* points, const int64_t* prs, const float* queries, const int64_t* qrs,
                    int64_t batch, int64_t n_points, int64_t n_queries, float radius) {
    if (!prs || !qrs || batch <= 0 || n_points < 0 || n_queries < 0 || !(radius >= 0.f)) return ML3D_E_INVALID;
    if (n_points > 0x7fffffffll / GRID_CAP - 4096 || n_queries > 0x7ffffff0ll) return ML3D_E_INVALID;
    if ((n_points > 0 && !points) || (n_queries > 0 && !queries)) return ML3D_E_INVALID;
    return 0;
}

extern "C" int ml3d_radius_count(const float* points, const int64_t* points_row_splits, const float* queries,
                                 const int64_t* queries_row_splits, int64_t batch, int64_t n_points,
                                 int64_t n_queries, float radius, int64_t* out_row_splits, int64_t* out_stats,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    int rc = rad_args(points, points_row_splits, queries, queries_row_splits, batch, n_points, n_queries, radius);
    if (rc) return rc;
    if (!out_row_splits || !out_stats) return ML3D_E_INVALID;
    RadWs W;
    if (!rad_carve(workspace, workspace_bytes, n_points, n_queries, batch, &W)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Segs ps = {points_row_splits, 0, 0, (int)batch};
    Segs qs = {queries_row_splits, 0, 0, (int)batch};
    if (grid_build_fixed(points, ps, W.grid, radius, st)) return ML3D_E_LAUNCH;
    (void)hipMemsetAsync(W.counts, 0, sizeof(int) * (size_t)(n_queries + 2), st);
    (void)hipMemsetAsync(out_stats, 0, 16, st);
    RadArgs A;
    A.G = grid_view(W.grid);
    A.queries = queries; A.qsegs = qs; A.psegs = ps; A.nq = n_queries;
    A.r = radius; A.r2 = radius * radius;
    if (n_queries > 0) {
        hipLaunchKernelGGL(radius_count, dim3((unsigned)((n_queries + 3) / 4)), dim3(256), 0, st, A, W.counts + 1);
        if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
        if (scan_inclusive_i32(W.counts + 1, n_queries, W.block_sums, st)) return ML3D_E_LAUNCH;
    }
    hipLaunchKernelGGL(radius_splits, dim3((unsigned)((n_queries + 1 + 255) / 256)), dim3(256), 0, st, W.counts,
                       n_queries, out_row_splits, (unsigned long long*)out_stats);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_radius_fill(const float* points, const int64_t* points_row_splits, const float* queries,
                                const int64_t* queries_row_splits, int64_t batch, int64_t n_points,
                                int64_t n_queries, float radius, const int64_t* row_splits, int64_t total_neighbors,
                                int index_local, int64_t dense_cols, int32_t pad_value, int32_t* out_index,
                                float* out_dist2, void* workspace, size_t workspace_bytes, void* spill,
                                size_t spill_bytes, void* stream) {
    int rc = rad_args(points, points_row_splits, queries, queries_row_splits, batch, n_points, n_queries, radius);
    if (rc) return rc;
    if (!row_splits || total_neighbors < 0 || dense_cols < 0) return ML3D_E_INVALID;
    if (n_queries == 0) return 0;
    if (!out_index) return ML3D_E_INVALID;
    // rows longer than the LDS buffer are sorted in a u64[total] scratch: the caller's `spill` when it hands one in (so the
    // workspace that carries the grid never has to grow or move between the two phases), else the tail of the workspace
    if (spill ? spill_bytes < sizeof(u64) * (size_t)total_neighbors + 8
              : workspace_bytes < ml3d_radius_workspace_bytes(n_points, n_queries, batch, total_neighbors))
        return ML3D_E_WORKSPACE;
    RadWs W;
    if (!rad_carve(workspace, workspace_bytes, n_points, n_queries, batch, &W)) return ML3D_E_WORKSPACE;
    if (spill) W.spill = (u64*)(((uintptr_t)spill + 7) & ~(uintptr_t)7);
    hipStream_t st = (hipStream_t)stream;
    RadArgs A;
    A.G = grid_view(W.grid);
    A.queries = queries;
    A.qsegs = Segs{queries_row_splits, 0, 0, (int)batch};
    A.psegs = Segs{points_row_splits, 0, 0, (int)batch};
    A.nq = n_queries;
    A.r = radius; A.r2 = radius * radius;
    hipLaunchKernelGGL(radius_fill, dim3((unsigned)((n_queries + 3) / 4)), dim3(256), 0, st, A, row_splits, index_local,
                       dense_cols, pad_value, out_index, out_dist2, W.spill);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

static size_t rad_dense_fixed_bytes(int64_t n_points, int64_t n_queries, int64_t batch) {
    return rad_align(grid_ws_bytes(n_points, batch)) + rad_align(sizeof(int) * (size_t)(n_queries + 2)) + rad_align(16) + 256;
}

extern "C" size_t ml3d_radius_dense_workspace_bytes(int64_t n_points, int64_t n_queries, int64_t batch, int cap) {
    if (n_points < 0 || n_queries < 0 || batch <= 0 || cap <= 0 || cap > RAD_LDS_ROW) return 0;
    return rad_dense_fixed_bytes(n_points, n_queries, batch) + rad_align(sizeof(int32_t) * (size_t)n_queries * (size_t)cap) + 256;
}

struct RadDenseWs { GridWs grid; int* counts; unsigned long long* stats; int32_t* stash; };

static bool rad_dense_carve(void* ws, size_t bytes, int64_t n_points, int64_t n_queries, int64_t batch, int cap, RadDenseWs* out) {
    if (bytes < ml3d_radius_dense_workspace_bytes(n_points, n_queries, batch, cap)) return false;
    char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    const size_t gb = grid_ws_bytes(n_points, batch);
    if (!grid_ws_carve(p, gb, n_points, batch, &out->grid)) return false;
    p += rad_align(gb);
    out->counts = (int*)p;                  p += rad_align(sizeof(int) * (size_t)(n_queries + 2));
    out->stats = (unsigned long long*)p;    p += rad_align(16);
    out->stash = (int32_t*)p;
    return true;
}

extern "C" int ml3d_radius_dense_gather(const float* points, const int64_t* points_row_splits, const float* queries,
                                        const int64_t* queries_row_splits, int64_t batch, int64_t n_points,
                                        int64_t n_queries, float radius, int cap, int reuse_grid, int64_t* out_stats,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    int rc = rad_args(points, points_row_splits, queries, queries_row_splits, batch, n_points, n_queries, radius);
    if (rc) return rc;
    if (!out_stats || cap <= 0 || cap > RAD_LDS_ROW) return ML3D_E_INVALID;
    RadDenseWs W;
    if (!rad_dense_carve(workspace, workspace_bytes, n_points, n_queries, batch, cap, &W)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Segs ps = {points_row_splits, 0, 0, (int)batch};
    Segs qs = {queries_row_splits, 0, 0, (int)batch};
    // reuse_grid: the workspace already holds the grid of THESE points at THIS radius (built by an earlier gather whose expand
    // has been enqueued): the conv and the pool search of a KPConv layer share supports and radius (concat_batcher.py:234-262)
    if (!reuse_grid && grid_build_fixed(points, ps, W.grid, radius, st)) return ML3D_E_LAUNCH;
    (void)hipMemsetAsync(out_stats, 0, 16, st);
    if (n_queries == 0) return 0;
    RadArgs A;
    A.G = grid_view(W.grid);
    A.queries = queries; A.qsegs = qs; A.psegs = ps; A.nq = n_queries;
    A.r = radius; A.r2 = radius * radius;
    hipLaunchKernelGGL(radius_gather, dim3((unsigned)((n_queries + 3) / 4)), dim3(256), 0, st, A, cap, W.stash, W.counts,
                       (unsigned long long*)out_stats);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_radius_dense_expand(int64_t n_points, int64_t n_queries, int64_t batch, int cap, int64_t dense_cols,
                                        int32_t pad_value, int32_t* out_index, void* workspace, size_t workspace_bytes,
                                        void* stream) {
    if (n_points < 0 || n_queries < 0 || batch <= 0 || dense_cols < 0 || dense_cols > cap) return ML3D_E_INVALID;
    if (n_queries == 0 || dense_cols == 0) return 0;
    if (!out_index) return ML3D_E_INVALID;
    RadDenseWs W;
    if (!rad_dense_carve(workspace, workspace_bytes, n_points, n_queries, batch, cap, &W)) return ML3D_E_WORKSPACE;
    const int64_t total = n_queries * dense_cols;
    const unsigned nb = (unsigned)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144);
    hipLaunchKernelGGL(radius_expand, dim3(nb), dim3(256), 0, (hipStream_t)stream, W.stash, W.counts, n_queries, cap, dense_cols,
                       pad_value, out_index);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_ragged_to_dense(const void* values, const int64_t* row_splits, int64_t rows, int64_t out_cols,
                                    int64_t elem_bytes, const void* default_value, void* out, void* stream) {
    if (rows < 0 || out_cols < 0 || elem_bytes <= 0 || (elem_bytes & 3)) return ML3D_E_INVALID;
    if (rows == 0 || out_cols == 0) return 0;
    if (!row_splits || !default_value || !out) return ML3D_E_INVALID;
    int64_t total = rows * out_cols * (elem_bytes / 4);
    unsigned nb = (unsigned)((total + 255) / 256 < 65535 * 4 ? (total + 255) / 256 : 65535 * 4);
    hipLaunchKernelGGL(ragged_to_dense_k, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)values,
                       row_splits, rows, out_cols, (int)(elem_bytes / 4), (const uint32_t*)default_value, (uint32_t*)out);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}
