// grid.hip — build of the counting-sorted uniform grid (see grid.h).
//
// HBM traffic per build, per point: 12 B (bbox) + 12 B (occupancy probe) +
// 12 B (histogram) + 12 B + 16 B (scatter read + sorted write); the cell table
// adds <= 2 * 4 * GRID_CAP bytes per point for zero + scan.  Everything is a
// coalesced stream; the only scattered accesses are the 4-byte atomics on the
// cell table, which stays L2-resident (<= 32 B per point).
#include "grid.h"

#include <math.h>

namespace ml3d {

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

static void grid_sizes(int64_t n_total, int64_t batch, int64_t* total_cells, int64_t* bitmap_words,
                       int64_t* n_blocks) {
    *total_cells = (int64_t)GRID_CAP * n_total + (int64_t)GRID_SLACK * batch;
    *bitmap_words = (*total_cells + 31) / 32 + batch;
    *n_blocks = (*total_cells + 2 + 1023) / 1024 + 1;
}

size_t grid_ws_bytes(int64_t n_total, int64_t batch) {
    int64_t tc, bw, nb;
    grid_sizes(n_total, batch, &tc, &bw, &nb);
    size_t b = 0;
    b += align_up(sizeof(GridSeg) * (size_t)batch);
    b += align_up(sizeof(unsigned) * 6 * (size_t)batch);
    b += align_up(sizeof(unsigned) * GRID_LEVELS * (size_t)batch);
    b += align_up(sizeof(unsigned) * GRID_LEVELS * (size_t)bw);
    b += align_up(sizeof(int) * (size_t)(tc + 2));
    b += align_up(sizeof(int) * (size_t)nb);
    // (+ GRID_SORTED_SLACK entries behind the cell-sorted array: the k-NN scan reads a few entries past a run's end, knn.hip scan_run)
    b += align_up(sizeof(float4) * (size_t)((n_total > 0 ? n_total : 1) + GRID_SORTED_SLACK));
    return b + 256;
}

bool grid_ws_carve(void* ws, size_t bytes, int64_t n_total, int64_t batch, GridWs* out) {
    if (bytes < grid_ws_bytes(n_total, batch)) return false;
    int64_t tc, bw, nb;
    grid_sizes(n_total, batch, &tc, &bw, &nb);
    char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    out->segs = (GridSeg*)p;      p += align_up(sizeof(GridSeg) * (size_t)batch);
    out->bbox = (unsigned*)p;     p += align_up(sizeof(unsigned) * 6 * (size_t)batch);
    out->occ = (unsigned*)p;      p += align_up(sizeof(unsigned) * GRID_LEVELS * (size_t)batch);
    out->bitmap = (unsigned*)p;   p += align_up(sizeof(unsigned) * GRID_LEVELS * (size_t)bw);
    out->cells = (int*)p;         p += align_up(sizeof(int) * (size_t)(tc + 2));
    out->block_sums = (int*)p;    p += align_up(sizeof(int) * (size_t)nb);
    out->sorted = (float4*)p;
    out->total_cells = tc;
    out->bitmap_words = bw;
    out->n_total = n_total;
    out->batch = (int)batch;
    return true;
}

// ---- K1: bounding box per segment -------------------------------------------------------------
__global__ void grid_bbox_init(unsigned* bbox, unsigned* occ, int batch) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    for (int a = 0; a < 3; ++a) { bbox[6 * s + a] = 0xffffffffu; bbox[6 * s + 3 + a] = 0u; }
    for (int l = 0; l < GRID_LEVELS; ++l) occ[GRID_LEVELS * s + l] = 0u;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// segment of a packed row, found by the whole wave at once when the batch is small: lane i looks at splits[i + 1] (ONE
// coalesced load, a ballot and a popcount) instead of a binary search of log2(batch) DEPENDENT global loads -- a 1024-point
// workgroup of the box / occupancy passes did two such searches before its first useful load (10 latencies for 32 items:
// most of the 35-60 us these launches took in the KPConv batch build)
// (first, last) rows of a workgroup located from ONE load of the splits: no second dependent access for the item starts
__device__ __forceinline__ void seg_locate_wave2(const Segs& S, int64_t r0, int64_t r1, int& s0, int64_t& l0, int& s1, int64_t& l1) {
    if (!S.splits || S.batch > 63) { seg_locate(S, r0, s0, l0); seg_locate(S, r1, s1, l1); return; }
    const int lane = threadIdx.x & 63;
    const int64_t e = lane <= S.batch ? S.splits[lane] : (int64_t)0x7fffffffffffffffll;      // lane i: START of item i (i = batch: total)
    // item of a row = (number of starts <= row) - 1, clamped to the last item; empty items share their start with the next
    const int c0 = __popcll(__ballot(lane <= S.batch && e <= r0)), c1 = __popcll(__ballot(lane <= S.batch && e <= r1));
    s0 = min(max(c0 - 1, 0), S.batch - 1);
    s1 = min(max(c1 - 1, 0), S.batch - 1);
    l0 = r0 - __shfl(e, s0);
    l1 = r1 - __shfl(e, s1);
}

__device__ __forceinline__ void seg_locate_wave(const Segs& S, int64_t packed, int& s, int64_t& local) {
    if (!S.splits || S.batch > 64) { seg_locate(S, packed, s, local); return; }
    const int lane = threadIdx.x & 63;
    const int64_t e = lane < S.batch ? S.splits[lane + 1] : (int64_t)0x7fffffffffffffffll;
    s = __popcll(__ballot(e <= packed));            // items whose end is at or before the row (empty items included)
    s = s < S.batch ? s : S.batch - 1;
    local = packed - S.splits[s];
}

__global__ void __launch_bounds__(256)
grid_bbox(const float* __restrict__ pts, Segs S, int64_t n_total, unsigned* bbox) {
    // A block covers 1024 consecutive packed points.  When they all belong to one batch item (the
    // common case) the block reduces in registers/LDS and issues 6 atomics in total; a block that
    // straddles an item boundary falls back to per-thread atomics.
    __shared__ float red[4][6];
    const int64_t first = (int64_t)blockIdx.x * 1024;
    const int64_t last = first + 1023 < n_total ? first + 1023 : n_total - 1;
    int s0, s1; int64_t l0, l1;
    seg_locate_wave2(S, first, last, s0, l0, s1, l1);
    const bool one_item = (s0 == s1);            // block-uniform
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    if (!one_item) {
        // straddling block (batched calls with short items): every wave-step covers 64 consecutive points —
        // when they share one item the wave reduces and issues 6 atomics, only the wave-step that contains
        // an item boundary falls back to per-lane atomics
        int64_t i = first + threadIdx.x;
        for (int it = 0; it < 4; ++it, i += 256) {
            const bool valid = i < n_total;
            int s = -1; int64_t local = 0;
            float x = 0.f, y = 0.f, z = 0.f;
            // the wave-step's first row is located by the whole wave at once; the other 63 rows start from that item and
            // step over the (rare) boundaries instead of 64 binary searches of dependent loads -- the few workgroups
            // that straddle items used to be the launch's critical path (35 us for 320 k points)
            int sa; int64_t la;
            seg_locate_wave(S, (first + it * 256 + (threadIdx.x & ~63)) < n_total ? first + it * 256 + (threadIdx.x & ~63) : n_total - 1,
                            sa, la);
            if (valid) {
                s = sa;
                if (S.splits) {
                    while (s + 1 < S.batch && i >= S.splits[s + 1]) ++s;
                    local = i - S.splits[s];
                } else {
                    seg_locate(S, i, s, local);
                }
                const float* p = pts + 3 * (seg_begin_global(S, s) + local);
                x = p[0]; y = p[1]; z = p[2];
            }
            const int s_first = __builtin_amdgcn_readfirstlane(s);
            if (__all(!valid || s == s_first)) {
                if (__any(valid)) {
                    const float v[6] = {wave_min(valid ? x : 3.0e38f), wave_min(valid ? y : 3.0e38f), wave_min(valid ? z : 3.0e38f),
                                        wave_max(valid ? x : -3.0e38f), wave_max(valid ? y : -3.0e38f), wave_max(valid ? z : -3.0e38f)};
                    if ((threadIdx.x & 63) == 0 && s_first >= 0) {
                        for (int a = 0; a < 3; ++a) {
                            atomicMin(&bbox[6 * s_first + a], f2ord(v[a]));
                            atomicMax(&bbox[6 * s_first + 3 + a], f2ord(v[3 + a]));
                        }
                    }
                }
            } else {
                // several items in one wave-step (short items: the coarse levels of a KPConv batch): the lanes of an item are
                // contiguous, so a SEGMENTED reduction (shuffle down while the partner is in the same item) leaves each
                // item's box in its first lane -- 6 atomics per item and wave-step, not 6 per point (7800 atomics on 192
                // words made the 1300-point levels as slow as the 320 k-point one)
                const int lane = threadIdx.x & 63;
                const int sv = valid ? s : -2 - lane;               // idle lanes: items of their own
                float lo[3] = {x, y, z}, hi[3] = {x, y, z};
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int so = __shfl_down(sv, o);
                    float l2[3], h2[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a) { l2[a] = __shfl_down(lo[a], o); h2[a] = __shfl_down(hi[a], o); }
                    // contiguous items: if the lane `o` further on is in my item, so is everything in between
                    if (lane + o < 64 && so == sv) {
#pragma unroll
                        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], l2[a]); hi[a] = fmaxf(hi[a], h2[a]); }
                    }
                }
                const int sp = __shfl_up(sv, 1);
                if (valid && (lane == 0 || sp != sv)) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        atomicMin(&bbox[6 * s + a], f2ord(lo[a]));
                        atomicMax(&bbox[6 * s + 3 + a], f2ord(hi[a]));
                    }
                }
            }
        }
        return;
    }
    {
        // the item's rows are contiguous: row (first + j) of the block is point (global start of the item + l0 + j).  The four
        // strided loads per lane are requested together (clamped index, masked use): one exposed latency, not four
        const float* base = pts + 3 * (S.splits ? first : (int64_t)s0 * S.stride + l0);
        float x[4], y[4], z[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int64_t j = min((int64_t)(threadIdx.x + 256 * it), last - first);
            x[it] = base[3 * j]; y[it] = base[3 * j + 1]; z[it] = base[3 * j + 2];
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            mn[0] = fminf(mn[0], x[it]); mx[0] = fmaxf(mx[0], x[it]);
            mn[1] = fminf(mn[1], y[it]); mx[1] = fmaxf(mx[1], y[it]);
            mn[2] = fminf(mn[2], z[it]); mx[2] = fmaxf(mx[2], z[it]);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo = wave_min(mn[a]), hi = wave_max(mx[a]);
        if (lane == 0) { red[wv][a] = lo; red[wv][3 + a] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        int a = threadIdx.x;
        float v = red[0][a];
        for (int w = 1; w < 4; ++w) v = a < 3 ? fminf(v, red[w][a]) : fmaxf(v, red[w][a]);
        if (a < 3) atomicMin(&bbox[6 * s0 + a], f2ord(v));
        else atomicMax(&bbox[6 * s0 + a], f2ord(v));
    }
}

// ---- K2: finest probe grid per segment --------------------------------------------------------
__device__ __forceinline__ bool dims_fit(const float ext[3], float c, int64_t cap, int dims[3]) {
    int64_t prod = 1;
    for (int a = 0; a < 3; ++a) {
        float q = ext[a] / c;
        if (!(q < 1.0e6f)) return false;
        dims[a] = (int)q + 1;
        prod *= dims[a];
        if (prod > cap) return false;
    }
    return true;
}

__global__ void grid_setup0(Segs S, const unsigned* bbox, GridSeg* segs, int batch) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    GridSeg g;
    int64_t n = seg_len(S, s);
    int64_t pb = seg_begin_packed(S, s);
    g.n = (int)n;
    g.sorted_base = (int)pb;
    g.cell_base = (int)(GRID_CAP * pb + (int64_t)GRID_SLACK * s);
    float ext[3], amax = 0.f, emax = 0.f;
    for (int a = 0; a < 3; ++a) {
        float lo = n > 0 ? ord2f(bbox[6 * s + a]) : 0.f;
        float hi = n > 0 ? ord2f(bbox[6 * s + 3 + a]) : 0.f;
        g.lo[a] = lo;
        ext[a] = hi - lo;
        emax = fmaxf(emax, ext[a]);
        amax = fmaxf(amax, fmaxf(fabsf(lo), fabsf(hi)));
    }
    g.margin = 4.0e-6f * (amax + emax) + 1.0e-30f;
    int64_t cap = (int64_t)GRID_CAP * n + GRID_SLACK;
    float c = emax > 0.f ? emax : 1.0f;
    int d[3] = {1, 1, 1}, dn[3];
    if (emax > 0.f) {
        // halve while the dense table still fits: ends within 2x of the finest admissible size
        dims_fit(ext, c, cap, d);
        for (int it = 0; it < 40; ++it) {
            if (!dims_fit(ext, c * 0.5f, cap, dn)) break;
            c *= 0.5f;
            d[0] = dn[0]; d[1] = dn[1]; d[2] = dn[2];
        }
    }
    g.c0 = c;
    g.c = c;
    g.inv_c = 1.0f / c;
    for (int a = 0; a < 3; ++a) { g.dims0[a] = d[a]; g.dims[a] = d[a]; g.ext[a] = ext[a]; }
    g.dim_est = 2.5f;
    // the probe is a statistic of the cloud's density: a sixteenth (a quarter) of a large (medium) cloud is plenty
    g.probe_stride = n >= 32768 ? 16 : (n >= 8192 ? 4 : 1);
    segs[s] = g;
}

// ---- K3: occupancy probe at GRID_LEVELS power-of-two resolutions ------------------------------
__global__ void __launch_bounds__(256)
grid_occupancy(const float* __restrict__ pts, Segs S, int64_t n_total, const GridSeg* __restrict__ segs,
               unsigned* bitmap, int64_t bitmap_words, unsigned* occ, int dense_stride) {
    // one bit per (level, cell); the number of bits a block newly sets is summed per block (ballot
    // + popcount) so the per-item counters see one atomic per block and level, not one per point.
    // dense_stride > 0 (uniform batches whose items all probe every dense_stride-th point: grid_build): thread t IS probe t -- item
    // t / per, point (t % per) * dense_stride -- instead of one thread per point of which 15 in 16 leave at once.
    __shared__ unsigned cnt[GRID_LEVELS];
    const int64_t first = (int64_t)blockIdx.x * blockDim.x;
    int s0, s1; int64_t l0 = 0, l1;
    int64_t per = 0;
    if (dense_stride > 0) {
        per = (S.n_uniform + dense_stride - 1) / dense_stride;
        const int64_t total = per * S.batch;
        const int64_t last = first + blockDim.x - 1 < total ? first + blockDim.x - 1 : total - 1;
        s0 = (int)(first / per); s1 = (int)(last / per);
    } else {
        const int64_t last = first + blockDim.x - 1 < n_total ? first + blockDim.x - 1 : n_total - 1;
        seg_locate_wave(S, first, s0, l0);
        seg_locate_wave(S, last, s1, l1);
    }
    const bool one_item = (s0 == s1);            // block-uniform
    if (threadIdx.x < GRID_LEVELS) cnt[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t i = first + threadIdx.x;
    bool valid;
    int s = s0; int64_t local = l0 + threadIdx.x;
    if (dense_stride > 0) {
        s = (int)(i / per);
        local = (i - (int64_t)s * per) * dense_stride;
        valid = s < S.batch && local < S.n_uniform;
        if (!valid) s = s0;
    } else {
        valid = i < n_total;
        if (valid && !one_item) seg_locate(S, i, s, local);
    }
    int cx = 0, cy = 0, cz = 0;
    const GridSeg* g = &segs[s];
    valid = valid && (local % g->probe_stride == 0);
    if (valid) {
        const float* p = pts + 3 * (seg_begin_global(S, s) + local);
        float inv = 1.0f / g->c0;
        cx = cell_coord(p[0], g->lo[0], inv, g->dims0[0]);
        cy = cell_coord(p[1], g->lo[1], inv, g->dims0[1]);
        cz = cell_coord(p[2], g->lo[2], inv, g->dims0[2]);
    }
    const int64_t bit_base = (int64_t)g->cell_base + 32 * (int64_t)s;  // word-aligned per segment
    for (int l = 0; l < GRID_LEVELS; ++l) {
        bool fresh = false;
        if (valid) {
            int dx = ((g->dims0[0] - 1) >> l) + 1, dy = ((g->dims0[1] - 1) >> l) + 1;
            int64_t id = (cx >> l) + (int64_t)dx * ((cy >> l) + (int64_t)dy * (cz >> l));
            int64_t bit = bit_base + id;
            unsigned m = 1u << (unsigned)(bit & 31);
            unsigned old = atomicOr(&bitmap[(int64_t)l * bitmap_words + (bit >> 5)], m);
            fresh = !(old & m);
        }
        if (one_item) {
            unsigned long long b = __ballot(fresh);
            if ((threadIdx.x & 63) == 0 && b) atomicAdd(&cnt[l], (unsigned)__popcll(b));
        } else if (fresh) {
            atomicAdd(&occ[GRID_LEVELS * s + l], 1u);
        }
    }
    __syncthreads();
    if (one_item && threadIdx.x < GRID_LEVELS && cnt[threadIdx.x])
        atomicAdd(&occ[GRID_LEVELS * s0 + threadIdx.x], cnt[threadIdx.x]);
}

// ---- K4: final cell size from the occupancy curve ----------------------------------------------
// Model: points fall into the occupied cells of a level like a Poisson process of intensity lam per
// cell, so the mean occupancy of NON-EMPTY cells is m(lam) = lam / (1 - exp(-lam)).  The probe
// measured m at 5 resolutions on every probe_stride-th point; invert to lam per level, read the
// local dimension D off the growth of lam between levels, and pick the cell size whose full-data
// intensity gives the requested mean occupancy.
__device__ __forceinline__ float occ_of_lam(float lam) { return lam / (1.0f - expf(-lam)); }
__device__ __forceinline__ float lam_of_occ(float m) {
    if (m <= 1.0005f) return 1.0e-3f;
    float lo = 1.0e-3f, hi = 64.0f;
    if (m >= hi) return m;
    for (int it = 0; it < 40; ++it) {
        float mid = 0.5f * (lo + hi);
        if (occ_of_lam(mid) < m) lo = mid; else hi = mid;
    }
    return 0.5f * (lo + hi);
}

__device__ __forceinline__ void finish_grid(GridSeg& g, float c, int64_t cap) {
    float emax = fmaxf(g.ext[0], fmaxf(g.ext[1], g.ext[2]));
    int d[3] = {1, 1, 1};
    if (emax > 0.f) {
        for (int it = 0; it < 200; ++it) {
            if (dims_fit(g.ext, c, cap, d)) break;
            c *= 1.1f;
        }
    } else {
        c = 1.0f;
    }
    g.c = c;
    g.inv_c = 1.0f / c;
    for (int a = 0; a < 3; ++a) g.dims[a] = d[a];
}

__global__ void grid_setup1(const unsigned* occ, GridSeg* segs, int batch, float target) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    GridSeg g = segs[s];
    int64_t n = g.n;
    float emax = fmaxf(g.ext[0], fmaxf(g.ext[1], g.ext[2]));
    int64_t cap = (int64_t)GRID_CAP * n + GRID_SLACK;
    float c = g.c0;
    if (n > 0 && emax > 0.f) {
        const float n_probe = (float)((n + g.probe_stride - 1) / g.probe_stride);
        const float lam_goal = lam_of_occ(target) / (float)g.probe_stride;
        float lam_prev = 0.f, lam_cur = 0.f;
        int lsel = -1;
        for (int l = 0; l < GRID_LEVELS; ++l) {
            unsigned cnt = occ[GRID_LEVELS * s + l];
            lam_cur = lam_of_occ(n_probe / (float)(cnt > 0u ? cnt : 1u));
            if (lam_cur >= lam_goal) { lsel = l; break; }
            lam_prev = lam_cur;
        }
        if (lsel == 0) {
            c = g.c0;
        } else if (lsel > 0) {
            float D = fminf(fmaxf(log2f(lam_cur / lam_prev), 1.0f), 3.0f);
            g.dim_est = D;
            c = g.c0 * (float)(1 << (lsel - 1)) * powf(lam_goal / lam_prev, 1.0f / D);
        } else {
            c = g.c0 * (float)(1 << (GRID_LEVELS - 1));
            if (n <= 64) c = emax * 2.0f;  // tiny item: one cell, i.e. brute force
        }
    }
    finish_grid(g, c, cap);
    segs[s] = g;
}

// grid of a thinner subset of the parent's points: same box, c scaled by (n_parent / n)^(1/D)
__global__ void grid_setup_derived(Segs S, const GridSeg* parent, GridSeg* segs, int batch) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    GridSeg g = parent[s];
    int64_t n = seg_len(S, s);
    int64_t pb = seg_begin_packed(S, s);
    float ratio = n > 0 ? (float)g.n / (float)n : 1.0f;
    float c = g.c * powf(fmaxf(ratio, 1.0f), 1.0f / g.dim_est);
    g.n = (int)n;
    g.sorted_base = (int)pb;
    g.cell_base = (int)(GRID_CAP * pb + (int64_t)GRID_SLACK * s);
    if (n <= 64) c = fmaxf(g.ext[0], fmaxf(g.ext[1], g.ext[2])) * 2.0f;
    finish_grid(g, c, (int64_t)GRID_CAP * n + GRID_SLACK);
    segs[s] = g;
}

// ---- K5: histogram ------------------------------------------------------------------------------
__device__ __forceinline__ int point_cell(const GridSeg* g, float x, float y, float z) {
    int cx = cell_coord(x, g->lo[0], g->inv_c, g->dims[0]);
    int cy = cell_coord(y, g->lo[1], g->inv_c, g->dims[1]);
    int cz = cell_coord(z, g->lo[2], g->inv_c, g->dims[2]);
    return g->cell_base + cx + g->dims[0] * (cy + g->dims[1] * cz);
}

__global__ void grid_hist(const float* __restrict__ pts, Segs S, int64_t n_total,
                          const GridSeg* __restrict__ segs, int* cells) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    int s; int64_t local;
    seg_locate(S, i, s, local);
    const float* p = pts + 3 * (seg_begin_global(S, s) + local);
    int cid = point_cell(&segs[s], p[0], p[1], p[2]);
    atomicAdd(&cells[cid + 2], 1);
}

// ---- K6: in-place inclusive scan of a[0 .. n) ---------------------------------------------------
// A workgroup owns SCAN_TILE = 4096 consecutive elements, a wave 1024 of them as 16 rows of 64: every load / store is
// one coalesced 256-byte row, the row is scanned across the lanes with shuffles and the running total rides along in a
// register.  Three launches (tile sums -> scan of the sums in ONE 1024-thread workgroup -> final), 12 bytes of traffic
// per element; the middle launch handles 7.5 k sums for the 30 M-cell table of a 64-frame level-0 grid in one pass
// (the first version walked them 256 at a time: 0.56 ms of a single CU).
constexpr int SCAN_TILE = 4096;


__global__ void __launch_bounds__(256) scan_block_sums(const int* __restrict__ a, int64_t n, int* block_sums) {
    __shared__ int ws[4];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    int v = 0;
#pragma unroll
    for (int j = 0; j < SCAN_TILE / 256; ++j) {
        int64_t i = base + j * 256 + threadIdx.x;
        if (i < n) v += a[i];
    }
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__global__ void __launch_bounds__(1024) scan_sums(int* block_sums, int64_t nb) {
    // ONE workgroup: thread t owns a contiguous slice of the sums, slices are combined by a wave + workgroup scan
    __shared__ int wsum[16];
    const int t = threadIdx.x;
    const int64_t per = (nb + 1023) / 1024;
    const int64_t lo = (int64_t)t * per, hi = lo + per < nb ? lo + per : nb;
    int s = 0;
    for (int64_t i = lo; i < hi; ++i) s += block_sums[i];
    int incl = wave_inclusive_scan(s);
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    int carry = 0;
    for (int w = 0; w < (t >> 6); ++w) carry += wsum[w];
    int run = carry + incl - s;                     // exclusive prefix of this thread's slice
    for (int64_t i = lo; i < hi; ++i) {
        int v = block_sums[i];
        block_sums[i] = run;
        run += v;
    }
}

__global__ void __launch_bounds__(256) scan_final(int* a, int64_t n, const int* __restrict__ block_sums) {
    __shared__ int ws[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)wv * 1024;
    int v[16], tot = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int64_t i = base + r * 64 + lane;
        v[r] = i < n ? a[i] : 0;
        tot += v[r];
    }
    tot = wave_sum(tot);
    if (lane == 0) ws[wv] = tot;
    __syncthreads();
    int carry = block_sums[blockIdx.x];
    for (int w = 0; w < wv; ++w) carry += ws[w];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int incl = wave_inclusive_scan(v[r]) + carry;
        int64_t i = base + r * 64 + lane;
        if (i < n) a[i] = incl;
        carry = __shfl(incl, 63);
    }
}

// short arrays (<= SCAN_ONE_MAX elements: the histograms of a small radix sort, the counts / cells of the coarse KPConv levels)
// in ONE launch of ONE 1024-thread workgroup: wave w owns a contiguous slice (rows of 64), sums it, the 16 slice sums meet in
// LDS, then the wave scans its slice with the carry of the slices before it.  The three-launch form costs ~15 us of launch
// latency per call however small n is, and a KPConv batch build issues ~60 scans per step.
constexpr int SCAN_ONE_MAX = 32768;

__global__ void __launch_bounds__(1024) scan_one(int* a, int n) {
    __shared__ int ws[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int rows = (n + 1023) / 1024;                 // rows of 64 per wave
    const int base = wv * rows * 64;
    int tot = 0;
    for (int r = 0; r < rows; ++r) {
        const int i = base + r * 64 + lane;
        if (i < n) tot += a[i];
    }
    tot = wave_sum(tot);
    if (lane == 0) ws[wv] = tot;
    __syncthreads();
    int carry = 0;
    for (int w = 0; w < wv; ++w) carry += ws[w];
    for (int r = 0; r < rows; ++r) {
        const int i = base + r * 64 + lane;
        const int incl = wave_inclusive_scan(i < n ? a[i] : 0) + carry;
        if (i < n) a[i] = incl;
        carry = __shfl(incl, 63);
    }
}

// ---- K7: scatter into the cell-sorted float4 array ---------------------------------------------
__global__ void grid_scatter(const float* __restrict__ pts, Segs S, int64_t n_total,
                             const GridSeg* __restrict__ segs, int* cells, float4* sorted) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    int s; int64_t local;
    seg_locate(S, i, s, local);
    const float* p = pts + 3 * (seg_begin_global(S, s) + local);
    float x = p[0], y = p[1], z = p[2];
    int cid = point_cell(&segs[s], x, y, z);
    int pos = atomicAdd(&cells[cid + 1], 1);
    sorted[pos] = make_float4(x, y, z, __int_as_float((int)local));
}

// Zero fill of the cell tables as a KERNEL, not hipMemsetAsync: on ROCm 7.2 a memset node of a captured HIP graph (the model-class
// patch loop replays this build as a graph) did not reliably clear the table on the second and later replays -- the stale
// histogram sent grid_scatter out of bounds ("write access to a read-only page", gpurun r5g).  16-byte stores, tail by words;
// both pointers in the workspace are 256-byte aligned (grid_ws_carve) and the lengths multiples of 4.
__global__ void __launch_bounds__(256) grid_zero(uint4* __restrict__ p, size_t n16, uint32_t* __restrict__ tail, int ntail) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (size_t j = i; j < n16; j += step) p[j] = make_uint4(0u, 0u, 0u, 0u);
    if (i < (size_t)ntail) tail[i] = 0u;
}

void zero_async(void* ptr, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return;
    if ((((uintptr_t)ptr) & 15) || (bytes & 3)) { (void)hipMemsetAsync(ptr, 0, bytes, stream); return; }
    const size_t n16 = bytes / 16;
    const int ntail = (int)((bytes - 16 * n16) / 4);
    const size_t want = (n16 + 255) / 256;
    const unsigned nb = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(grid_zero, dim3(nb), dim3(256), 0, stream, (uint4*)ptr, n16, (uint32_t*)((char*)ptr + 16 * n16), ntail);
}

#define ML3D_LAUNCH_CHECK()                         \
    do {                                            \
        if (hipGetLastError() != hipSuccess) return -3; \
    } while (0)

int scan_inclusive_i32(int* a, int64_t n, int* block_sums, hipStream_t stream) {
    if (n <= 0) return 0;
    if (n <= SCAN_ONE_MAX) {
        hipLaunchKernelGGL(scan_one, dim3(1), dim3(1024), 0, stream, a, (int)n);
        ML3D_LAUNCH_CHECK();
        return 0;
    }
    int sbk = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(scan_block_sums, dim3(sbk), dim3(256), 0, stream, a, n, block_sums);
    ML3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_sums, dim3(1), dim3(1024), 0, stream, block_sums, (int64_t)sbk);
    ML3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(scan_final, dim3(sbk), dim3(256), 0, stream, a, n, block_sums);
    ML3D_LAUNCH_CHECK();
    return 0;
}

static int grid_sort(const float* points, Segs S, const GridWs& ws, hipStream_t stream) {
    int64_t n = ws.n_total;
    int nb = (int)((n + 255) / 256);
    if (n > 0) {
        hipLaunchKernelGGL(grid_hist, dim3(nb), dim3(256), 0, stream, points, S, n, ws.segs, ws.cells);
        ML3D_LAUNCH_CHECK();
        if (scan_inclusive_i32(ws.cells + 2, ws.total_cells, ws.block_sums, stream)) return -3;
        hipLaunchKernelGGL(grid_scatter, dim3(nb), dim3(256), 0, stream, points, S, n, ws.segs, ws.cells,
                           ws.sorted);
        ML3D_LAUNCH_CHECK();
    }
    return 0;
}


int grid_build(const float* points, Segs S, const GridWs& ws, float target_occ, hipStream_t stream) {
    int64_t n = ws.n_total;
    int B = ws.batch;
    if (B <= 0) return 0;
    if (target_occ <= 0.f) target_occ = 4.0f;
    int sb = (B + 63) / 64;
    // (bitmap and cell table are neighbours in the workspace, grid_ws_carve: one fill instead of two)
    zero_async(ws.bitmap, (size_t)((char*)(ws.cells + ws.total_cells + 2) - (char*)ws.bitmap), stream);
    hipLaunchKernelGGL(grid_bbox_init, dim3(sb), dim3(64), 0, stream, ws.bbox, ws.occ, B);
    ML3D_LAUNCH_CHECK();
    if (n > 0) {
        int nb4 = (int)((n + 1023) / 1024);
        hipLaunchKernelGGL(grid_bbox, dim3(nb4), dim3(256), 0, stream, points, S, n, ws.bbox);
        ML3D_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(grid_setup0, dim3(sb), dim3(64), 0, stream, S, ws.bbox, ws.segs, B);
    ML3D_LAUNCH_CHECK();
    int nb = (int)((n + 255) / 256);
    if (n > 0) {
        // (the probe stride is a function of the item size alone -- grid_setup0 -- so a uniform batch knows it on the host)
        const int dense = (!S.splits && S.n_uniform >= 32768) ? 16 : 0;
        const int nbo = dense ? (int)((((S.n_uniform + dense - 1) / dense) * (int64_t)S.batch + 255) / 256) : nb;
        hipLaunchKernelGGL(grid_occupancy, dim3(nbo), dim3(256), 0, stream, points, S, n, ws.segs, ws.bitmap,
                           ws.bitmap_words, ws.occ, dense);
        ML3D_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(grid_setup1, dim3(sb), dim3(64), 0, stream, ws.occ, ws.segs, B, target_occ);
    ML3D_LAUNCH_CHECK();
    return grid_sort(points, S, ws, stream);
}

// fixed cell size: a fixed-radius search wants cell ~ radius
__global__ void grid_setup_fixed(GridSeg* segs, int batch, float cell) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    GridSeg g = segs[s];
    float c = cell;
    if (!(c > 0.f)) c = g.c0;
    if (c < g.c0 * 0.5f) c = g.c0 * 0.5f;   // never finer than the dense table admits
    finish_grid(g, c, (int64_t)GRID_CAP * g.n + GRID_SLACK);
    segs[s] = g;
}

int bbox_compute(const float* points, Segs S, int64_t n, unsigned* bbox, unsigned* occ_scratch, hipStream_t stream) {
    int B = S.batch;
    if (B <= 0) return 0;
    hipLaunchKernelGGL(grid_bbox_init, dim3((B + 63) / 64), dim3(64), 0, stream, bbox, occ_scratch, B);
    ML3D_LAUNCH_CHECK();
    if (n > 0) {
        hipLaunchKernelGGL(grid_bbox, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, points, S, n, bbox);
        ML3D_LAUNCH_CHECK();
    }
    return 0;
}

int grid_build_fixed(const float* points, Segs S, const GridWs& ws, float cell, hipStream_t stream) {
    int B = ws.batch;
    if (B <= 0) return 0;
    int sb = (B + 63) / 64;
    zero_async(ws.cells, sizeof(int) * (size_t)(ws.total_cells + 2), stream);
    if (bbox_compute(points, S, ws.n_total, ws.bbox, ws.occ, stream)) return -3;
    hipLaunchKernelGGL(grid_setup0, dim3(sb), dim3(64), 0, stream, S, ws.bbox, ws.segs, B);
    ML3D_LAUNCH_CHECK();
    hipLaunchKernelGGL(grid_setup_fixed, dim3(sb), dim3(64), 0, stream, ws.segs, B, cell);
    ML3D_LAUNCH_CHECK();
    return grid_sort(points, S, ws, stream);
}

int grid_build_derived(const float* points, Segs S, const GridWs& ws, const GridWs& parent, hipStream_t stream) {
    int B = ws.batch;
    if (B <= 0) return 0;
    zero_async(ws.cells, sizeof(int) * (size_t)(ws.total_cells + 2), stream);
    hipLaunchKernelGGL(grid_setup_derived, dim3((B + 63) / 64), dim3(64), 0, stream, S, parent.segs, ws.segs, B);
    ML3D_LAUNCH_CHECK();
    return grid_sort(points, S, ws, stream);
}

}  // namespace ml3d
