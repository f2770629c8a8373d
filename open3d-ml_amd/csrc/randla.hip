// randla.hip — RandLA-Net inference forward for gfx950 (eval mode, BatchNorm folded).
//
// Replaces the PyTorch op chain of ml3d/torch/models/randlanet.py:241-298 (see
// include/ml3d_hip.h for the line-by-line map).  Layout: every feature map is
// point-major [batch * n_l, C] f32 so a neighbour's feature row is ONE
// contiguous burst; the (B, C, N, K) intermediates the reference materialises
// (randlanet.py:547-605, 633-639) never leave the CU: a tile of points keeps
// its 16-neighbour slab X[p][c][k] in LDS, the attention scores, the softmax
// over K and the weighted sum live in registers (one thread owns all 16
// neighbours of one (point, channel)), and only the pooled [N, C] rows go back
// to HBM.
//
// Kernels
//   linear_act      y = act(b + [a0 | a1[gather]] . WT)        (fc0, mlp1, mlp, decoder, fc1)
//   lfa_stage<1>    LocalSpatialEncoding #1 + AttentivePooling #1   -> p1 [N, d/2]
//   lfa_stage<2>    LSE #2 + AttentivePooling #2 + mlp2 + shortcut  -> enc [N, 2d]
//   gather_max      random_sample: max over the K neighbours of the kept prefix
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "gemm.h"
#include <gfx950_ops.h>
#include "grid.h"
#include "ml3d_hip.h"

namespace ml3d {

// A/B switches (speed, never results): every ML3D_* variable this file understands is read ONCE, at the first forward call
// of the process; the library keeps no other state.  (The variant tests run one process per setting.)
struct Knobs {
    bool attn_xcd, attn_split, dec_split, dec_fc1, mlp_shaped;
    bool force_valu, no_fuse;
    int linear;
    int attn_grid, attn16_grid;
    long long fuse_rows;
};
// The A/B switches of rounds 1-3 are settled (numbers in profiles/DESIGN_rounds_1_to_4.md §3.3, §9): XCD-aware tile walk, split score Linear, split
// decoder, decoder-last + fc1 chain and shaped MLP chains are ON wherever their shape conditions hold; the generic VALU kernels
// and the per-layer launches remain as the fallback for widths / sizes the MFMA kernels do not cover.  The product library
// reads nothing from the environment (no process-wide state); only the tests' HOST EMULATOR build
// (-DML3D_TEST_HOOKS) reads ML3D_RANDLA_FUSE_ROWS, once: the row count from which per-point chains fuse, lowered there so that
// small levels reach the fused kernels.
static const Knobs& knobs() {
    static const Knobs k = [] {
        Knobs v;
        v.attn_xcd = v.attn_split = v.dec_split = v.dec_fc1 = v.mlp_shaped = true;
        v.force_valu = v.no_fuse = false;
        v.linear = 0;
        v.attn_grid = 2560;
        v.attn16_grid = 4096;
        v.fuse_rows = 64 * 1024;
#ifdef ML3D_TEST_HOOKS
        if (const char* e = getenv("ML3D_RANDLA_FUSE_ROWS")) v.fuse_rows = atoll(e);
#endif
        return v;
    }();
    return k;
}

// A/B switch (build time, tools/build_variant.sh): which attention stages run on the bf16 matrix pipe (three-way split).
// bit 0: D = 128 / 256 (lfa_attn_b3), bit 1: D = 64 (lfa_attn_wave_b3); 0 keeps the f32-MFMA kernels everywhere
#ifndef ML3D_ATTN_B3
#define ML3D_ATTN_B3 3
#endif
// deep per-point Linears of the forward (mlp2 | shortcut of the 128- and 256-wide layers, the 512 -> 512 mlp) on the bf16 matrix pipe from this
// K on (0: never).  Same-box A/B (profiles/r06_linear_b3_ab.log): never 6965 frames/s, K >= 64: 7066, K >= 128: 6937 (the K = 128 GEMMs lose
// more to their pack launch than the pipe returns), K >= 256: 7104
#ifndef ML3D_LIN_B3_MINK
#define ML3D_LIN_B3_MINK 256
#endif
// per-point chains (decoder-last + fc1, pool2 + mlp2 | shortcut of the 64-wide layer) on the bf16 pipe with activations in registers
#ifndef ML3D_CHAIN_B3
#define ML3D_CHAIN_B3 1
#endif
#ifndef ML3D_B3_LB_DIV
#define ML3D_B3_LB_DIV 1        // (register-pressure probe only: 2 lifts the budget to 512 VGPRs)
#endif
#ifndef ML3D_B3_TP256_S1
#define ML3D_B3_TP256_S1 4
#endif
#ifndef ML3D_B3_TP256_S2
#define ML3D_B3_TP256_S2 2
#endif

constexpr int RK = 16;        // neighbours per point (num_neighbors in every reference config)
constexpr int XROW = 20;      // LDS row pitch of the K-slab: 16 + 4 pad floats keeps b128 reads conflict-free
constexpr int LFA_THREADS = 256;

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
// the same value in two instructions (v_mul, v_max) when 0 < slope < 1: max(v, slope * v) picks v for v > 0
__device__ __forceinline__ float lrelu_max(float v, float slope) { return fmax_raw(v, v * slope); }

// ------------------------------------------------------------------------------------------------
// generic per-point linear layer with optional [a0 | a1[gather]] concat input
// ------------------------------------------------------------------------------------------------
struct LinArgs {
    const float* a0; int c0;             // rows [m][c0]
    const float* a1; int c1;             // optional second source (gathered)
    const int32_t* gather;               // [m] item-local row of a1, or nullptr -> a1 row = m
    int64_t rows_per_item;               // n of the output level
    int64_t a1_rows_per_item;            // n of the gathered level
    const float* wt;                     // [c0 + c1][cout]
    const float* bias;                   // [cout]
    const float* bias2;                  // optional second bias added to `bias` (mlp2 + shortcut fusion)
    float* out;                          // [m][cout]
    int64_t m_total;
    int cout;
    int act;                             // 0 none, 1 leaky relu
    float slope;
    // optional scratch for the weights' three bf16 planes (gemm_pack_bf16x3_bytes(c0 + c1, cout) bytes, 16-byte aligned): a deep
    // dense Linear (K >= ML3D_LIN_B3_MINK, K % 32 == 0, no gather) then runs on the bf16 matrix pipe (gemm_tile_bf3, float32-equivalent)
    void* pack_ws; size_t pack_bytes;
};

// fc0 (+ folded bn0 + lrelu 0.2) and the first encoder layer's mlp1 (8 -> 8, lrelu 0.2) in one pass over the input rows
// (randlanet.py:266-271, 680): one thread per point, both weight sets in scalar registers (`__restrict__`: scalar loads), 12-36
// bytes in and two 32-byte rows out.  As two launches -- the generic linear_act, 4 rows x 1 column per thread, then an 8 -> 8
// chain -- this was 0.12 + 0.04 ms per 64-frame step for 0.22 GB of traffic.
__global__ void __launch_bounds__(256)
head_fc0_mlp1(const float* __restrict__ x, int c0, const float* __restrict__ w0, const float* __restrict__ b0,
              const float* __restrict__ w1, const float* __restrict__ b1, int64_t m, float* __restrict__ feat,
              float* __restrict__ f1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    float f[8], g[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = b0[c];
    const float* xr = x + i * c0;
    for (int j = 0; j < c0; ++j) {
        const float v = xr[j];
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] = fmaf(v, w0[j * 8 + c], f[c]);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) { f[c] = lrelu(f[c], 0.2f); g[c] = b1[c]; }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < 8; ++c) g[c] = fmaf(f[j], w1[j * 8 + c], g[c]);
#pragma unroll
    for (int c = 0; c < 8; ++c) g[c] = lrelu(g[c], 0.2f);
    float4* fo = reinterpret_cast<float4*>(feat + i * 8);
    float4* go = reinterpret_cast<float4*>(f1 + i * 8);
    fo[0] = make_float4(f[0], f[1], f[2], f[3]); fo[1] = make_float4(f[4], f[5], f[6], f[7]);
    go[0] = make_float4(g[0], g[1], g[2], g[3]); go[1] = make_float4(g[4], g[5], g[6], g[7]);
}

constexpr int LIN_RM = 4;  // rows per thread

__global__ void __launch_bounds__(256) linear_act(LinArgs A) {
    int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t groups = (A.m_total + LIN_RM - 1) / LIN_RM;
    if (item >= groups * A.cout) return;
    int o = (int)(item % A.cout);
    int64_t m0 = (item / A.cout) * LIN_RM;
    float acc[LIN_RM];
    const float* r0[LIN_RM];
    const float* r1[LIN_RM];
#pragma unroll
    for (int r = 0; r < LIN_RM; ++r) {
        int64_t m = m0 + r < A.m_total ? m0 + r : A.m_total - 1;
        acc[r] = (A.bias ? A.bias[o] : 0.f) + (A.bias2 ? A.bias2[o] : 0.f);
        r0[r] = A.a0 + m * A.c0;
        r1[r] = nullptr;
        if (A.a1) {
            int64_t row = m;
            if (A.gather) {
                int64_t b = m / A.rows_per_item;
                row = b * A.a1_rows_per_item + A.gather[m];
            }
            r1[r] = A.a1 + row * A.c1;
        }
    }
    for (int i = 0; i < A.c0; ++i) {
        float w = A.wt[(int64_t)i * A.cout + o];
#pragma unroll
        for (int r = 0; r < LIN_RM; ++r) acc[r] = fmaf(r0[r][i], w, acc[r]);
    }
    if (A.a1) {
        for (int i = 0; i < A.c1; ++i) {
            float w = A.wt[(int64_t)(A.c0 + i) * A.cout + o];
#pragma unroll
            for (int r = 0; r < LIN_RM; ++r) acc[r] = fmaf(r1[r][i], w, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < LIN_RM; ++r) {
        if (m0 + r < A.m_total) {
            float v = acc[r];
            if (A.act) v = lrelu(v, A.slope);
            A.out[(m0 + r) * A.cout + o] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LocalFeatureAggregation, two fused stages
// ------------------------------------------------------------------------------------------------
struct LfaArgs {
    const float* xyz;          // [batch, n0, 3] — level l is the prefix [:n]
    const int32_t* nidx;       // [batch * n, 16] item-local neighbour rows
    int64_t n, n0, m_total;    // points per item at this level, stride of xyz items, batch * n
    const float* gfeat;        // stage 1: f1 [m, h]   stage 2: p1 [m, h]
    const float* lse1_wt;      // [10][h]
    const float* lse1_b;
    const float* lse2_wt;      // [h][h]        (stage 2)
    const float* lse2_b;
    const float* score_wt;     // [d][d]
    const float* score_b;
    const float* pool_wt;      // stage 1: [d][h]; stage 2: [d][d]
    const float* pool_b;
    const float* mlp2_wt;      // [d][2d]       (stage 2)
    const float* mlp2_b;
    const float* short_wt;     // [d_in][2d]    (stage 2)
    const float* short_b;
    const float* feat_in;      // [m, d_in]     (stage 2: LFA input, for the shortcut)
    int d_in;
    float* out;                // stage 1: p1 [m, h]; stage 2: enc [m, 2d]
    int64_t xcd_chunk;         // > 0: XCD-aware tile walk, tiles per chunk (see xcd_tile)
    const float* gscore;       // optional [m, d]: gfeat . score_WT[0:h, :] per POINT (see lfa_attn_pf<.., SPLIT>)
    // optional [m]: a permutation of the point rows (cloud-major, spatially sorted inside a cloud -- the cell order of
    // the neighbour pyramid's grid).  Tile t then works on points order[t * TP .. ] instead of rows t * TP ..: the
    // results are identical, but consecutive tiles share neighbour rows in L1/L2 (ORD template variants).
    const int32_t* order;
};

// XCD-aware tile order.  Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md, observed), and each XCD has its
// own 4 MiB L2.  The neighbour gathers of a tile touch random rows of the tile's own cloud (RandLA points are
// shuffled), so the tiles are cut into chunks of ~one cloud (or a fraction of one) and chunk c is walked only
// by the workgroups of XCD c % 8: an XCD's L2 then holds one cloud's feature rows instead of all of them.
// i = index among this XCD's tiles; returns the tile, -1 when the XCD is done, or >= tiles for a hole.
__device__ __forceinline__ int64_t xcd_tile(int64_t i, int xcd, int64_t chunk, int64_t tiles) {
    const int64_t ci = i / chunk;
    const int64_t c = ci * 8 + xcd;
    if (c * chunk >= tiles) return -1;
    return c * chunk + (i - ci * chunk);
}

static int64_t xcd_chunk_tiles(int64_t tiles, int64_t batch) {
    if (batch <= 0 || tiles < 64) return 0;
    int64_t nch = batch;
    while (nch < 32) nch *= 2;                  // >= 4 chunks per XCD keeps the XCDs balanced
    int64_t chunk = (tiles + nch - 1) / nch;
    return chunk > 0 ? chunk : 0;
}

template <int D>
struct LfaCfg {
    static constexpr int H = D / 2;
    static constexpr int TP = (LFA_THREADS / D) > 0 ? (LFA_THREADS / D) : 1;   // points per tile
    static constexpr int CPT = D > LFA_THREADS ? D / LFA_THREADS : 1;         // channels per thread
    static constexpr int CT = D > LFA_THREADS ? LFA_THREADS : D;              // threads across channels
    static constexpr int XSLAB = D * XROW + 4;                                // floats per point slab (+4: skew)
};

// attention over the 16 neighbours of (point p, channel c): scores = score_b + X . score_wt[:, c];
// softmax over k; returns sum_k softmax_k * X[p][c][k]          (randlanet.py:633-637)
template <int D>
__device__ __forceinline__ float attentive_pool(const float* Xp, int c, const float* __restrict__ score_wt,
                                                float bias) {
    float acc[RK];
#pragma unroll
    for (int k = 0; k < RK; ++k) acc[k] = bias;
    for (int j = 0; j < D; ++j) {
        float w = score_wt[j * D + c];
        const float4* xr = reinterpret_cast<const float4*>(Xp + j * XROW);
#pragma unroll
        for (int q = 0; q < RK / 4; ++q) {
            float4 x = xr[q];
            acc[4 * q + 0] = fmaf(x.x, w, acc[4 * q + 0]);
            acc[4 * q + 1] = fmaf(x.y, w, acc[4 * q + 1]);
            acc[4 * q + 2] = fmaf(x.z, w, acc[4 * q + 2]);
            acc[4 * q + 3] = fmaf(x.w, w, acc[4 * q + 3]);
        }
    }
    float mx = acc[0];
#pragma unroll
    for (int k = 1; k < RK; ++k) mx = fmaxf(mx, acc[k]);
    float sum = 0.f, agg = 0.f;
    const float* xc = Xp + c * XROW;
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        float e = expf(acc[k] - mx);
        sum += e;
        agg = fmaf(e, xc[k], agg);
    }
    return agg / sum;
}

template <int D, int STAGE>
__global__ void __launch_bounds__(LFA_THREADS) lfa_stage(LfaArgs A) {
    using C = LfaCfg<D>;
    constexpr int H = C::H, TP = C::TP, CPT = C::CPT, CT = C::CT;
    HIP_DYNAMIC_SHARED(float, smem)
    // carve (all offsets multiples of 4 floats = 16 B)
    float* X = smem;                                 // [TP][XSLAB]   stage 1: final slab; stage 2: r1 slab
    float* X2 = X + TP * C::XSLAB;                   // [TP][XSLAB]   stage 2 only
    float* REL = X2 + (STAGE == 2 ? TP * C::XSLAB : 0);   // [TP][16][12]
    float* AGG = REL + TP * RK * 12;                 // [TP][D]
    float* P2 = AGG + TP * D;                        // [TP][D]       stage 2
    float* FIN = P2 + (STAGE == 2 ? TP * D : 0);     // [TP][d_in]    stage 2
    int* NROW = reinterpret_cast<int*>(FIN + (STAGE == 2 ? TP * ((A.d_in + 3) & ~3) : 0));  // [TP][16]

    const int tid = threadIdx.x;
    const int64_t tiles = (A.m_total + TP - 1) / TP;
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t m_base = tile * TP;
        // ---- phase 1: relative position encoding inputs (randlanet.py:579-594) ----------------
        for (int e = tid; e < TP * RK; e += LFA_THREADS) {
            int p = e / RK, k = e % RK;
            int64_t m = m_base + p;
            if (m < A.m_total) {
                int64_t b = m / A.n, nl = m - b * A.n;
                int nb = A.nidx[m * RK + k];
                const float* q = A.xyz + 3 * (b * A.n0 + nl);
                const float* s = A.xyz + 3 * (b * A.n0 + nb);
                float qx = q[0], qy = q[1], qz = q[2], sx = s[0], sy = s[1], sz = s[2];
                float dx = qx - sx, dy = qy - sy, dz = qz - sz;
                float dist = sqrtf(dx * dx + dy * dy + dz * dz);
                float* r = REL + (p * RK + k) * 12;
                r[0] = dist; r[1] = dx; r[2] = dy; r[3] = dz; r[4] = qx; r[5] = qy; r[6] = qz;
                r[7] = sx; r[8] = sy; r[9] = sz;
                NROW[p * RK + k] = (int)(b * A.n + nb);
            } else {
                float* r = REL + (p * RK + k) * 12;
                for (int j = 0; j < 10; ++j) r[j] = 0.f;
                NROW[p * RK + k] = 0;
            }
        }
        if (STAGE == 2) {
            for (int e = tid; e < TP * A.d_in; e += LFA_THREADS) {
                int p = e / A.d_in, i = e - p * A.d_in;
                int64_t m = m_base + p;
                FIN[p * ((A.d_in + 3) & ~3) + i] = m < A.m_total ? A.feat_in[m * A.d_in + i] : 0.f;
            }
        }
        __syncthreads();
        // ---- phase 2: neighbour slab X[p][c][k]: c < H gathered features, c >= H encoded rel ----
        float* XG = STAGE == 1 ? X : X2;   // where the gathered half goes
        for (int e = tid; e < TP * RK * D; e += LFA_THREADS) {
            int c = e % D, k = (e / D) % RK, p = e / (D * RK);
            float v;
            if (c < H) {
                v = A.gfeat[(int64_t)NROW[p * RK + k] * H + c];
                XG[p * C::XSLAB + c * XROW + k] = v;
            } else {
                int cc = c - H;
                const float* r = REL + (p * RK + k) * 12;
                v = A.lse1_b[cc];
#pragma unroll
                for (int j = 0; j < 10; ++j) v = fmaf(r[j], A.lse1_wt[j * H + cc], v);
                v = lrelu(v, 0.2f);
                X[p * C::XSLAB + c * XROW + k] = v;
            }
        }
        __syncthreads();
        if (STAGE == 2) {
            // ---- phase 2b: r2 = lrelu(lse2(r1)) for all 16 neighbours of (p, cc) --------------
            for (int e = tid; e < TP * H; e += LFA_THREADS) {
                int cc = e % H, p = e / H;
                float acc[RK];
                float bb = A.lse2_b[cc];
#pragma unroll
                for (int k = 0; k < RK; ++k) acc[k] = bb;
                const float* Xp = X + p * C::XSLAB + H * XROW;
                for (int j = 0; j < H; ++j) {
                    float w = A.lse2_wt[j * H + cc];
                    const float4* xr = reinterpret_cast<const float4*>(Xp + j * XROW);
#pragma unroll
                    for (int q = 0; q < RK / 4; ++q) {
                        float4 x = xr[q];
                        acc[4 * q + 0] = fmaf(x.x, w, acc[4 * q + 0]);
                        acc[4 * q + 1] = fmaf(x.y, w, acc[4 * q + 1]);
                        acc[4 * q + 2] = fmaf(x.z, w, acc[4 * q + 2]);
                        acc[4 * q + 3] = fmaf(x.w, w, acc[4 * q + 3]);
                    }
                }
                float* dst = X2 + p * C::XSLAB + (H + cc) * XROW;
#pragma unroll
                for (int k = 0; k < RK; ++k) dst[k] = lrelu(acc[k], 0.2f);
            }
            __syncthreads();
        }
        // ---- phase 3/4: attention scores, softmax over K, weighted sum ------------------------
        {
            const float* XS = STAGE == 1 ? X : X2;
            int p = tid / CT, c0 = tid % CT;
            if (p < TP) {
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    int c = c0 + q * CT;
                    AGG[p * D + c] = attentive_pool<D>(XS + p * C::XSLAB, c, A.score_wt, A.score_b[c]);
                }
            }
        }
        __syncthreads();
        // ---- phase 5: pooling MLP (SharedMLP d -> d_out, lrelu 0.2) -----------------------------
        constexpr int PO = STAGE == 1 ? H : D;
        for (int e = tid; e < TP * PO; e += LFA_THREADS) {
            int o = e % PO, p = e / PO;
            float v = A.pool_b[o];
            const float* a = AGG + p * D;
            for (int c = 0; c < D; ++c) v = fmaf(a[c], A.pool_wt[c * PO + o], v);
            v = lrelu(v, 0.2f);
            if (STAGE == 1) {
                int64_t m = m_base + p;
                if (m < A.m_total) A.out[m * H + o] = v;
            } else {
                P2[p * D + o] = v;
            }
        }
        if (STAGE == 2) {
            __syncthreads();
            // ---- phase 6: lrelu_0.01(mlp2(p2) + shortcut(feat))  (randlanet.py:692) ----------
            const int dpad = (A.d_in + 3) & ~3;
            for (int e = tid; e < TP * 2 * D; e += LFA_THREADS) {
                int o = e % (2 * D), p = e / (2 * D);
                float v = A.mlp2_b[o];
                const float* a = P2 + p * D;
                for (int c = 0; c < D; ++c) v = fmaf(a[c], A.mlp2_wt[c * 2 * D + o], v);
                float s = A.short_b[o];
                const float* f = FIN + p * dpad;
                for (int i = 0; i < A.d_in; ++i) s = fmaf(f[i], A.short_wt[i * 2 * D + o], s);
                v = lrelu(v + s, 0.01f);
                int64_t m = m_base + p;
                if (m < A.m_total) A.out[m * 2 * D + o] = v;
            }
        }
        __syncthreads();
    }
}


// ------------------------------------------------------------------------------------------------
// MFMA path (D in {32, 64, 128, 256}): the per-neighbour GEMMs on v_mfma_f32_32x32x2_f32
// (exact f32: bitwise an fmaf chain, 157 TFLOP/s peak — MI355X_MICROARCH.md).
//
// Rows of the GEMM are (point, neighbour) pairs: a 32-row MFMA tile = 2 points x 16 neighbours.
//   stage 2 only:  R2[rows, H] = lrelu(R1[rows, H] . lse2_WT + b)          (LocalSpatialEncoding #2)
//   both stages :  S[rows, D]  = X[rows, D] . score_WT + b                  (AttentivePooling Linear)
// X = [gathered neighbour features | encoded relative positions] lives in LDS as the A operand
// (row pitch D+4 floats: conflict-free ds_read_b128, four K-steps per read).  Each wave owns one
// 32-column tile of the weight matrix IN REGISTERS for the whole kernel (D/2 VGPRs) and walks the
// row tiles.  In the 32x32 C layout a lane holds one column and 8 of the 16 neighbours of each of
// the two points, so the softmax over K is 8 in-lane values + ONE lane^32 exchange; the weighted
// sum reads X back from LDS and only agg[point, D] leaves the CU.
// K is split between the wave halves: lanes 0-31 feed k in [0, K/2), lanes 32-63 k in [K/2, K).
// ------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

static int device_cu_count() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus < 1)
        cus = 256;
    return cus;
}

#define SYNC_ATTN() block_sync_lds()   // LDS-only workgroup barrier (grid.h): global loads / stores stay in flight

template <int D>
struct MfmaCfg {
    static constexpr int H = D / 2;
    static constexpr int NT = D / 32;                        // score column tiles
    static constexpr int NT2 = (H + 31) / 32;                // lse2 column tiles
    static constexpr int WAVES = NT > 4 ? NT : 4;
    static constexpr int THREADS = WAVES * 64;
    static constexpr int TP = D <= 64 ? 8 : (D == 128 ? 4 : 2);   // points per tile (half-size tiles measured slower)
    static constexpr int ROWS = TP * RK;
    static constexpr int RT = ROWS / 32;                     // 32-row MFMA tiles
    static constexpr int RG = WAVES / NT;                    // waves sharing a column tile take different row tiles
    static constexpr int RG2 = WAVES / NT2;
    static constexpr int XP = D + 4;                         // LDS row pitch of X
    static constexpr int RP = H + 4;                         // LDS row pitch of R1
    // H <= 32: one wave owns all H columns of its row tile, so lse2 can run IN PLACE on X[:, H:] (its A operand is
    // in registers before its result is stored) and the separate R1 tile (ROWS x RP floats of LDS) disappears:
    // 60 -> 41 KB per workgroup at D = 64, i.e. three resident workgroups per CU instead of two
    static constexpr bool INPLACE = NT2 == 1;
};

__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// acc += A[rows of tile, K] . B[K, 32 cols]; A from LDS (pitch AP), B from registers
template <int KD, int AP>
__device__ __forceinline__ f32x16 mfma_rows(const float* a_row /* &A[row][hi*KD/2] */, const float (&b)[KD / 2],
                                            f32x16 acc) {
#pragma unroll
    for (int s4 = 0; s4 < KD / 8; ++s4) {
        float4 a = *reinterpret_cast<const float4*>(a_row + 4 * s4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[4 * s4 + 0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[4 * s4 + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[4 * s4 + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[4 * s4 + 3], acc, 0, 0, 0);
    }
    return acc;
}


// softmax over one point's 16 neighbours in the 32x32 MFMA C layout (8 scores in this lane, the other 8 in lane ^ 32)
// and the weighted sum of the lane's feature column, on PACKED f32 pairs (accumulator registers 2j, 2j+1 and the
// adjacent X rows they belong to): v_pk_fma for the exp2 arguments and the weighted sum, v_pk_add for the
// denominators -- 20 instead of 32 VALU instructions per 8 scores.  exp(s - max) = exp2(s * log2e - max * log2e).
typedef float v2f __attribute__((ext_vector_type(2)));
template <int XP, class ACC>
__device__ __forceinline__ void softmax_wsum8(const ACC& sc, int o /* 0 or 8 */, const float* __restrict__ xrow0,
                                              float& num, float& den) {
    constexpr float LOG2E = 1.4426950408889634f;
    float mx = sc[o];
#pragma unroll
    for (int r = 1; r < 8; ++r) mx = fmax_raw(mx, sc[o + r]);
    mx = fmax_raw(mx, __shfl_xor(mx, 32));
    const float nmx = -mx * LOG2E;
    const v2f nm = {nmx, nmx}, l2 = {LOG2E, LOG2E};
    v2f sum2 = {0.f, 0.f}, ag2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r0 = 2 * j;                                        // local rows r0, r0 + 1 -> tile rows (r & 3) + 8 * (r >> 2)
        const v2f a = __builtin_elementwise_fma((v2f){sc[o + r0], sc[o + r0 + 1]}, l2, nm);
        const v2f e = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
        const float* xr = xrow0 + ((r0 & 3) + 8 * (r0 >> 2)) * XP;
        const v2f x = {xr[0], xr[XP]};
        sum2 += e;
        ag2 = __builtin_elementwise_fma(e, x, ag2);
    }
    const float sum = sum2.x + sum2.y, ag = ag2.x + ag2.y;
    num = ag + __shfl_xor(ag, 32);
    den = sum + __shfl_xor(sum, 32);
}

// ------------------------------------------------------------------------------------------------
// lfa_attn_pf — the same attention stage with the NEXT tile's global traffic in flight under the
// current tile's MFMA phases.  A tile needs three dependent global hops (neighbour index -> neighbour
// xyz / neighbour feature row); in lfa_attn_mfma every workgroup sits through them at the top of each
// tile (~3.5 us of a ~12 us tile).  Here each thread owns, across tiles, the same slots of the tile's
// inputs: G float4 pieces of the gathered feature rows and (threads < ROWS) one (point, neighbour) row of
// relative positions.  Schedule per tile t:
//     registers -> LDS: X[:, 0:H] <- feature pieces, REL <- relative positions          (loads of t landed)
//     request the neighbour indices of t+1                                             (hop 1)
//     barrier; lse1 [; lse2]                                                           (MFMA)
//     request xyz + feature pieces of t+1 through those indices                        (hops 2, 3)
//     barrier; scores, softmax, weighted sum (the long MFMA phase); barrier
// The barriers are block_sync_lds (LDS only), so the requests stay in flight across them.  lse1/lse2
// weights move from registers to LDS (B operand by ds_read) to make room for the in-flight registers,
// and with lse2 in place (H <= 32) the lse1 -> lse2 hand-off is inside one wave: 3 barriers per tile.
// ------------------------------------------------------------------------------------------------
// SPLIT: the score Linear acts on X = [gathered neighbour features | encoded positions], and its first half is linear
// in a PER-POINT quantity: W . [f[nb] ; r] = (W_top . f)[nb] + W_bot . r.  With A.gscore = f . W_top^T precomputed
// per point by one small GEMM (1/16 of the work it replaces), the kernel gathers the neighbour's gscore row straight
// into the accumulator layout (32 lanes read 128 contiguous bytes of a row) and runs the MFMAs over K = H instead
// of D: half the score MFMAs and half the resident weight registers.  f[nb] is still gathered into X for the
// weighted sum.  Used for D >= 128, where the extra D floats per neighbour are small next to the MFMA time saved.
template <int D, int STAGE, bool SPLIT, bool ORD>
__global__ void __launch_bounds__((MfmaCfg<D>::THREADS), (D <= 64 ? 3 : (MfmaCfg<D>::THREADS / 256))) lfa_attn_pf(LfaArgs A) {
    using C = MfmaCfg<D>;
    constexpr int H = C::H, ROWS = C::ROWS, XP = C::XP, RP = C::RP, THREADS = C::THREADS;
    constexpr int HP = C::NT2 * 32;                               // lse weight pitch (column tiles, zero padded)
    constexpr int Q = H / 4;                                      // float4 pieces per gathered row
    constexpr int G = ROWS * Q / THREADS;                         // pieces per thread
    static_assert(ROWS * Q % THREADS == 0 && ROWS <= THREADS, "tile shape");
    HIP_DYNAMIC_SHARED(float, smem)
    float* X = smem;                                              // [ROWS][XP]
    float* R1 = X + ROWS * XP;                                    // [ROWS][RP]   (stage 2, unless in place)
    float* REL = R1 + ((STAGE == 2 && !C::INPLACE) ? ROWS * RP : 0);   // [ROWS][12]
    float* W1 = REL + ROWS * 12;                                  // [12][HP]
    float* W2 = W1 + 12 * HP;                                     // [H][HP]      (stage 2)
    uint32_t* NROWG = reinterpret_cast<uint32_t*>(W2 + (STAGE == 2 ? H * HP : 0));   // [ROWS] global neighbour row (SPLIT)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, col = lane & 31;

    const int ct = wave % C::NT, rg = wave / C::NT;
    constexpr int KS = SPLIT ? H : D, K0 = SPLIT ? H : 0;         // score MFMAs run over input channels [K0, K0 + KS)
    float bs[KS / 2];
#pragma unroll
    for (int s = 0; s < KS / 2; ++s) bs[s] = A.score_wt[(K0 + hi * (KS / 2) + s) * D + ct * 32 + col];
    const float sbias = A.score_b[ct * 32 + col];
    const int ct2 = wave % C::NT2, rg2 = wave / C::NT2;
    const int col2 = ct2 * 32 + col;
    for (int e = tid; e < 12 * HP; e += THREADS) {
        const int k = e / HP, c = e - k * HP;
        W1[e] = (k < 10 && c < H) ? A.lse1_wt[k * H + c] : 0.f;
    }
    float l2bias = 0.f;
    if constexpr (STAGE == 2) {
        for (int e = tid; e < H * HP; e += THREADS) {
            const int k = e / HP, c = e - k * HP;
            W2[e] = c < H ? A.lse2_wt[k * H + c] : 0.f;
        }
        l2bias = col2 < H ? A.lse2_b[col2] : 0.f;
    }
    const float b1 = col2 < H ? A.lse1_b[col2] : 0.f;

    const int64_t tiles = (A.m_total + C::TP - 1) / C::TP;
    const bool xw = A.xcd_chunk > 0;
    const int64_t w_step = xw ? (int64_t)(gridDim.x >> 3) : (int64_t)gridDim.x;
    int64_t wi = xw ? (int64_t)(blockIdx.x >> 3) : (int64_t)blockIdx.x;
    auto next_tile = [&]() -> int64_t {
        for (;;) {
            int64_t t = wi;
            if (xw) t = xcd_tile(wi, (int)(blockIdx.x & 7), A.xcd_chunk, tiles);
            wi += w_step;
            if (t < 0) return -1;
            if (t < tiles) return t;
            if (!xw) return -1;
        }
    };

    // ---- this thread's slots of a tile's inputs ---------------------------------------------------
    int gi[G], nb_mine = 0;              // hop 1: neighbour rows (item-local)
    float4 gq[G];                        // hop 3: gathered feature pieces
    float qx = 0.f, qy = 0.f, qz = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;   // hop 2
    bool mine_valid = false;
    uint32_t grow_mine = 0;              // global row of this thread's neighbour (SPLIT)
    // (per-tile bookkeeping is wave-uniform and 32-bit: f32 MFMA and VALU share the SIMD's issue cycles on gfx950, so a
    //  64-bit division per lane per tile costs as much as two MFMAs; the launcher guarantees m_total, n0 < 2^30, n >= TP)
    const uint32_t n_pts = (uint32_t)A.n, m_tot = (uint32_t)A.m_total;
    uint32_t gmp[ORD ? G : 1], mp_mine = 0, m_first = 0;     // ORD: point rows of this thread's slots / of the tile's first point
    auto request_idx = [&](uint32_t tile) {
        const uint32_t have = m_tot - tile * C::TP;                 // points of the tile that exist
        const uint32_t lim = (have < (uint32_t)C::TP ? have : (uint32_t)C::TP) * RK;
        if constexpr (ORD) {
            const int32_t* ord = A.order + tile * C::TP;
            m_first = (uint32_t)ord[0];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const uint32_t row = (uint32_t)(tid + i * THREADS) / Q;
                gi[i] = -1;
                if (row < lim) {
                    gmp[i] = (uint32_t)ord[row / RK];
                    gi[i] = A.nidx[(int64_t)gmp[i] * RK + (row & (RK - 1))];
                }
            }
            if (tid < ROWS) {
                nb_mine = -1;
                if ((uint32_t)tid < lim) {
                    mp_mine = (uint32_t)ord[tid / RK];
                    nb_mine = A.nidx[(int64_t)mp_mine * RK + (tid & (RK - 1))];
                }
            }
        } else {
            const int32_t* nb = A.nidx + (int64_t)tile * (C::TP * RK);  // the tile's (point, neighbour) rows are contiguous
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const uint32_t row = (uint32_t)(tid + i * THREADS) / Q;
                gi[i] = row < lim ? nb[row] : -1;
            }
            if (tid < ROWS) nb_mine = (uint32_t)tid < lim ? nb[tid] : -1;
        }
    };
    auto request_data = [&](uint32_t tile) {
        // scalar: the cloud of the tile's first point; the tile's other points are in it or in the next one (ORD: the
        // order is cloud-major, so the same holds for order[t * TP ..])
        const uint32_t m0 = ORD ? (uint32_t)__builtin_amdgcn_readfirstlane((int)m_first) : tile * C::TP;
        const uint32_t b0 = m0 / n_pts, l0 = m0 - b0 * n_pts;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int e = tid + i * THREADS;
            const int row = e / Q, q = e - row * Q;
            gq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gi[i] >= 0) {
                uint32_t b;
                if constexpr (ORD) b = b0 + (gmp[i] >= (b0 + 1) * n_pts ? 1u : 0u);
                else b = b0 + ((l0 + (uint32_t)(row / RK) >= n_pts) ? 1u : 0u);
                gq[i] = *reinterpret_cast<const float4*>(A.gfeat + ((int64_t)b * n_pts + (uint32_t)gi[i]) * H + 4 * q);
            }
        }
        if (tid < ROWS) {
            mine_valid = nb_mine >= 0;
            if (mine_valid) {
                uint32_t b, l;
                if constexpr (ORD) {
                    b = b0 + (mp_mine >= (b0 + 1) * n_pts ? 1u : 0u);
                    l = mp_mine - b * n_pts;
                } else {
                    const uint32_t lp = l0 + (uint32_t)(tid / RK);
                    const bool wrap = lp >= n_pts;
                    b = b0 + (wrap ? 1u : 0u);
                    l = wrap ? lp - n_pts : lp;
                }
                grow_mine = b * n_pts + (uint32_t)nb_mine;
                const float* xb = A.xyz + 3 * ((int64_t)b * A.n0);
                const float* qp = xb + 3 * l;
                const float* sp = xb + 3 * (uint32_t)nb_mine;
                qx = qp[0]; qy = qp[1]; qz = qp[2]; sx = sp[0]; sy = sp[1]; sz = sp[2];
            }
        }
    };

    int64_t cur = next_tile();
    if (cur >= 0) { request_idx((uint32_t)cur); request_data((uint32_t)cur); }
    block_sync_lds();                                            // W1 / W2 staged
    while (cur >= 0) {
        const int64_t nxt = next_tile();
        const int64_t m_base = cur * C::TP;
        // ---- registers -> LDS -----------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int e = tid + i * THREADS;
            const int row = e / Q, q = e - row * Q;
            *reinterpret_cast<float4*>(X + row * XP + 4 * q) = gq[i];
        }
        if (tid < ROWS) {
            float* r = REL + tid * 12;
            if (mine_valid) {
                const float dx = qx - sx, dy = qy - sy, dz = qz - sz;
                r[0] = sqrtf(dx * dx + dy * dy + dz * dz);
                r[1] = dx; r[2] = dy; r[3] = dz; r[4] = qx; r[5] = qy; r[6] = qz; r[7] = sx; r[8] = sy; r[9] = sz;
            } else {
#pragma unroll
                for (int j = 0; j < 10; ++j) r[j] = 0.f;
            }
            r[10] = 0.f; r[11] = 0.f;                              // K padding of the lse1 MFMA
            // byte offset of the neighbour's gscore row (the launcher keeps m * D * 4 below 2^32); rows past the end of
            // the data read row 0 -- their scores feed outputs that are never stored
            if constexpr (SPLIT) NROWG[tid] = mine_valid ? grow_mine * (uint32_t)(D * 4) : 0u;
        }
        if (nxt >= 0) request_idx((uint32_t)nxt);
        block_sync_lds();
        // ---- r1 = lrelu(lse1(rel)) on MFMA (K = 12) -> X[:, H:] (stage 1 / in place) or R1 -----------------
        for (int rt = rg2; rt < C::RT; rt += C::RG2) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = b1;
            const float* ar = REL + (rt * 32 + col) * 12 + hi * 6;
            const float* wb = W1 + hi * 6 * HP + col2;
            const float2 a01 = *reinterpret_cast<const float2*>(ar);
            const float2 a23 = *reinterpret_cast<const float2*>(ar + 2);
            const float2 a45 = *reinterpret_cast<const float2*>(ar + 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a01.x, wb[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a01.y, wb[HP], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a23.x, wb[2 * HP], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a23.y, wb[3 * HP], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a45.x, wb[4 * HP], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a45.y, wb[5 * HP], acc, 0, 0, 0);
            if (col2 < H) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * 32 + mfma_row(r, hi);
                    const float v = lrelu_max(acc[r], 0.2f);
                    if (STAGE == 1 || C::INPLACE) X[row * XP + H + col2] = v; else R1[row * RP + col2] = v;
                }
            }
        }
        if constexpr (STAGE == 2) {
            // in place the row tile's r1 was written by this very wave; otherwise other waves' columns are needed
            if constexpr (C::INPLACE) wave_lds_sync(); else block_sync_lds();
            // ---- r2 = lrelu(lse2(r1)) on MFMA -> X[:, H:] ------------------------------------------------
            for (int rt = rg2; rt < C::RT; rt += C::RG2) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = l2bias;
                const float* ar = C::INPLACE ? X + (rt * 32 + col) * XP + H + hi * (H / 2)
                                             : R1 + (rt * 32 + col) * RP + hi * (H / 2);
                const float* wb = W2 + hi * (H / 2) * HP + col2;
#pragma unroll
                for (int s4 = 0; s4 < H / 8; ++s4) {
                    const float4 a = *reinterpret_cast<const float4*>(ar + 4 * s4);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wb[(4 * s4 + 0) * HP], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wb[(4 * s4 + 1) * HP], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wb[(4 * s4 + 2) * HP], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wb[(4 * s4 + 3) * HP], acc, 0, 0, 0);
                }
                if (col2 < H) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        X[(rt * 32 + mfma_row(r, hi)) * XP + H + col2] = lrelu_max(acc[r], 0.2f);
                }
            }
        }
        if (nxt >= 0) request_data((uint32_t)nxt);
        block_sync_lds();
        // ---- scores on MFMA, softmax over the 16 neighbours, weighted sum ------------------------------
        for (int rt = rg; rt < C::RT; rt += C::RG) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = sbias;
            float gs[SPLIT ? 16 : 1];
            if constexpr (SPLIT) {
                // the neighbours' per-point score halves, requested before the MFMA chain and added after it
                const char* gbase = reinterpret_cast<const char*>(A.gscore + ct * 32);    // wave-uniform base
                const uint32_t col4 = 4u * col;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const uint4 nr = *reinterpret_cast<const uint4*>(NROWG + rt * 32 + 8 * q4 + 4 * hi);   // rows mfma_row(4 q4 .. 4 q4 + 3)
                    gs[4 * q4 + 0] = *reinterpret_cast<const float*>(gbase + (nr.x + col4));
                    gs[4 * q4 + 1] = *reinterpret_cast<const float*>(gbase + (nr.y + col4));
                    gs[4 * q4 + 2] = *reinterpret_cast<const float*>(gbase + (nr.z + col4));
                    gs[4 * q4 + 3] = *reinterpret_cast<const float*>(gbase + (nr.w + col4));
                }
            }
            acc = mfma_rows<KS, XP>(X + (rt * 32 + col) * XP + K0 + hi * (KS / 2), bs, acc);
            if constexpr (SPLIT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += gs[r];
            }
            const float* xc = X + (rt * 32) * XP + ct * 32 + col;
            float num[2], den[2];
            // point pt of the row tile: rows 16 pt + 4 hi + {0..3, 8..11}
            softmax_wsum8<XP>(acc, 0, xc + (4 * hi) * XP, num[0], den[0]);
            softmax_wsum8<XP>(acc, 8, xc + (16 + 4 * hi) * XP, num[1], den[1]);
            const float agg_mine = (hi ? num[1] : num[0]) / (hi ? den[1] : den[0]);
            int64_t m = m_base + 2 * rt + hi;                    // half 0 stores point 0, half 1 point 1
            if (m < A.m_total) {
                if constexpr (ORD) m = A.order[m];               // the point's row in the cloud-major spatial order
                A.out[m * D + ct * 32 + col] = agg_mine;
            }
        }
        block_sync_lds();
        cur = nxt;
    }
}

template <int D, int STAGE>
static size_t pf_smem_bytes() {
    using C = MfmaCfg<D>;
    constexpr int HP = C::NT2 * 32;
    return ((size_t)C::ROWS * C::XP + ((STAGE == 2 && !C::INPLACE) ? (size_t)C::ROWS * C::RP : 0) + (size_t)C::ROWS * 12 +
            12 * HP + (STAGE == 2 ? (size_t)C::H * HP : 0) + (size_t)C::ROWS) * 4;
}


// ------------------------------------------------------------------------------------------------
// lfa_attn_wave (D <= 64) — the attention stage with NO workgroup barrier in the tile loop.
// In lfa_attn_mfma / lfa_attn_pf a workgroup's waves split the column tiles of one 128-row tile, so
// they meet at 3-4 barriers per tile; with one wave of each resident workgroup per SIMD, any wave that
// queues behind another workgroup's MFMAs stalls its three siblings on the other SIMDs (measured:
// removing the score MFMAs, 74 % of the MFMA work, removes more time than they take at peak, i.e. the
// phases do not overlap).  Here ALL weights sit in LDS once per workgroup (score D x D, lse2, lse1: 21.5 KB
// at D = 64) as the MFMA B operand, and every WAVE owns whole 32-row tiles (2 points x 16 neighbours)
// end to end in its private LDS patch: gather + relative positions (requested one tile ahead, as in
// lfa_attn_pf), lse1, lse2 in place, scores for all column tiles, softmax, weighted sum.  Waves only
// synchronise with themselves (wave_lds_sync), drift freely, and with no weights in registers 12 waves
// fit per CU (LDS-bound: 10.2 KB patch each).
// ------------------------------------------------------------------------------------------------
template <int D>
struct WaveAttnCfg {
    static constexpr int H = D / 2;
    static constexpr int NT = D / 32;
    static constexpr int XP = D + 4;
    static constexpr int Q = H / 4;                  // float4 pieces per gathered row
    static constexpr int G = 32 * Q / 64;            // pieces per lane
    static constexpr int W = D == 64 ? 12 : 16;      // waves per workgroup
    static constexpr int PATCH = 32 * XP + 32 * 12 + 32;  // floats per wave: X, relative positions, gscore row offsets
    static constexpr int WFLOATS = D * D + H * 32 + 12 * 32;
};

// SPLIT (see lfa_attn_pf): A.gscore = f . W_top^T + score bias per POINT; the neighbours' rows are gathered straight
// into the score accumulators (requested before the lse phases, so they land under those MFMAs) and the score MFMAs
// run over the position half of X only (K = H): 32 instead of 64 MFMAs per tile at D = 64.
template <int D, int STAGE, bool SPLIT, bool ORD>
__global__ void __launch_bounds__((WaveAttnCfg<D>::W * 64)) lfa_attn_wave(LfaArgs A) {
    using C = WaveAttnCfg<D>;
    constexpr int H = C::H, XP = C::XP, Q = C::Q, G = C::G, NT = C::NT;
    static_assert(H <= 32 && (32 * Q) % 64 == 0, "lfa_attn_wave: D in {32, 64}");
    HIP_DYNAMIC_SHARED(float, smem)
    float* WS = smem;                                 // [D][D]   score weights (transposed Linear: [in][out])
    float* W2 = WS + D * D;                           // [H][32]  lse2 (zero padded columns)
    float* W1 = W2 + H * 32;                          // [12][32] lse1 (K padded 10 -> 12)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, col = lane & 31;
    float* X = W1 + 12 * 32 + wave * C::PATCH;        // [32][XP]  this wave's tile
    float* REL = X + 32 * XP;                         // [32][12]
    uint32_t* GOFF = reinterpret_cast<uint32_t*>(REL + 32 * 12);   // [32] byte offset of each row's gscore row (SPLIT)
    constexpr int KS = SPLIT ? H : D, K0 = SPLIT ? H : 0;          // score MFMAs run over input channels [K0, K0 + KS)

    for (int e = tid; e < KS * D; e += C::W * 64) WS[e] = A.score_wt[K0 * D + e];
    for (int e = tid; e < H * 32; e += C::W * 64) {
        const int k = e >> 5, c = e & 31;
        W2[e] = (STAGE == 2 && c < H) ? A.lse2_wt[k * H + c] : 0.f;
    }
    for (int e = tid; e < 12 * 32; e += C::W * 64) {
        const int k = e >> 5, c = e & 31;
        // K slot 10 carries the bias (REL[:, 10] = 1), slot 11 is zero
        W1[e] = c < H ? (k < 10 ? A.lse1_wt[k * H + c] : (k == 10 ? A.lse1_b[c] : 0.f)) : 0.f;
    }
    float sbias[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sbias[t] = SPLIT ? 0.f : A.score_b[t * 32 + col];    // (SPLIT: the bias is inside gscore)
    const float l2bias = (STAGE == 2 && col < H) ? A.lse2_b[col] : 0.f;
    __syncthreads();                                  // the only workgroup barrier

    // Everything about a tile except the lane's own row is wave-uniform and kept in SGPRs (the wave index is made
    // scalar explicitly): the f32 MFMAs and the VALU share the SIMD's issue cycles on gfx950 (tools/micro), so
    // every vector instruction outside the MFMAs is paid in full.  Point indices fit 32 bits (launcher checks).
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t tiles = (uint32_t)((A.m_total + 1) / 2);       // 2 points per wave tile
    const uint32_t n_pts = (uint32_t)A.n, m_tot = (uint32_t)A.m_total;
    const bool xw = A.xcd_chunk > 0;
    const uint32_t w_step = (xw ? (gridDim.x >> 3) : gridDim.x) * C::W;
    uint32_t wi = (xw ? (blockIdx.x >> 3) : blockIdx.x) * C::W + swave;
    auto next_tile = [&]() -> int64_t {
        for (;;) {
            int64_t t = wi;
            if (xw) t = xcd_tile((int64_t)wi, (int)(blockIdx.x & 7), A.xcd_chunk, (int64_t)tiles);
            wi += w_step;
            if (t < 0) return -1;
            if (t < (int64_t)tiles) return t;
            if (!xw) return -1;
        }
    };

    int gi[G], nb_mine = -1;
    float4 gq[G];
    float qx = 0.f, qy = 0.f, qz = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    bool mine_valid = false;
    uint32_t goff_mine = 0;
    uint32_t mo0 = 0, mo1 = 0;           // ORD: the tile's two point rows (order[2 t], order[2 t + 1]; wave-uniform values)
    auto request_idx = [&](uint32_t tile) {
        const uint32_t lim = (m_tot - tile * 2) >= 2 ? 32u : 16u; // rows of the tile that exist
        if constexpr (ORD) {
            mo0 = (uint32_t)A.order[tile * 2];
            mo1 = lim == 32u ? (uint32_t)A.order[tile * 2 + 1] : mo0;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const uint32_t row = (uint32_t)(lane + 64 * i) / Q;
                gi[i] = row < lim ? __builtin_nontemporal_load(A.nidx + (int64_t)(row < RK ? mo0 : mo1) * RK + (row & (RK - 1))) : -1;
            }
            nb_mine = (uint32_t)lane < lim
                          ? __builtin_nontemporal_load(A.nidx + (int64_t)(lane < RK ? mo0 : mo1) * RK + (lane & (RK - 1))) : -1;
        } else {
            const int32_t* nb = A.nidx + (int64_t)tile * 2 * RK;  // 32 contiguous (point, neighbour) rows
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const uint32_t row = (uint32_t)(lane + 64 * i) / Q;
                gi[i] = row < lim ? __builtin_nontemporal_load(nb + row) : -1;
            }
            nb_mine = (uint32_t)lane < lim ? __builtin_nontemporal_load(nb + lane) : -1;
        }
    };
    auto request_data = [&](uint32_t tile) {
        // the tile's two points: cloud b0 / local index l0, and its successor (possibly the next cloud's first point);
        // ORD: two arbitrary rows, each located by its own scalar division
        uint32_t b0, l0, b1, l1;
        if constexpr (ORD) {
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mo0);
            const uint32_t m1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mo1);
            b0 = m0 / n_pts; l0 = m0 - b0 * n_pts;
            b1 = m1 / n_pts; l1 = m1 - b1 * n_pts;
        } else {
            const uint32_t m0 = tile * 2;
            b0 = m0 / n_pts; l0 = m0 - b0 * n_pts;
            const bool wrap = l0 + 1 == n_pts;
            b1 = wrap ? b0 + 1 : b0; l1 = wrap ? 0u : l0 + 1;
        }
        const float* f0 = A.gfeat + (int64_t)b0 * n_pts * H;      // scalar bases, 32-bit lane offsets
        const float* f1 = A.gfeat + (int64_t)b1 * n_pts * H;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int e = lane + 64 * i;
            const int row = e / Q, q = e - row * Q;
            gq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gi[i] >= 0) gq[i] = *reinterpret_cast<const float4*>((row < RK ? f0 : f1) + (uint32_t)gi[i] * H + 4 * q);
        }
        mine_valid = lane < 32 && nb_mine >= 0;
        if (mine_valid) {
            const bool p1 = lane >= RK;
            goff_mine = ((p1 ? b1 : b0) * n_pts + (uint32_t)nb_mine) * (uint32_t)(D * 4);
            const float* xb = A.xyz + 3 * ((int64_t)(p1 ? b1 : b0) * A.n0);
            const float* qp = xb + 3 * (p1 ? l1 : l0);
            const float* sp = xb + 3 * (uint32_t)nb_mine;
            qx = qp[0]; qy = qp[1]; qz = qp[2]; sx = sp[0]; sy = sp[1]; sz = sp[2];
        }
    };

    int64_t cur = next_tile();
    if (cur >= 0) { request_idx((uint32_t)cur); request_data((uint32_t)cur); }
    while (cur >= 0) {
        const int64_t nxt = next_tile();
        const uint32_t mo0_cur = mo0, mo1_cur = mo1;             // (request_idx(nxt) overwrites them)
        // ---- registers -> the wave's patch ---------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int e = lane + 64 * i;
            const int row = e / Q, q = e - row * Q;
            *reinterpret_cast<float4*>(X + row * XP + 4 * q) = gq[i];
        }
        if (lane < 32) {
            float* r = REL + lane * 12;
            if (mine_valid) {
                const float dx = qx - sx, dy = qy - sy, dz = qz - sz;
                r[0] = sqrtf(dx * dx + dy * dy + dz * dz);
                r[1] = dx; r[2] = dy; r[3] = dz; r[4] = qx; r[5] = qy; r[6] = qz; r[7] = sx; r[8] = sy; r[9] = sz;
            } else {
#pragma unroll
                for (int j = 0; j < 10; ++j) r[j] = 0.f;
            }
            r[10] = 1.f; r[11] = 0.f;                             // bias slot, K padding
            if constexpr (SPLIT) GOFF[lane] = mine_valid ? goff_mine : 0u;    // (rows past the data read row 0: never stored)
        }
        if (nxt >= 0) request_idx((uint32_t)nxt);
        wave_lds_sync();
        f32x16 sc[NT];
        if constexpr (SPLIT) {
            // the neighbours' per-point score halves become the accumulators' initial value
            const char* gbase = reinterpret_cast<const char*>(A.gscore);
            const uint32_t col4 = 4u * col;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const uint4 go = *reinterpret_cast<const uint4*>(GOFF + 8 * q4 + 4 * hi);     // rows mfma_row(4 q4 .. 4 q4 + 3)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    sc[t][4 * q4 + 0] = *reinterpret_cast<const float*>(gbase + (go.x + col4 + 128u * t));
                    sc[t][4 * q4 + 1] = *reinterpret_cast<const float*>(gbase + (go.y + col4 + 128u * t));
                    sc[t][4 * q4 + 2] = *reinterpret_cast<const float*>(gbase + (go.z + col4 + 128u * t));
                    sc[t][4 * q4 + 3] = *reinterpret_cast<const float*>(gbase + (go.w + col4 + 128u * t));
                }
            }
        }
        // ---- r1 = lrelu(lse1(rel)) (K = 12) -> X[:, H:] ---------------------------------------------------
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ar = REL + col * 12 + hi * 6;
            const float* wb = W1 + hi * 6 * 32 + col;
            const float2 a01 = *reinterpret_cast<const float2*>(ar);
            const float2 a23 = *reinterpret_cast<const float2*>(ar + 2);
            const float2 a45 = *reinterpret_cast<const float2*>(ar + 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a01.x, wb[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a01.y, wb[32], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a23.x, wb[64], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a23.y, wb[96], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a45.x, wb[128], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a45.y, wb[160], acc, 0, 0, 0);
            if (col < H) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {                  // slope product on register pairs (v_pk_mul_f32)
                    const v2f v = {acc[r], acc[r + 1]}, sv = v * 0.2f;
                    X[mfma_row(r, hi) * XP + H + col] = fmax_raw(v.x, sv.x);
                    X[mfma_row(r + 1, hi) * XP + H + col] = fmax_raw(v.y, sv.y);
                }
            }
        }
        wave_lds_sync();
        if constexpr (STAGE == 2) {
            // ---- r2 = lrelu(lse2(r1)) in place on X[:, H:] (every A read precedes the first write: the MFMAs sit between)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = l2bias;          // loop-invariant image: the first MFMA reads it as srcC
            const float* ar = X + col * XP + H + hi * (H / 2);
            const float* wb = W2 + hi * (H / 2) * 32 + col;
#pragma unroll
            for (int s4 = 0; s4 < H / 8; ++s4) {
                const float4 a = *reinterpret_cast<const float4*>(ar + 4 * s4);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wb[(4 * s4 + 0) * 32], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wb[(4 * s4 + 1) * 32], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wb[(4 * s4 + 2) * 32], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wb[(4 * s4 + 3) * 32], acc, 0, 0, 0);
            }
            wave_lds_sync();
            if (col < H) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {                  // slope product on register pairs (v_pk_mul_f32)
                    const v2f v = {acc[r], acc[r + 1]}, sv = v * 0.2f;
                    X[mfma_row(r, hi) * XP + H + col] = fmax_raw(v.x, sv.x);
                    X[mfma_row(r + 1, hi) * XP + H + col] = fmax_raw(v.y, sv.y);
                }
            }
            wave_lds_sync();
        }
        if (nxt >= 0) request_data((uint32_t)nxt);
        // ---- scores for all column tiles (A read once), softmax over the 16 neighbours, weighted sum --------
        if constexpr (!SPLIT) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[t][r] = sbias[t];
        }
        {
            const float* ar = X + col * XP + K0 + hi * (KS / 2);
            const float* wb = WS + hi * (KS / 2) * D + col;
#pragma unroll
            for (int s4 = 0; s4 < KS / 8; ++s4) {
                const float4 a = *reinterpret_cast<const float4*>(ar + 4 * s4);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, wb[(4 * s4 + 0) * D + 32 * t], sc[t], 0, 0, 0);
                    sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, wb[(4 * s4 + 1) * D + 32 * t], sc[t], 0, 0, 0);
                    sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, wb[(4 * s4 + 2) * D + 32 * t], sc[t], 0, 0, 0);
                    sc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, wb[(4 * s4 + 3) * D + 32 * t], sc[t], 0, 0, 0);
                }
            }
        }
        const uint32_t mi = (uint32_t)cur * 2 + hi;              // half 0 stores point 0, half 1 point 1
        const uint32_t m = ORD ? (hi ? mo1_cur : mo0_cur) : mi;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float* xc = X + 32 * t + col;
            float num[2], den[2];
            softmax_wsum8<XP>(sc[t], 0, xc + (4 * hi) * XP, num[0], den[0]);
            softmax_wsum8<XP>(sc[t], 8, xc + (16 + 4 * hi) * XP, num[1], den[1]);
            const float agg_mine = (hi ? num[1] : num[0]) / (hi ? den[1] : den[0]);
            // streaming data (the output rows, the neighbour indices) goes around the L2's LRU: the XCD's 4 MB are for the
            // cloud's gathered rows (features + gscore), which every tile re-reads at random
            if (mi < m_tot) __builtin_nontemporal_store(agg_mine, A.out + (int64_t)m * D + 32 * t + col);
        }
        wave_lds_sync();                                          // the patch is rewritten at the top of the loop
        cur = nxt;
    }
}

template <int D, int STAGE>
static int launch_attn_wave(LfaArgs a, hipStream_t st) {
    using C = WaveAttnCfg<D>;
    const int64_t tiles = (a.m_total + 1) / 2;
    static const int cus = device_cu_count();
    int64_t blocks = (tiles + C::W - 1) / C::W;
    unsigned grid = (unsigned)(blocks < cus ? blocks : cus);     // one 12/16-wave workgroup per CU (LDS-bound)
    const bool xcd_on = knobs().attn_xcd;
    a.xcd_chunk = xcd_on ? xcd_chunk_tiles(tiles, a.n > 0 ? a.m_total / a.n : 0) : 0;
    if (a.xcd_chunk > 0) grid = (grid + 7u) & ~7u;
    const size_t sm = sizeof(float) * ((size_t)C::WFLOATS + (size_t)C::W * C::PATCH);
    auto go = [&](auto kern) -> int {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)
            return ML3D_E_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::W * 64), sm, st, a);
        return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
    };
    if (a.gscore) return a.order ? go(lfa_attn_wave<D, STAGE, true, true>) : go(lfa_attn_wave<D, STAGE, true, false>);
    return a.order ? go(lfa_attn_wave<D, STAGE, false, true>) : go(lfa_attn_wave<D, STAGE, false, false>);
}

// ------------------------------------------------------------------------------------------------
// lfa_attn_b3 (D in {128, 256}, round 6) — the attention stage of lfa_attn_pf with every deep product on the BF16
// matrix pipe: a float is exactly h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m), a product of two
// bf16 is exact in the float accumulator, and  a b = ah bh + (ah bm + am bh) + (am bm + ah bl + al bh) + O(2^-25 |a b|)
// -- six v_mfma_f32_*_bf16 per 16-deep step against eight f32 MFMAs of TWICE the duration each (gemm.hip: measured
// error equal to the f32 kernel's).  What makes it pay here, where gemm.hip's form did not (three weight planes = 1.5 x
// the float footprint):
//   * ALL weights sit in REGISTERS, split once per workgroup: wave w owns the 32-column tile w % NT of the score matrix
//     (K = H, SPLIT form with the score bias inside gscore: 12 VGPRs per 16-deep step) and the 16-channel tile w % NC of lse2 (12 VGPRs per 32-deep step);
//     D = 256: 96 + 48 VGPRs of a 256-VGPR budget (8 waves per CU, 2 per SIMD).  The 64 KB of f32 lse2 weights leave LDS.
//   * the A operands (r1 = lse1(rel), r2 = lse2(r1)) are split ONCE, by the wave that produces them, on their way to LDS:
//     both lse products run TRANSPOSED (C^T = W^T . R^T on the 16 x 16 forms), so a lane ends up with 4 CONSECUTIVE
//     channels of one (point, neighbour) row = one ds_write_b64 per plane and one ds_write_b128 of the float row.
//   * 64 / 128 rows per tile instead of 32 / 64: every wave has work in every phase (lfa_attn_pf<256> ran lse2 on 4 of
//     its 8 waves) and the barriers are amortised over twice the rows.
// LDS: X [ROWS][D + 4] f32 (gathered features | r) for the weighted sum, ONE set of planes [3][ROWS][H + 8] bf16 that
// holds r1 and then r2 (a barrier between lse2's last read and its first write), REL [ROWS][13], NROWG [ROWS]:
// 122 KB (D = 256) / 130 KB (D = 128), one workgroup per CU.  Prefetch of the next tile's gathers exactly as lfa_attn_pf.
// lse1 (K = 10 + bias slot) stays on the f32 MFMA (16 x 16 x 4: three per 16 x 16 block, the same count the bf16x3 form needs).
// ------------------------------------------------------------------------------------------------
template <int D, int STAGE>
struct B3Cfg {
    static constexpr int H = D / 2;
    static constexpr int WAVES = 8, THREADS = WAVES * 64;
    // points per tile.  D = 256, stage 2 (144 VGPRs of weights): 64-row tiles need 56 VGPRs more than the wave has
    static constexpr int TP = D >= 256 ? (STAGE == 1 ? ML3D_B3_TP256_S1 : ML3D_B3_TP256_S2) : 8;
    static constexpr int ROWS = TP * RK;                    // 64 / 128 (point, neighbour) rows
    static constexpr int NT = D / 32;                       // score column tiles (32 wide)
    static constexpr int RT = ROWS / 32;                    // 32-row tiles
    static constexpr int RG = WAVES / NT;                   // waves that share a column tile take row tiles rg, rg + RG, ..
    static constexpr int SRT = RT / RG;                     // row tiles per wave in the score phase
    static constexpr int NC = H / 16;                       // lse channel tiles (16 wide, transposed products)
    static constexpr int RT16 = ROWS / 16;                  // 16-row tiles
    static constexpr int RG2 = WAVES / NC;
    static constexpr int LRT = RT16 / RG2;                  // 16-row tiles per wave in the lse phases
    static constexpr int KS = H / 16;                       // 16-deep steps of the score product (32 x 32 x 16)
    static constexpr int KS2 = H / 32;                      // 32-deep steps of lse2 (16 x 16 x 32)
    static constexpr int XP = D + 4;                        // float pitch of X
    static constexpr int PP = H + 8;                        // bf16 pitch of a plane row (16 bytes of padding: conflict-free b128 reads)
    static constexpr int RELP = 13;                         // float pitch of REL (odd: the 16 rows of a B fragment hit 16 banks)
    static constexpr int PLANE = ROWS * PP;                 // bf16 per plane
    static constexpr bool AHEAD = D < 256;                  // operand fragments of step s + 1 requested before the MFMAs of step s (12 VGPRs)
    static_assert(NT <= WAVES && WAVES % NT == 0 && RT % RG == 0, "score tiling");
    static_assert(NC <= WAVES && WAVES % NC == 0 && RT16 % RG2 == 0, "lse tiling");
    static constexpr size_t smem_bytes() {
        return (size_t)ROWS * XP * 4 + (size_t)3 * PLANE * 2 + (size_t)ROWS * RELP * 4 + (size_t)ROWS * 4;
    }
};

// four floats (consecutive channels) -> three packed bf16 quadruples
__device__ __forceinline__ void b3_split4(float v0, float v1, float v2, float v3, uint2& h, uint2& m, uint2& l) {
    h.x = bf16_pack2(v0, v1); h.y = bf16_pack2(v2, v3);
    const float r0 = sub_f32(v0, __uint_as_float(h.x << 16)), r1 = sub_f32(v1, __uint_as_float(h.x & 0xffff0000u));
    const float r2 = sub_f32(v2, __uint_as_float(h.y << 16)), r3 = sub_f32(v3, __uint_as_float(h.y & 0xffff0000u));
    m.x = bf16_pack2(r0, r1); m.y = bf16_pack2(r2, r3);
    l.x = bf16_pack2(sub_f32(r0, __uint_as_float(m.x << 16)), sub_f32(r1, __uint_as_float(m.x & 0xffff0000u)));
    l.y = bf16_pack2(sub_f32(r2, __uint_as_float(m.y << 16)), sub_f32(r3, __uint_as_float(m.y & 0xffff0000u)));
}
// eight floats (consecutive K of one weight column) -> the three bf16 operand fragments of an MFMA
__device__ __forceinline__ void b3_split8(const float (&v)[8], ml3d_u32x4& h, ml3d_u32x4& m, ml3d_u32x4& l) {
    uint2 h0, m0, l0, h1, m1, l1;
    b3_split4(v[0], v[1], v[2], v[3], h0, m0, l0);
    b3_split4(v[4], v[5], v[6], v[7], h1, m1, l1);
    h = (ml3d_u32x4){h0.x, h0.y, h1.x, h1.y};
    m = (ml3d_u32x4){m0.x, m0.y, m1.x, m1.y};
    l = (ml3d_u32x4){l0.x, l0.y, l1.x, l1.y};
}
// the six products of a three-way split, small terms first (the order of gemm.hip)
#define B3_PRODUCTS(MFMA, acc, a, b)                 \
    acc = MFMA(a[2], b[0], acc);                      \
    acc = MFMA(a[0], b[2], acc);                      \
    acc = MFMA(a[1], b[1], acc);                      \
    acc = MFMA(a[1], b[0], acc);                      \
    acc = MFMA(a[0], b[1], acc);                      \
    acc = MFMA(a[0], b[0], acc);

template <int D, int STAGE, bool ORD>
__global__ void __launch_bounds__((B3Cfg<D, STAGE>::THREADS / ML3D_B3_LB_DIV)) lfa_attn_b3(LfaArgs A) {
    using C = B3Cfg<D, STAGE>;
    constexpr int H = C::H, ROWS = C::ROWS, XP = C::XP, PP = C::PP, RELP = C::RELP, THREADS = C::THREADS;
    constexpr int Q = H / 4;                                      // float4 pieces per gathered row
    constexpr int G = ROWS * Q / THREADS;                         // pieces per thread
    static_assert(ROWS * Q % THREADS == 0 && ROWS <= THREADS, "tile shape");
    HIP_DYNAMIC_SHARED(float, smem)
    float* X = smem;                                              // [ROWS][XP]
    uint16_t* PL = reinterpret_cast<uint16_t*>(X + ROWS * XP);    // [3][ROWS][PP] bf16: r1, then r2
    float* REL = reinterpret_cast<float*>(PL + 3 * C::PLANE);     // [ROWS][RELP]
    uint32_t* NROWG = reinterpret_cast<uint32_t*>(REL + ROWS * RELP);   // [ROWS] byte offset of the neighbour's gscore row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, col = lane & 31;                    // 32 x 32 forms
    const int l16 = lane & 15, kq = lane >> 4;                    // 16 x 16 forms
    const int ct = wave % C::NT, rg = wave / C::NT;               // score: column tile, first row tile
    const int ch0 = (wave % C::NC) * 16, rg2 = wave / C::NC;      // lse: channel tile, first 16-row tile

    // ---- the wave's weights -> registers, split three ways (once per workgroup) --------------------------------
    ml3d_u32x4 bsw[C::KS][3];                                     // score_WT[H + k][ct * 32 + col]: B operand, 32 x 32 x 16
#pragma unroll
    for (int ks = 0; ks < C::KS; ++ks) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = A.score_wt[(int64_t)(H + 16 * ks + 8 * hi + e) * D + ct * 32 + col];
        b3_split8(v, bsw[ks][0], bsw[ks][1], bsw[ks][2]);
    }
    ml3d_u32x4 w2[STAGE == 2 ? C::KS2 : 1][3];                    // lse2_WT[k][ch0 + l16]: A operand (W^T), 16 x 16 x 32
    ml3d_f32x4 l2b = {0.f, 0.f, 0.f, 0.f};
    if constexpr (STAGE == 2) {
#pragma unroll
        for (int ks = 0; ks < C::KS2; ++ks) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = A.lse2_wt[(32 * ks + 8 * kq + e) * H + ch0 + l16];
            b3_split8(v, w2[ks][0], w2[ks][1], w2[ks][2]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) l2b[r] = A.lse2_b[ch0 + 4 * kq + r];      // C^T rows = channels
    }
    float w1[3];                                                  // lse1_WT[k = kq + 4 s][ch0 + l16]; K slot 10 = bias, 11 = 0
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int k = kq + 4 * s;
        w1[s] = k < 10 ? A.lse1_wt[k * H + ch0 + l16] : (k == 10 ? A.lse1_b[ch0 + l16] : 0.f);
    }

    const int64_t tiles = (A.m_total + C::TP - 1) / C::TP;
    const bool xw = A.xcd_chunk > 0;
    const int64_t w_step = xw ? (int64_t)(gridDim.x >> 3) : (int64_t)gridDim.x;
    int64_t wi = xw ? (int64_t)(blockIdx.x >> 3) : (int64_t)blockIdx.x;
    auto next_tile = [&]() -> int64_t {
        for (;;) {
            int64_t t = wi;
            if (xw) t = xcd_tile(wi, (int)(blockIdx.x & 7), A.xcd_chunk, tiles);
            wi += w_step;
            if (t < 0) return -1;
            if (t < tiles) return t;
            if (!xw) return -1;
        }
    };

    // ---- this thread's slots of a tile's inputs (as lfa_attn_pf) ---------------------------------------------------
    int gi[G], nb_mine = 0;
    float4 gq[G];
    float qx = 0.f, qy = 0.f, qz = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    bool mine_valid = false;
    uint32_t grow_mine = 0;
    const uint32_t n_pts = (uint32_t)A.n, m_tot = (uint32_t)A.m_total;
    uint32_t gmp[ORD ? G : 1], mp_mine = 0, m_first = 0;
    auto request_idx = [&](uint32_t tile) {
        const uint32_t have = m_tot - tile * C::TP;
        const uint32_t lim = (have < (uint32_t)C::TP ? have : (uint32_t)C::TP) * RK;
        if constexpr (ORD) {
            const int32_t* ord = A.order + tile * C::TP;
            m_first = (uint32_t)ord[0];
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const uint32_t row = (uint32_t)(tid + i * THREADS) / Q;
                gi[i] = -1;
                if (row < lim) {
                    gmp[i] = (uint32_t)ord[row / RK];
                    gi[i] = A.nidx[(int64_t)gmp[i] * RK + (row & (RK - 1))];
                }
            }
            if (tid < ROWS) {
                nb_mine = -1;
                if ((uint32_t)tid < lim) {
                    mp_mine = (uint32_t)ord[tid / RK];
                    nb_mine = A.nidx[(int64_t)mp_mine * RK + (tid & (RK - 1))];
                }
            }
        } else {
            const int32_t* nb = A.nidx + (int64_t)tile * (C::TP * RK);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const uint32_t row = (uint32_t)(tid + i * THREADS) / Q;
                gi[i] = row < lim ? nb[row] : -1;
            }
            if (tid < ROWS) nb_mine = (uint32_t)tid < lim ? nb[tid] : -1;
        }
    };
    auto request_data = [&](uint32_t tile) {
        const uint32_t m0 = ORD ? (uint32_t)__builtin_amdgcn_readfirstlane((int)m_first) : tile * C::TP;
        const uint32_t b0 = m0 / n_pts, l0 = m0 - b0 * n_pts;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int e = tid + i * THREADS;
            const int row = e / Q, q = e - row * Q;
            gq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gi[i] >= 0) {
                uint32_t b;
                if constexpr (ORD) b = b0 + (gmp[i] >= (b0 + 1) * n_pts ? 1u : 0u);
                else b = b0 + ((l0 + (uint32_t)(row / RK) >= n_pts) ? 1u : 0u);
                gq[i] = *reinterpret_cast<const float4*>(A.gfeat + ((int64_t)b * n_pts + (uint32_t)gi[i]) * H + 4 * q);
            }
        }
        if (tid < ROWS) {
            mine_valid = nb_mine >= 0;
            if (mine_valid) {
                uint32_t b, l;
                if constexpr (ORD) {
                    b = b0 + (mp_mine >= (b0 + 1) * n_pts ? 1u : 0u);
                    l = mp_mine - b * n_pts;
                } else {
                    const uint32_t lp = l0 + (uint32_t)(tid / RK);
                    const bool wrap = lp >= n_pts;
                    b = b0 + (wrap ? 1u : 0u);
                    l = wrap ? lp - n_pts : lp;
                }
                grow_mine = b * n_pts + (uint32_t)nb_mine;
                const float* xb = A.xyz + 3 * ((int64_t)b * A.n0);
                const float* qp = xb + 3 * l;
                const float* sp = xb + 3 * (uint32_t)nb_mine;
                qx = qp[0]; qy = qp[1]; qz = qp[2]; sx = sp[0]; sy = sp[1]; sz = sp[2];
            }
        }
    };
    // one 16 x 16 block of r (C^T layout: this lane holds channels ch0 + 4 kq .. + 3 of row `row`) -> the planes [+ X]
    auto store_r = [&](int row, const ml3d_f32x4& acc, bool to_x) {
        const float v0 = lrelu_max(acc[0], 0.2f), v1 = lrelu_max(acc[1], 0.2f), v2 = lrelu_max(acc[2], 0.2f),
                    v3 = lrelu_max(acc[3], 0.2f);
        uint2 h, m, l;
        b3_split4(v0, v1, v2, v3, h, m, l);
        uint16_t* d = PL + row * PP + ch0 + 4 * kq;
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + C::PLANE) = m;
        *reinterpret_cast<uint2*>(d + 2 * C::PLANE) = l;
        if (to_x) *reinterpret_cast<float4*>(X + row * XP + H + ch0 + 4 * kq) = make_float4(v0, v1, v2, v3);
    };

    int64_t cur = next_tile();
    if (cur >= 0) { request_idx((uint32_t)cur); request_data((uint32_t)cur); }
    while (cur >= 0) {
        const int64_t nxt = next_tile();
        const int64_t m_base = cur * C::TP;
        // ---- registers -> LDS -----------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int e = tid + i * THREADS;
            const int row = e / Q, q = e - row * Q;
            *reinterpret_cast<float4*>(X + row * XP + 4 * q) = gq[i];
        }
        if (tid < ROWS) {
            float* r = REL + tid * RELP;
            if (mine_valid) {
                const float dx = qx - sx, dy = qy - sy, dz = qz - sz;
                r[0] = sqrtf(dx * dx + dy * dy + dz * dz);
                r[1] = dx; r[2] = dy; r[3] = dz; r[4] = qx; r[5] = qy; r[6] = qz; r[7] = sx; r[8] = sy; r[9] = sz;
            } else {
#pragma unroll
                for (int j = 0; j < 10; ++j) r[j] = 0.f;
            }
            r[10] = 1.f; r[11] = 0.f;                              // bias slot, K padding
            NROWG[tid] = mine_valid ? grow_mine * (uint32_t)(D * 4) : 0u;    // (rows past the data read row 0: never stored)
        }
        if (nxt >= 0) request_idx((uint32_t)nxt);
        block_sync_lds();
        // ---- r1 = lrelu(lse1(rel)), transposed on the f32 MFMA (K = 12) -> planes (stage 1: + X[:, H:]) -------------
#pragma unroll 1
        for (int j = 0; j < C::LRT; ++j) {
            const int r0 = (rg2 + C::RG2 * j) * 16;
            const float* br = REL + (r0 + l16) * RELP + kq;
            ml3d_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[0], br[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[1], br[4], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[2], br[8], acc, 0, 0, 0);
            store_r(r0 + l16, acc, STAGE == 1);
        }
        // the neighbours' per-point score halves (gscore = f . W_top^T + bias, one row per POINT) become the score accumulators'
        // initial values: requested here, they land under the lse2 MFMAs / the barriers
        f32x16 sacc[C::SRT];
        {
            const char* gbase = reinterpret_cast<const char*>(A.gscore + ct * 32);    // wave-uniform base
            const uint32_t col4 = 4u * col;
#pragma unroll
            for (int j = 0; j < C::SRT; ++j) {
                const int rt = rg + C::RG * j;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const uint4 nr = *reinterpret_cast<const uint4*>(NROWG + rt * 32 + 8 * q4 + 4 * hi);   // rows mfma_row(4 q4 .. 4 q4 + 3)
                    sacc[j][4 * q4 + 0] = *reinterpret_cast<const float*>(gbase + (nr.x + col4));
                    sacc[j][4 * q4 + 1] = *reinterpret_cast<const float*>(gbase + (nr.y + col4));
                    sacc[j][4 * q4 + 2] = *reinterpret_cast<const float*>(gbase + (nr.z + col4));
                    sacc[j][4 * q4 + 3] = *reinterpret_cast<const float*>(gbase + (nr.w + col4));
                }
            }
        }
        if constexpr (STAGE == 2) {
            block_sync_lds();
            // ---- r2 = lrelu(lse2(r1)), transposed on the bf16 pipe; every read of r1 precedes the first write of r2 ------
            ml3d_f32x4 acc2[C::LRT];
#pragma unroll
            for (int j = 0; j < C::LRT; ++j) {
                const int r0 = (rg2 + C::RG2 * j) * 16;
                const uint16_t* br = PL + (r0 + l16) * PP + 8 * kq;
                acc2[j] = l2b;
                ml3d_u32x4 b[2][3];
                if constexpr (C::AHEAD) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[0][p] = *reinterpret_cast<const ml3d_u32x4*>(br + p * C::PLANE);
                }
#pragma unroll
                for (int ks = 0; ks < C::KS2; ++ks) {
                    if constexpr (C::AHEAD) {
                        if (ks + 1 < C::KS2) {
#pragma unroll
                            for (int p = 0; p < 3; ++p) b[(ks + 1) & 1][p] = *reinterpret_cast<const ml3d_u32x4*>(br + p * C::PLANE + 32 * (ks + 1));
                        }
                    } else {
#pragma unroll
                        for (int p = 0; p < 3; ++p) b[ks & 1][p] = *reinterpret_cast<const ml3d_u32x4*>(br + p * C::PLANE + 32 * ks);
                    }
                    B3_PRODUCTS(mfma_bf16_16x16x32, acc2[j], w2[ks], b[ks & 1])
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            block_sync_lds();
#pragma unroll
            for (int j = 0; j < C::LRT; ++j) store_r((rg2 + C::RG2 * j) * 16 + l16, acc2[j], true);
        }
        if (nxt >= 0) request_data((uint32_t)nxt);
        block_sync_lds();
        // ---- scores on the bf16 pipe (K = H: the position half; the feature half arrives as gscore), softmax, weighted sum ---
#pragma unroll
        for (int j = 0; j < C::SRT; ++j) {
            const int rt = rg + C::RG * j;
            f32x16 acc = sacc[j];
            // (the next step's three A fragments are requested before this step's six MFMAs; the scheduling barrier keeps the
            //  compiler from hoisting every step's loads to the top -- 96 VGPRs the weights need)
            const uint16_t* ar = PL + (rt * 32 + col) * PP + 8 * hi;
            ml3d_u32x4 a[2][3];
            if constexpr (C::AHEAD) {
#pragma unroll
                for (int p = 0; p < 3; ++p) a[0][p] = *reinterpret_cast<const ml3d_u32x4*>(ar + p * C::PLANE);
            }
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                if constexpr (C::AHEAD) {
                    if (ks + 1 < C::KS) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) a[(ks + 1) & 1][p] = *reinterpret_cast<const ml3d_u32x4*>(ar + p * C::PLANE + 16 * (ks + 1));
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[ks & 1][p] = *reinterpret_cast<const ml3d_u32x4*>(ar + p * C::PLANE + 16 * ks);
                }
                B3_PRODUCTS(mfma_bf16_32x32x16, acc, a[ks & 1], bsw[ks])
                __builtin_amdgcn_sched_barrier(0);
            }
            const float* xc = X + (rt * 32) * XP + ct * 32 + col;
            float num[2], den[2];
            softmax_wsum8<XP>(acc, 0, xc + (4 * hi) * XP, num[0], den[0]);
            softmax_wsum8<XP>(acc, 8, xc + (16 + 4 * hi) * XP, num[1], den[1]);
            const float agg_mine = (hi ? num[1] : num[0]) * __builtin_amdgcn_rcpf(hi ? den[1] : den[0]);     // (v_rcp_f32 + multiply, 1 ulp: as lfa_attn_mfma16)
            int64_t m = m_base + 2 * rt + hi;                    // half 0 stores point 0 of the row tile, half 1 point 1
            if (m < A.m_total) {
                if constexpr (ORD) m = A.order[m];
                A.out[m * D + ct * 32 + col] = agg_mine;
            }
        }
        block_sync_lds();
        cur = nxt;
    }
}

template <int D, int STAGE>
static int launch_attn_b3(LfaArgs a, hipStream_t st) {
    using C = B3Cfg<D, STAGE>;
    static const int cus = device_cu_count();
    const int64_t tiles = (a.m_total + C::TP - 1) / C::TP;
    unsigned grid = (unsigned)(tiles < cus ? tiles : cus);       // one 8-wave workgroup per CU, weights split once per workgroup
    a.xcd_chunk = knobs().attn_xcd ? xcd_chunk_tiles(tiles, a.n > 0 ? a.m_total / a.n : 0) : 0;
    if (a.xcd_chunk > 0) grid = (grid + 7u) & ~7u;
    const size_t sm = C::smem_bytes();
    auto go = [&](auto kern) -> int {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)
            return ML3D_E_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), sm, st, a);
        return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
    };
    return a.order ? go(lfa_attn_b3<D, STAGE, true>) : go(lfa_attn_b3<D, STAGE, false>);
}

// ------------------------------------------------------------------------------------------------
// lfa_attn_wave_b3 (D = 64, round 6) — lfa_attn_wave with lse2 and the score product on the BF16 matrix pipe (three-way
// split, six v_mfma_f32_32x32x16_bf16 per 16-deep step: see lfa_attn_b3) and NO LDS round trip for their A operands:
//   * the lse products run TRANSPOSED (C^T = W^T . R^T): a lane then holds, for ITS (point, neighbour) row, the channels
//     {0-3, 8-11, 16-19, 24-27} + 4 (lane / 32) -- four runs of four consecutive channels.  They are split three ways in
//     registers (packed bf16 pairs), and ONE v_permlane32_swap per dword pair trades the runs the two lanes of a row need from
//     each other: the result IS the operand fragment of the next product (row = lane % 32, K = 8 (lane / 32) + 16 step ..+7).
//     r1 and r2 never go to LDS as matrix operands; only the float r rows the weighted sum reads by COLUMN are stored
//     (four ds_write_b128 per lane).
//   * the weights sit in LDS once per workgroup as bf16 planes (score [3][64][32 + 8], lse2^T [3][32][32 + 8]: 23 KB -- the
//     f32 form took 21.5 KB), read as conflict-free ds_read_b128 fragments; lse1 (K = 10 + bias slot) keeps the f32 MFMA with its
//     weights in six registers.
// Per 32-row tile: 6 f32 MFMAs + 12 + 24 bf16 MFMAs = 1536 matrix cycles against 3456 for the f32 kernel, and the bf16 MFMAs
// leave the SIMD's VALU issue slots free (the f32 ones do not: tools/micro/mfma_valu_overlap.hip).
// ------------------------------------------------------------------------------------------------
template <int D>
struct WaveB3Cfg {
    static constexpr int H = D / 2;                 // 32
    static constexpr int NT = D / 32;               // score column tiles
    static constexpr int XP = D + 4;
    static constexpr int Q = H / 4, G = 32 * Q / 64;
    static constexpr int W = 12;                    // waves per workgroup
    static constexpr int PATCH = 32 * XP + 32 * 12 + 32;     // floats per wave: X, relative positions, gscore row offsets
    static constexpr int WP = H + 8;                // bf16 pitch of a weight row (K = H + 16 bytes: conflict-free b128 fragments)
    static constexpr int WSP = D * WP, W2P = H * WP;         // bf16 per plane
    static constexpr size_t smem_bytes() { return (size_t)3 * (WSP + W2P) * 2 + (size_t)H * 4 + (size_t)W * PATCH * 4; }
    static_assert(H == 32, "lfa_attn_wave_b3: D = 64");
};

template <int D, int STAGE, bool ORD>
__global__ void __launch_bounds__((WaveB3Cfg<D>::W * 64)) lfa_attn_wave_b3(LfaArgs A) {
    using C = WaveB3Cfg<D>;
    constexpr int H = C::H, XP = C::XP, Q = C::Q, G = C::G, NT = C::NT, WP = C::WP;
    HIP_DYNAMIC_SHARED(float, smem)
    uint16_t* WS = reinterpret_cast<uint16_t*>(smem);            // [3][D][WP]   score_WT[H + k][col] as planes[col][k]
    uint16_t* W2 = WS + 3 * C::WSP;                               // [3][H][WP]   lse2_WT[k][ch] as planes[ch][k]
    float* B2 = reinterpret_cast<float*>(W2 + 3 * C::W2P);        // [H]          lse2 bias
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, col = lane & 31;
    float* X = B2 + H + wave * C::PATCH;                          // [32][XP]  this wave's tile
    float* REL = X + 32 * XP;                                     // [32][12]
    uint32_t* GOFF = reinterpret_cast<uint32_t*>(REL + 32 * 12); // [32] byte offset of each row's gscore row

    // ---- weights -> bf16 planes in LDS (four consecutive K of one column per item) -------------------------------------
    for (int e = tid; e < D * (H / 4); e += C::W * 64) {
        const int c = e % D, k4 = (e / D) * 4;
        uint2 h, m, l;
        b3_split4(A.score_wt[(H + k4 + 0) * D + c], A.score_wt[(H + k4 + 1) * D + c], A.score_wt[(H + k4 + 2) * D + c],
                  A.score_wt[(H + k4 + 3) * D + c], h, m, l);
        uint16_t* d = WS + c * WP + k4;
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + C::WSP) = m;
        *reinterpret_cast<uint2*>(d + 2 * C::WSP) = l;
    }
    if constexpr (STAGE == 2) {
        for (int e = tid; e < H * (H / 4); e += C::W * 64) {
            const int c = e % H, k4 = (e / H) * 4;
            uint2 h, m, l;
            b3_split4(A.lse2_wt[(k4 + 0) * H + c], A.lse2_wt[(k4 + 1) * H + c], A.lse2_wt[(k4 + 2) * H + c],
                      A.lse2_wt[(k4 + 3) * H + c], h, m, l);
            uint16_t* d = W2 + c * WP + k4;
            *reinterpret_cast<uint2*>(d) = h;
            *reinterpret_cast<uint2*>(d + C::W2P) = m;
            *reinterpret_cast<uint2*>(d + 2 * C::W2P) = l;
        }
        for (int e = tid; e < H; e += C::W * 64) B2[e] = A.lse2_b[e];
    }
    // lse1^T A operand: W1^T[ch = col][k = 6 hi + s]; K slot 10 carries the bias (REL[:, 10] = 1), slot 11 is zero
    float w1[6];
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int k = 6 * hi + s;
        w1[s] = k < 10 ? A.lse1_wt[k * H + col] : (k == 10 ? A.lse1_b[col] : 0.f);
    }
    __syncthreads();                                  // the only workgroup barrier

    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t tiles = (uint32_t)((A.m_total + 1) / 2);       // 2 points per wave tile
    const uint32_t n_pts = (uint32_t)A.n, m_tot = (uint32_t)A.m_total;
    const bool xw = A.xcd_chunk > 0;
    const uint32_t w_step = (xw ? (gridDim.x >> 3) : gridDim.x) * C::W;
    uint32_t wi = (xw ? (blockIdx.x >> 3) : blockIdx.x) * C::W + swave;
    auto next_tile = [&]() -> int64_t {
        for (;;) {
            int64_t t = wi;
            if (xw) t = xcd_tile((int64_t)wi, (int)(blockIdx.x & 7), A.xcd_chunk, (int64_t)tiles);
            wi += w_step;
            if (t < 0) return -1;
            if (t < (int64_t)tiles) return t;
            if (!xw) return -1;
        }
    };

    int gi[G], nb_mine = -1;
    float4 gq[G];
    float qx = 0.f, qy = 0.f, qz = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    bool mine_valid = false;
    uint32_t goff_mine = 0;
    uint32_t mo0 = 0, mo1 = 0;
    auto request_idx = [&](uint32_t tile) {
        const uint32_t lim = (m_tot - tile * 2) >= 2 ? 32u : 16u;
        if constexpr (ORD) {
            mo0 = (uint32_t)A.order[tile * 2];
            mo1 = lim == 32u ? (uint32_t)A.order[tile * 2 + 1] : mo0;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const uint32_t row = (uint32_t)(lane + 64 * i) / Q;
                gi[i] = row < lim ? __builtin_nontemporal_load(A.nidx + (int64_t)(row < RK ? mo0 : mo1) * RK + (row & (RK - 1))) : -1;
            }
            nb_mine = (uint32_t)lane < lim
                          ? __builtin_nontemporal_load(A.nidx + (int64_t)(lane < RK ? mo0 : mo1) * RK + (lane & (RK - 1))) : -1;
        } else {
            const int32_t* nb = A.nidx + (int64_t)tile * 2 * RK;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const uint32_t row = (uint32_t)(lane + 64 * i) / Q;
                gi[i] = row < lim ? __builtin_nontemporal_load(nb + row) : -1;
            }
            nb_mine = (uint32_t)lane < lim ? __builtin_nontemporal_load(nb + lane) : -1;
        }
    };
    auto request_data = [&](uint32_t tile) {
        uint32_t b0, l0, b1, l1;
        if constexpr (ORD) {
            const uint32_t m0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mo0);
            const uint32_t m1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)mo1);
            b0 = m0 / n_pts; l0 = m0 - b0 * n_pts;
            b1 = m1 / n_pts; l1 = m1 - b1 * n_pts;
        } else {
            const uint32_t m0 = tile * 2;
            b0 = m0 / n_pts; l0 = m0 - b0 * n_pts;
            const bool wrap = l0 + 1 == n_pts;
            b1 = wrap ? b0 + 1 : b0; l1 = wrap ? 0u : l0 + 1;
        }
        const float* f0 = A.gfeat + (int64_t)b0 * n_pts * H;
        const float* f1 = A.gfeat + (int64_t)b1 * n_pts * H;
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int e = lane + 64 * i;
            const int row = e / Q, q = e - row * Q;
            gq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gi[i] >= 0) gq[i] = *reinterpret_cast<const float4*>((row < RK ? f0 : f1) + (uint32_t)gi[i] * H + 4 * q);
        }
        mine_valid = lane < 32 && nb_mine >= 0;
        if (mine_valid) {
            const bool p1 = lane >= RK;
            goff_mine = ((p1 ? b1 : b0) * n_pts + (uint32_t)nb_mine) * (uint32_t)(D * 4);
            const float* xb = A.xyz + 3 * ((int64_t)(p1 ? b1 : b0) * A.n0);
            const float* qp = xb + 3 * (p1 ? l1 : l0);
            const float* sp = xb + 3 * (uint32_t)nb_mine;
            qx = qp[0]; qy = qp[1]; qz = qp[2]; sx = sp[0]; sy = sp[1]; sz = sp[2];
        }
    };
    // lrelu of a transposed 32 x 32 result (this lane: row col, channels 8 g + 4 hi .. + 3, g = 0..3) -> the float rows of X[:, H:]
    // and the three-way split as operand fragments of the next product: frag[p][ks] = K 16 ks + 8 hi .. + 7 of row col
    auto finish_r = [&](const f32x16& acc, const float* bias /* LDS [H] or null */, bool to_x, ml3d_u32x4 (&frag)[3][2]) {
        uint2 pk[3][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) b = *reinterpret_cast<const float4*>(bias + 8 * g + 4 * hi);
            const float v0 = lrelu_max(acc[4 * g + 0] + b.x, 0.2f), v1 = lrelu_max(acc[4 * g + 1] + b.y, 0.2f),
                        v2 = lrelu_max(acc[4 * g + 2] + b.z, 0.2f), v3 = lrelu_max(acc[4 * g + 3] + b.w, 0.2f);
            if (to_x) *reinterpret_cast<float4*>(X + col * XP + H + 8 * g + 4 * hi) = make_float4(v0, v1, v2, v3);
            b3_split4(v0, v1, v2, v3, pk[0][g], pk[1][g], pk[2][g]);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                // runs g = 2 ks (channels 16 ks + 4 hi ..) and g = 2 ks + 1 (16 ks + 8 + 4 hi ..): the low half-wave needs the high one's
                // first run, the high half-wave the low one's second run
                uint32_t x0 = pk[p][2 * ks].x, x1 = pk[p][2 * ks].y, y0 = pk[p][2 * ks + 1].x, y1 = pk[p][2 * ks + 1].y;
                lane32_swap(x0, y0);
                lane32_swap(x1, y1);
                frag[p][ks] = (ml3d_u32x4){x0, x1, y0, y1};
            }
    };

    int64_t cur = next_tile();
    if (cur >= 0) { request_idx((uint32_t)cur); request_data((uint32_t)cur); }
    while (cur >= 0) {
        const int64_t nxt = next_tile();
        const uint32_t mo0_cur = mo0, mo1_cur = mo1;
        // ---- registers -> the wave's patch ---------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int e = lane + 64 * i;
            const int row = e / Q, q = e - row * Q;
            *reinterpret_cast<float4*>(X + row * XP + 4 * q) = gq[i];
        }
        if (lane < 32) {
            float* r = REL + lane * 12;
            if (mine_valid) {
                const float dx = qx - sx, dy = qy - sy, dz = qz - sz;
                r[0] = sqrtf(dx * dx + dy * dy + dz * dz);
                r[1] = dx; r[2] = dy; r[3] = dz; r[4] = qx; r[5] = qy; r[6] = qz; r[7] = sx; r[8] = sy; r[9] = sz;
            } else {
#pragma unroll
                for (int j = 0; j < 10; ++j) r[j] = 0.f;
            }
            r[10] = 1.f; r[11] = 0.f;                             // bias slot, K padding
            GOFF[lane] = mine_valid ? goff_mine : 0u;             // (rows past the data read row 0: never stored)
        }
        if (nxt >= 0) request_idx((uint32_t)nxt);
        wave_lds_sync();
        // the neighbours' per-point score halves (gscore, bias included) become the accumulators' initial value
        f32x16 sc[NT];
        {
            const char* gbase = reinterpret_cast<const char*>(A.gscore);
            const uint32_t col4 = 4u * col;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const uint4 go = *reinterpret_cast<const uint4*>(GOFF + 8 * q4 + 4 * hi);     // rows mfma_row(4 q4 .. 4 q4 + 3)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    sc[t][4 * q4 + 0] = *reinterpret_cast<const float*>(gbase + (go.x + col4 + 128u * t));
                    sc[t][4 * q4 + 1] = *reinterpret_cast<const float*>(gbase + (go.y + col4 + 128u * t));
                    sc[t][4 * q4 + 2] = *reinterpret_cast<const float*>(gbase + (go.z + col4 + 128u * t));
                    sc[t][4 * q4 + 3] = *reinterpret_cast<const float*>(gbase + (go.w + col4 + 128u * t));
                }
            }
        }
        ml3d_u32x4 frag[3][2];
        // ---- r1^T = lse1_W^T . rel^T on the f32 MFMA (K = 12); lrelu, float rows -> X[:, H:], split -> fragments -----------
        {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* br = REL + col * 12 + hi * 6;
            const float2 b01 = *reinterpret_cast<const float2*>(br);
            const float2 b23 = *reinterpret_cast<const float2*>(br + 2);
            const float2 b45 = *reinterpret_cast<const float2*>(br + 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[0], b01.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[1], b01.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[2], b23.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[3], b23.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[4], b45.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[5], b45.y, acc, 0, 0, 0);
            finish_r(acc, nullptr, STAGE == 1, frag);
        }
        if constexpr (STAGE == 2) {
            // ---- r2^T = lse2_W^T . r1^T on the bf16 pipe (A = weight planes from LDS, B = the fragments) ------------------------------
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const uint16_t* ar = W2 + col * WP + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                ml3d_u32x4 a[3], b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) { a[p] = *reinterpret_cast<const ml3d_u32x4*>(ar + p * C::W2P + 16 * ks); b[p] = frag[p][ks]; }
                B3_PRODUCTS(mfma_bf16_32x32x16, acc, a, b)
            }
            finish_r(acc, B2, true, frag);
        }
        if (nxt >= 0) request_data((uint32_t)nxt);
        // ---- scores (A = the fragments, B = weight planes from LDS), softmax over the 16 neighbours, weighted sum ------------
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            ml3d_u32x4 a[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = frag[p][ks];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint16_t* br = WS + (32 * t + col) * WP + 8 * hi + 16 * ks;
                ml3d_u32x4 b[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const ml3d_u32x4*>(br + p * C::WSP);
                B3_PRODUCTS(mfma_bf16_32x32x16, sc[t], a, b)
            }
        }
        wave_lds_sync();                                          // X[:, H:] of every row is written
        const uint32_t mi = (uint32_t)cur * 2 + hi;               // half 0 stores point 0, half 1 point 1
        const uint32_t m = ORD ? (hi ? mo1_cur : mo0_cur) : mi;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float* xc = X + 32 * t + col;
            float num[2], den[2];
            softmax_wsum8<XP>(sc[t], 0, xc + (4 * hi) * XP, num[0], den[0]);
            softmax_wsum8<XP>(sc[t], 8, xc + (16 + 4 * hi) * XP, num[1], den[1]);
            const float agg_mine = (hi ? num[1] : num[0]) * __builtin_amdgcn_rcpf(hi ? den[1] : den[0]);     // (v_rcp_f32 + multiply, 1 ulp: as lfa_attn_mfma16)
            if (mi < m_tot) __builtin_nontemporal_store(agg_mine, A.out + (int64_t)m * D + 32 * t + col);
        }
        wave_lds_sync();                                          // the patch is rewritten at the top of the loop
        cur = nxt;
    }
}

template <int D, int STAGE>
static int launch_attn_wave_b3(LfaArgs a, hipStream_t st) {
    using C = WaveB3Cfg<D>;
    const int64_t tiles = (a.m_total + 1) / 2;
    static const int cus = device_cu_count();
    int64_t blocks = (tiles + C::W - 1) / C::W;
    unsigned grid = (unsigned)(blocks < cus ? blocks : cus);     // one 12-wave workgroup per CU (LDS-bound)
    a.xcd_chunk = knobs().attn_xcd ? xcd_chunk_tiles(tiles, a.n > 0 ? a.m_total / a.n : 0) : 0;
    if (a.xcd_chunk > 0) grid = (grid + 7u) & ~7u;
    const size_t sm = C::smem_bytes();
    auto go = [&](auto kern) -> int {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)
            return ML3D_E_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::W * 64), sm, st, a);
        return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
    };
    return a.order ? go(lfa_attn_wave_b3<D, STAGE, true>) : go(lfa_attn_wave_b3<D, STAGE, false>);
}

// launches the attention part of one stage; `a.out` receives agg [m, D].  D <= 64: the per-wave kernel, D >= 128: the
// workgroup-tile prefetching kernel.  (Preconditions -- 32-bit point indices, at least one full tile per cloud -- are checked
// by the caller, which sends everything else to the generic VALU kernel lfa_stage.)
template <int D>
static bool attn_mfma_fits(const LfaArgs& a) {
    return a.m_total < ((int64_t)1 << 30) && a.n0 < ((int64_t)1 << 30) && (D <= 64 || a.n >= MfmaCfg<D>::TP);
}

template <int D, int STAGE>
static int launch_attn_mfma(LfaArgs a, hipStream_t st) {
    if constexpr (D <= 64) {
        if constexpr (D == 64 && ((ML3D_ATTN_B3) & 2) != 0) {
            if (a.gscore) return launch_attn_wave_b3<D, STAGE>(a, st);
        }
        return launch_attn_wave<D, STAGE>(a, st);
    } else {
        // the bf16x3 kernel wants the SPLIT form (gscore), its tile inside one or two consecutive clouds and 32-bit gscore offsets
        if constexpr (((ML3D_ATTN_B3) & 1) != 0) {
            if (a.gscore && a.n >= B3Cfg<D, 1>::TP) return launch_attn_b3<D, STAGE>(a, st);     // (the caller put the score bias into gscore)
        }
        using C = MfmaCfg<D>;
        const int grid_cap = knobs().attn_grid;   // tuning knob
        int64_t tiles = (a.m_total + C::TP - 1) / C::TP;
        unsigned grid = (unsigned)(tiles < grid_cap ? tiles : grid_cap);   // persistent-ish: weights load once per block
        a.xcd_chunk = knobs().attn_xcd ? xcd_chunk_tiles(tiles, a.n > 0 ? a.m_total / a.n : 0) : 0;
        if (a.xcd_chunk > 0) grid = (grid + 7u) & ~7u;
        const size_t sm = pf_smem_bytes<D, STAGE>();
        auto go = [&](auto kern) -> int {
            if (sm > 48 * 1024 &&
                hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)
                return ML3D_E_LAUNCH;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), sm, st, a);
            return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
        };
        if (a.gscore) return a.order ? go(lfa_attn_pf<D, STAGE, true, true>) : go(lfa_attn_pf<D, STAGE, true, false>);
        return a.order ? go(lfa_attn_pf<D, STAGE, false, true>) : go(lfa_attn_pf<D, STAGE, false, false>);
    }
}


// ------------------------------------------------------------------------------------------------
// lfa_attn_mfma16 — the D = 16 (first encoder layer, 45 056 points per frame) attention stage on
// v_mfma_f32_16x16x4_f32: one 16x16 MFMA tile = the 16 neighbours of ONE point x 16 channels.
// Build phase: one thread per (point, neighbour) row keeps everything in registers — relative
// position, r1 = lse1(rel) (10->8), stage 2: r2 = lse2(r1) (8->8) — gathers its neighbour's
// 8-float feature row (two 16-byte loads) and writes the 16-float X row to LDS.
// MFMA phase: lane group g = lane>>4 feeds k in [4g, 4g+4): the lane's four A values are ONE
// ds_read_b128; B = score_WT lives in 4 VGPRs.  C layout: lane holds column lane&15 and rows
// 4g..4g+3, so softmax over the 16 neighbours = 4 in-lane values + lane^16 and lane^32 exchanges.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int A16_TP = 16;           // points per tile (256 rows = 256 threads)
constexpr int A16_XP = 20;           // X row pitch

// (the lse weights arrive as separate __restrict__ kernel arguments: with `noalias` the compiler may read them with
//  scalar loads inside the tile loop -- SGPR operands of the packed FMAs -- instead of one uniform VECTOR load per
//  16 bytes of weights per tile, which is what it must do for pointers that could alias the stores to A.out)
// EPI (d_in = 8: the first encoder layer of every reference config): the per-point Linears that follow the stage run on the
// wave's pooled rows before they leave the CU -- stage 1: pool1.mlp (16 -> 8, lrelu 0.2) so `out` is p1 [m, 8]; stage 2:
// pool2.mlp (16 -> 16, lrelu 0.2), then lrelu_0.01(mlp2(.) + shortcut(feat_in)) as one [24] x [32] product so `out` is the
// layer's output [m, 32] (randlanet.py:639, 690-692).  A wave parks the pooled rows of FOUR tiles (16 points) in an LDS patch
// and then runs the Linears as 16x16x4 MFMAs with all 16 rows real; the [m, 16] pooled arrays and the chain launches that read
// them back disappear.  (Per tile -- 4 real rows of 16 -- the 16 extra MFMAs of stage 2 cost more than the chain launch they
// replace: 0.96 against 0.67 + 0.16 ms; on gfx950 an f32 MFMA is issue time the VALU-bound stage cannot spare.)
template <int STAGE> struct A16Epi {
    static constexpr int EP = STAGE == 1 ? 20 : 24;      // pitch of a parked row: [pooled / pool2 output (16) | feat_in (8)] (+ pad)
};

template <int STAGE, bool ORD, bool EPI>
__global__ void __launch_bounds__(256, (EPI && STAGE == 2) ? 5 : 6)       // (stage 2 + epilogue: 30.4 KB of LDS, five workgroups per CU)
lfa_attn_mfma16(LfaArgs A, const float* __restrict__ lse1_wt, const float* __restrict__ lse1_b,
                const float* __restrict__ lse2_wt, const float* __restrict__ lse2_b) {
    constexpr int D = 16, H = 8;
    __shared__ __attribute__((aligned(16))) float X[A16_TP * RK * A16_XP];
    constexpr int EP = A16Epi<STAGE>::EP;
    __shared__ __attribute__((aligned(16))) float E[EPI ? 4 * 16 * EP : 4];      // [wave][4 tiles x 4 points][EP]
    __shared__ __attribute__((aligned(16))) uint32_t EM[EPI ? 4 * 16 : 4];        // their output rows (~0u: none)
    // stage 2's epilogue weights live in LDS (20 registers otherwise, on a kernel that has 80):
    // [mlp2 (16 rows) ; shortcut (8 rows)] x 32 | pool2.mlp [16][16] | pool2 bias [16] | mlp2 + shortcut bias [32]
    constexpr int WC_POOL = 24 * 32, WC_PB = WC_POOL + 256, WC_CB = WC_PB + 16, WC_N = WC_CB + 32;
    __shared__ float WC[(EPI && STAGE == 2) ? WC_N : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, col = lane & 15;
    float bs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bs[s] = A.score_wt[(4 * g + s) * D + col];
    const float sbias = A.score_b[col];
    // EPI stage 1: B operands of pool1.mlp (16 -> 8; K index 4 g + s as for the scores) and its bias, in registers
    float wp[4] = {0.f, 0.f, 0.f, 0.f}, pbias = 0.f;
    if constexpr (EPI && STAGE == 1) {
        if (col < H) {
#pragma unroll
            for (int s = 0; s < 4; ++s) wp[s] = A.pool_wt[(4 * g + s) * H + col];
            pbias = A.pool_b[col];
        }
    }
    if constexpr (EPI && STAGE == 2) {
        for (int i = tid; i < WC_N; i += 256)
            WC[i] = i < 16 * 32 ? A.mlp2_wt[i] : i < WC_POOL ? A.short_wt[i - 16 * 32] : i < WC_PB ? A.pool_wt[i - WC_POOL] :
                    i < WC_CB ? A.pool_b[i - WC_PB] : A.mlp2_b[i - WC_CB] + A.short_b[i - WC_CB];
        __syncthreads();
    }
    // tile bookkeeping is wave-uniform and 32-bit (launcher: m_total, n0 < 2^30, n >= 16): the VALU shares the SIMD's
    // issue cycles with the f32 MFMAs on gfx950, a 64-bit division per lane would cost more than the tile's MFMAs
    const uint32_t tiles = (uint32_t)((A.m_total + A16_TP - 1) / A16_TP);
    const uint32_t n_pts = (uint32_t)A.n, m_tot = (uint32_t)A.m_total;
    const bool xw = A.xcd_chunk > 0;
    const uint32_t w_step = xw ? (gridDim.x >> 3) : gridDim.x;
    uint32_t wi = xw ? (blockIdx.x >> 3) : blockIdx.x;
    auto next_tile = [&]() -> int64_t {
        for (;;) {
            int64_t t = wi;
            if (xw) t = xcd_tile((int64_t)wi, (int)(blockIdx.x & 7), A.xcd_chunk, (int64_t)tiles);
            wi += w_step;
            if (t < 0) return -1;
            if (t < (int64_t)tiles) return t;
            if (!xw) return -1;
        }
    };
    // rows 64w .. 64w+63 of a tile (points 4w .. 4w+3) are built AND consumed by wave w: no workgroup barrier
    float* Xw = X + wave * 64 * A16_XP;
    const uint32_t p = tid >> 4;                                              // thread = (point p, neighbour tid & 15) row
    // this thread's row of the NEXT tile is requested while the current one is computed: its neighbour index during the
    // build phase, then (through that index) the two xyz triples and the 8-float feature row during the MFMA phase
    uint32_t nb = 0;
    uint32_t mrow = 0;                   // ORD: this thread's point row (order[tile * 16 + p])
    bool valid = false;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
    auto request_idx = [&](uint32_t tile) {
        valid = tile * A16_TP + p < m_tot;
        if constexpr (ORD) {
            if (valid) {
                mrow = (uint32_t)A.order[tile * A16_TP + p];
                nb = (uint32_t)A.nidx[(int64_t)mrow * RK + (tid & 15)];
            }
        } else {
            if (valid) nb = (uint32_t)A.nidx[(int64_t)tile * (A16_TP * RK) + tid];
        }
    };
    auto request_data = [&](uint32_t tile) {
        // ORD: the wave's 4 points are consecutive in a cloud-major order, so they lie in at most two clouds: the
        // first lane's cloud by one scalar division (all lanes take part in the broadcast), the rest by a compare
        uint32_t b0o = 0;
        if constexpr (ORD) b0o = (uint32_t)__builtin_amdgcn_readfirstlane((int)mrow) / n_pts;
        if (valid) {
            uint32_t b, nl;
            if constexpr (ORD) {
                b = b0o + (mrow >= (b0o + 1) * n_pts ? 1u : 0u);
                nl = mrow - b * n_pts;
            } else {
                const uint32_t m_base = tile * A16_TP;
                const uint32_t b0 = m_base / n_pts, l0 = m_base - b0 * n_pts; // scalar
                const uint32_t lp = l0 + p;
                const bool wrap = lp >= n_pts;                                // the tile may cross into the next cloud once
                b = b0 + (wrap ? 1u : 0u);
                nl = wrap ? lp - n_pts : lp;
            }
            const float* xb = A.xyz + 3 * ((int64_t)b * A.n0);
            const float* q = xb + 3 * nl;
            const float* sp = xb + 3 * nb;
            q0 = q[0]; q1 = q[1]; q2 = q[2]; s0 = sp[0]; s1 = sp[1]; s2 = sp[2];
            const float4* gf = reinterpret_cast<const float4*>(A.gfeat + ((int64_t)b * n_pts + nb) * H);
            g0 = gf[0]; g1 = gf[1];
        }
    };
    int parked = 0;                      // EPI: tiles whose pooled rows wait in the wave's LDS patch
    int64_t cur = next_tile();
    if (cur >= 0) { request_idx((uint32_t)cur); request_data((uint32_t)cur); }
    while (cur >= 0) {
        const int64_t nxt = next_tile();
        const uint32_t m_base = (uint32_t)cur * A16_TP;
        const uint32_t mrow_cur = mrow;                                       // (request_idx(nxt) below overwrites mrow)
        {   // ---- build this thread's X row ---------------------------------------------------------
            float xr[D];
#pragma unroll
            for (int c = 0; c < D; ++c) xr[c] = 0.f;
            const bool have = valid;
            float rel[10];
            rel[4] = q0; rel[5] = q1; rel[6] = q2; rel[7] = s0; rel[8] = s1; rel[9] = s2;
            xr[0] = g0.x; xr[1] = g0.y; xr[2] = g0.z; xr[3] = g0.w;
            xr[4] = g1.x; xr[5] = g1.y; xr[6] = g1.z; xr[7] = g1.w;
            if (nxt >= 0) request_idx((uint32_t)nxt);
            if (have) {
                rel[1] = rel[4] - rel[7]; rel[2] = rel[5] - rel[8]; rel[3] = rel[6] - rel[9];
                rel[0] = sqrtf(rel[1] * rel[1] + rel[2] * rel[2] + rel[3] * rel[3]);
                // lse1 (10 -> 8) and lse2 (8 -> 8) as packed-f32 FMAs (v_pk_fma_f32: two output channels per instruction,
                // weight pairs straight from SGPRs) -- the same rate as the f32 MFMA without its 16-column granularity
                typedef float v2f __attribute__((ext_vector_type(2)));
                const v2f* w1 = reinterpret_cast<const v2f*>(lse1_wt);
                const v2f* bb1 = reinterpret_cast<const v2f*>(lse1_b);
                v2f r1[H / 2];
#pragma unroll
                for (int c = 0; c < H / 2; ++c) {
                    v2f v = bb1[c];
#pragma unroll
                    for (int j = 0; j < 10; ++j) v = __builtin_elementwise_fma((v2f){rel[j], rel[j]}, w1[j * (H / 2) + c], v);
                    const v2f sv = v * 0.2f;
                    r1[c] = (v2f){fmax_raw(v.x, sv.x), fmax_raw(v.y, sv.y)};
                }
                if (STAGE == 2) {
                    const v2f* w2 = reinterpret_cast<const v2f*>(lse2_wt);
                    const v2f* bb2 = reinterpret_cast<const v2f*>(lse2_b);
#pragma unroll
                    for (int c = 0; c < H / 2; ++c) {
                        v2f v = bb2[c];
#pragma unroll
                        for (int j = 0; j < H; ++j) {
                            const float rj = (j & 1) ? r1[j >> 1].y : r1[j >> 1].x;
                            v = __builtin_elementwise_fma((v2f){rj, rj}, w2[j * (H / 2) + c], v);
                        }
                        const v2f sv = v * 0.2f;
                        xr[H + 2 * c] = fmax_raw(v.x, sv.x);
                        xr[H + 2 * c + 1] = fmax_raw(v.y, sv.y);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < H / 2; ++c) { xr[H + 2 * c] = r1[c].x; xr[H + 2 * c + 1] = r1[c].y; }
                }
            } else {
#pragma unroll
                for (int c = 0; c < H; ++c) xr[c] = 0.f;
            }
            float4* dst = reinterpret_cast<float4*>(X + tid * A16_XP);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) dst[q4] = make_float4(xr[4 * q4], xr[4 * q4 + 1], xr[4 * q4 + 2], xr[4 * q4 + 3]);
        }
        wave_lds_sync();
        if (nxt >= 0) request_data((uint32_t)nxt);
        // ---- MFMA: wave w owns points 4w .. 4w+3 of the tile ------------------------------------------
        constexpr float LOG2E = 1.4426950408889634f;                         // exp(s - max) = exp2(s * log2e - max * log2e)
        // EPI: this lane's point of the epilogue is point g of the wave (whose X rows it helped build: mrow_cur is its row)
        const uint32_t m_own = ORD ? mrow_cur : m_base + (uint32_t)(tid >> 4);
        const bool own_ok = m_base + (uint32_t)(tid >> 4) < m_tot;
        float fin = 0.f, keep = 0.f;
        if constexpr (EPI && STAGE == 2) {
            if (own_ok && col < 8) fin = A.feat_in[(int64_t)m_own * 8 + col];         // (in flight under the four points' MFMAs)
        }
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const float* Xp = Xw + pp * RK * A16_XP;
            float4 a = *reinterpret_cast<const float4*>(Xp + col * A16_XP + 4 * g);
            f32x4 acc = {sbias, sbias, sbias, sbias};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bs[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bs[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bs[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bs[3], acc, 0, 0, 0);
            float mx = fmax_raw(fmax_raw(acc[0], acc[1]), fmax_raw(acc[2], acc[3]));
            mx = fmax_raw(mx, __shfl_xor(mx, 16));
            mx = fmax_raw(mx, __shfl_xor(mx, 32));
            const float nmx = -mx * LOG2E;
            float sum = 0.f, ag = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(acc[r], LOG2E, nmx));
                sum += e;
                ag = fmaf(e, Xp[(4 * g + r) * A16_XP + col], ag);
            }
            sum += __shfl_xor(sum, 16); ag += __shfl_xor(ag, 16);
            sum += __shfl_xor(sum, 32); ag += __shfl_xor(ag, 32);
            const uint32_t mi = m_base + 4 * wave + pp;                       // position of the point in the walk
            // ORD: the point's row is held by the lanes that built its X rows (lanes 16 pp .. 16 pp + 15 of this wave)
            const uint32_t m = ORD ? (uint32_t)__builtin_amdgcn_readlane((int)mrow_cur, 16 * pp) : mi;
            // (v_rcp_f32 + multiply: 2 instructions against ~10 for the IEEE division, 1 ulp -- this kernel is VALU-issue bound,
            //  profiles/r03_pmc_forward.md; the 1e-4 parity gate has five orders of magnitude of room)
            const float pooled = ag * __builtin_amdgcn_rcpf(sum);
            if constexpr (EPI) {
                keep = pp == g ? pooled : keep;
                (void)m; (void)mi;
            } else {
                if (g == 0 && mi < m_tot) A.out[(int64_t)m * D + col] = pooled;
            }
        }
        if constexpr (EPI) {
            float* Ew = E + wave * 16 * EP;
            uint32_t* Mw = EM + wave * 16;
            const int slot = 4 * parked + g;                                  // (parked: wave-uniform)
            Ew[slot * EP + col] = keep;
            if constexpr (STAGE == 2) { if (col < 8) Ew[slot * EP + 16 + col] = fin; }
            if (col == 0) Mw[slot] = own_ok ? m_own : 0xffffffffu;
            ++parked;
            if (parked == 4 || nxt < 0) {
                // rows of tiles not parked this time hold older (or no) data: every output row depends on its own input row
                // only, and their row id says "none"
                if (col == 0) { for (int t4 = parked; t4 < 4; ++t4) Mw[4 * t4 + g] = 0xffffffffu; }
                parked = 0;
                wave_lds_sync();
                const float* er = Ew + col * EP;                              // A operand: row i = col, K = 4 g + s
                const float4 a = *reinterpret_cast<const float4*>(er + 4 * g);
                if constexpr (STAGE == 2) {
                    pbias = WC[WC_PB + col];
#pragma unroll
                    for (int s = 0; s < 4; ++s) wp[s] = WC[WC_POOL + (4 * g + s) * D + col];
                }
                f32x4 y = {pbias, pbias, pbias, pbias};
                y = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wp[0], y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wp[1], y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wp[2], y, 0, 0, 0);
                y = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wp[3], y, 0, 0, 0);
                // D: lane (col, g) holds rows 4 g .. 4 g + 3, column col
                const uint4 mo4 = *reinterpret_cast<const uint4*>(Mw + 4 * g);
                const uint32_t mo[4] = {mo4.x, mo4.y, mo4.z, mo4.w};
                if constexpr (STAGE == 1) {
                    if (col < H) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (mo[r] != 0xffffffffu) A.out[(int64_t)mo[r] * H + col] = lrelu(y[r], 0.2f);
                    }
                } else {
                    // pool2's output rows replace the pooled ones (every lane's A read above is older than these writes: one
                    // wave, LDS operations in order), the shortcut's input sits behind them
#pragma unroll
                    for (int r = 0; r < 4; ++r) Ew[(4 * g + r) * EP + col] = lrelu(y[r], 0.2f);
                    wave_lds_sync();
                    // K = 24: steps 0..3 take k = 4 g + s (the pool2 output), steps 4, 5 take k = 16 + 2 g + s' (the layer input)
                    const float4 a0 = *reinterpret_cast<const float4*>(er + 4 * g);
                    const float2 a1 = *reinterpret_cast<const float2*>(er + 16 + 2 * g);
                    const float av[6] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y};
                    const float cb0 = WC[WC_CB + col], cb1 = WC[WC_CB + 16 + col];
                    f32x4 z0 = {cb0, cb0, cb0, cb0}, z1 = {cb1, cb1, cb1, cb1};
#pragma unroll
                    for (int s6 = 0; s6 < 6; ++s6) {
                        const int krow = s6 < 4 ? 4 * g + s6 : 16 + 2 * g + (s6 - 4);
                        z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s6], WC[krow * 32 + col], z0, 0, 0, 0);
                        z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s6], WC[krow * 32 + 16 + col], z1, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (mo[r] != 0xffffffffu) {
                            float* o = A.out + (int64_t)mo[r] * 32;
                            o[col] = lrelu(z0[r], 0.01f);
                            o[16 + col] = lrelu(z1[r], 0.01f);
                        }
                    }
                }
            }
        }
        wave_lds_sync();
        cur = nxt;
    }
}

// epilogue: the stage's per-point Linears inside the kernel (a.out = p1 [m, 8] / the layer output [m, 32]); needs a.d_in == 8
template <int STAGE>
static int launch_attn_mfma16(LfaArgs a, hipStream_t st, bool epilogue) {
    int64_t tiles = (a.m_total + A16_TP - 1) / A16_TP;
    const bool xcd_on = knobs().attn_xcd;
    a.xcd_chunk = xcd_on ? xcd_chunk_tiles(tiles, a.n > 0 ? a.m_total / a.n : 0) : 0;
    const int cap = knobs().attn16_grid;
    unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    if (a.xcd_chunk > 0) grid = (grid + 7u) & ~7u;
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a, a.lse1_wt, a.lse1_b, a.lse2_wt, a.lse2_b);
    };
    if (epilogue) { if (a.order) go(lfa_attn_mfma16<STAGE, true, true>); else go(lfa_attn_mfma16<STAGE, false, true>); }
    else { if (a.order) go(lfa_attn_mfma16<STAGE, true, false>); else go(lfa_attn_mfma16<STAGE, false, false>); }
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// Per-point MLP chains: up to 4 Linear(+folded BN)+activation layers back to back, intermediate activations in LDS.
// First-layer input rows may be [a0 | a1[gather]] (nearest_interpolation + cat of the decoder, randlanet.py:288-291); one
// optional extra input `cat` is appended to the input of layer `cat_layer` (mlp2(p2) + shortcut(feat) = one Linear over
// [p2 | feat], randlanet.py:692).  Chains run fused only when their shape has a compiled mlp_wave_s instance below; every
// other shape runs one tile GEMM (gemm.hip) per Linear.
// ------------------------------------------------------------------------------------------------
constexpr int CH_MAX = 4;

struct ChainLayer {
    const float* wt;     // [cin][cout]
    const float* bias;
    const float* bias2;  // optional
    int cin, cout;
    int act;             // 0 none, 1 leaky relu
    float slope;
};

struct ChainArgs {
    const float* a0; int c0;
    const float* a1; int c1;
    const int32_t* gather;
    int64_t rows_per_item, a1_rows_per_item;
    const float* cat; int cat_c; int cat_layer;
    int n_layers;
    ChainLayer L[CH_MAX];
    float* out;
    int64_t m_total;
};

// one 32x32 output tile's K loop with compile-time trip count and weight pitch
template <int KH, int NP>
__device__ __forceinline__ f32x16 wave_k_loop(const float* __restrict__ arow, const float* __restrict__ Bc, f32x16 acc) {
#pragma unroll
    for (int s4 = 0; s4 < KH; s4 += 4) {
        const float4 a = *reinterpret_cast<const float4*>(arow + s4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, Bc[(s4 + 0) * NP], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, Bc[(s4 + 1) * NP], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, Bc[(s4 + 2) * NP], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, Bc[(s4 + 3) * NP], acc, 0, 0, 0);
    }
    __builtin_amdgcn_iglp_opt(0);
    return acc;
}

// ------------------------------------------------------------------------------------------------
// mlp_wave_s — mlp_wave with the chain's SHAPE as a compile-time parameter, for the shapes RandLA-Net ships
// (fc1 with the last decoder stage in front of it; the pool2 + mlp2|shortcut chains of the first two encoder
// layers).  The runtime-shaped kernel spends more VALU instructions on index arithmetic (runtime pitches,
// 64-bit row addresses, divisions by runtime widths) than on its epilogues, and on gfx950 every VALU
// instruction is issue time taken from the f32 MFMAs.  Here every pitch, K trip count and LDS offset is a
// constant, the tile index is scalar, rows are addressed as scalar base + 32-bit lane offset, and the K loops
// are straight-line code.  Same numerics, same LDS-resident weight image, same barrier-free per-wave tiles.
// ------------------------------------------------------------------------------------------------
template <int C0_, int C1_, int CATL_, int CATC_, int NL_, int N0_, int N1_, int N2_, int N3_>
struct MlpShape {
    static constexpr int C0 = C0_, C1 = C1_, CATL = CATL_, CATC = CATC_, NL = NL_;
    static constexpr int n(int l) { return l == 0 ? N0_ : l == 1 ? N1_ : l == 2 ? N2_ : N3_; }          // real widths (last: upper bound)
    static constexpr int np(int l) { return (n(l) + 31) & ~31; }
    static constexpr int cin(int l) { return l == 0 ? C0 + C1 : n(l - 1) + (l == CATL ? CATC : 0); }
    static constexpr int w_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += cin(i) * np(i) + np(i); return o; }
    static constexpr int b_off(int l) { return w_off(l) + cin(l) * np(l); }
    static constexpr int w_total() { return (w_off(NL) + 3) & ~3; }
    static constexpr int pit(int par) { int w = 4; for (int l = par; l < NL; l += 2) w = cin(l) > w ? cin(l) : w; return w + 4; }
    static constexpr int patch() { return 32 * (pit(0) + pit(1)); }
};

template <class S, int L>
__device__ __forceinline__ void mlp_s_layer(const ChainArgs& A, float* smem, float* P0, float* P1, uint32_t m_row0, uint32_t m_tot,
                                            int hi, int cl, int lane) {
    constexpr int CIN = S::cin(L), NP = S::np(L), KH = CIN / 2;
    constexpr int PIN = S::pit(L & 1), POUT = S::pit((L + 1) & 1);
    constexpr bool LAST = L == S::NL - 1;
    const float* in = (L & 1) ? P1 : P0;
    float* outp = (L & 1) ? P0 : P1;
    const ChainLayer& Ly = A.L[L];
    const float* arow = in + cl * PIN + hi * KH;
#pragma unroll
    for (int ct = 0; ct < NP / 32; ++ct) {
        const float* Bc = smem + S::w_off(L) + (hi * KH) * NP + ct * 32 + cl;
        f32x16 acc;
        const float b = smem[S::b_off(L) + ct * 32 + cl];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b;
        acc = wave_k_loop<KH, NP>(arow, Bc, acc);
        const int col = ct * 32 + cl;
        if (LAST) {
            if (col < Ly.cout) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t m = m_row0 + mfma_row(r, hi);
                    float v = acc[r];
                    if (Ly.act) v = lrelu_max(v, Ly.slope);
                    if (m < m_tot) A.out[(int64_t)m * Ly.cout + col] = v;
                }
            }
        } else {
            // (hidden widths are the compile-time ones; columns >= n(L) of a padded tile are never read back)
            if (S::n(L) % 32 == 0 || col < S::n(L)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) outp[mfma_row(r, hi) * POUT + col] = lrelu_max(acc[r], Ly.slope);
            }
        }
    }
    if constexpr (!LAST) {
        if constexpr (L + 1 == S::CATL) {                                   // the next layer's extra input columns
            constexpr int QC = S::CATC / 4;
#pragma unroll
            for (int i = 0; i < (32 * QC + 63) / 64; ++i) {
                const int e = lane + 64 * i;
                if (32 * QC % 64 == 0 || e < 32 * QC) {
                    const int r = e / QC, k = (e - r * QC) * 4;
                    const uint32_t m = m_row0 + r;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m < m_tot) v = *reinterpret_cast<const float4*>(A.cat + (int64_t)m * S::CATC + k);
                    *reinterpret_cast<float4*>(outp + r * POUT + S::n(L) + k) = v;
                }
            }
        }
        wave_lds_sync();
        mlp_s_layer<S, L + 1>(A, smem, P0, P1, m_row0, m_tot, hi, cl, lane);
    }
}

template <class S, int NW>
__global__ void __launch_bounds__(NW * 64, 2) mlp_wave_s(ChainArgs A) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, cl = lane & 31;
    // ---- weight image (zero-padded 32-column tiles) + biases ------------------------------------------------
#pragma unroll
    for (int l = 0; l < S::NL; ++l) {
        const ChainLayer Ly = A.L[l];
        const int np = S::np(l), cin = S::cin(l);
        for (int e = tid; e < cin * np; e += NW * 64) {
            const int k = e / np, c = e - k * np;
            smem[S::w_off(l) + e] = c < Ly.cout ? Ly.wt[k * Ly.cout + c] : 0.f;
        }
        for (int c = tid; c < np; c += NW * 64) {
            float b = 0.f;
            if (c < Ly.cout) { if (Ly.bias) b = Ly.bias[c]; if (Ly.bias2) b += Ly.bias2[c]; }
            smem[S::b_off(l) + c] = b;
        }
    }
    __syncthreads();
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    float* P0 = smem + S::w_total() + swave * S::patch();
    float* P1 = P0 + 32 * S::pit(0);
    const uint32_t m_tot = (uint32_t)A.m_total;
    const uint32_t tiles = (m_tot + 31) / 32;
    const uint32_t t_step = gridDim.x * NW;
    constexpr int Q0 = (S::C0 + S::C1) / 4, QA = S::C0 / 4;
    constexpr int PRE = (32 * Q0 + 63) / 64;
    const uint32_t rpi = (uint32_t)A.rows_per_item, srpi = (uint32_t)A.a1_rows_per_item;

    // first-layer input rows of the NEXT tile are requested while the current one is computed; gathered rows
    // ([skip | nearest_interpolation(coarser)] of the decoder) take their index one step earlier still
    float4 pre[PRE];
    int gix[PRE];
    auto request_idx = [&](uint32_t t) {
        if constexpr (S::C1 > 0) {
#pragma unroll
            for (int i = 0; i < PRE; ++i) {
                const int e = lane + 64 * i;
                const int r = e / Q0, q = e - r * Q0;
                const uint32_t m = t * 32 + r;
                gix[i] = -1;
                if ((32 * Q0 % 64 == 0 || e < 32 * Q0) && q >= QA && m < m_tot) gix[i] = A.gather ? A.gather[m] : (int)m;
            }
        }
    };
    auto request_rows = [&](uint32_t t) {
        const uint32_t m0 = t * 32;
        uint32_t it0 = 0, l0 = m0;                                         // item of the tile's first row (scalar)
        if (S::C1 > 0 && A.gather) { it0 = m0 / rpi; l0 = m0 - it0 * rpi; }
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int e = lane + 64 * i;
            const int r = e / Q0, q = e - r * Q0;
            const uint32_t m = m0 + r;
            pre[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((32 * Q0 % 64 == 0 || e < 32 * Q0) && m < m_tot) {
                if (S::C1 == 0 || q < QA) {
                    pre[i] = *reinterpret_cast<const float4*>(A.a0 + (int64_t)m * S::C0 + 4 * q);
                } else {
                    int64_t grow = gix[i];
                    if (A.gather) grow += (int64_t)(it0 + ((l0 + r >= rpi) ? 1u : 0u)) * srpi;   // a tile crosses items at most once
                    pre[i] = *reinterpret_cast<const float4*>(A.a1 + grow * S::C1 + 4 * (q - QA));
                }
            }
        }
    };
    uint32_t t = blockIdx.x * NW + swave;
    if (t < tiles) { request_idx(t); request_rows(t); }
    if (t + t_step < tiles) request_idx(t + t_step);
    for (; t < tiles; t += t_step) {
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int e = lane + 64 * i;
            if (32 * Q0 % 64 == 0 || e < 32 * Q0) {
                const int r = e / Q0, q = e - r * Q0;
                *reinterpret_cast<float4*>(P0 + r * S::pit(0) + 4 * q) = pre[i];
            }
        }
        if (t + t_step < tiles) {
            request_rows(t + t_step);
            if (t + 2 * t_step < tiles) request_idx(t + 2 * t_step);
        }
        wave_lds_sync();
        mlp_s_layer<S, 0>(A, smem, P0, P1, t * 32, m_tot, hi, cl, lane);
        wave_lds_sync();          // patch 0 is rewritten at the top of the next tile
    }
}

// the shapes with a compiled instance (last entry of each: upper bound of the final width, any cout below it works)
typedef MlpShape<32, 32, -1, 0, 4, 32, 64, 32, 32> ShapeDecFc1;     // [skip 32 | interp 32] -> 32 -> 64 -> 32 -> classes
typedef MlpShape<32, 0, -1, 0, 3, 64, 32, 32, 0> ShapeFc1;          // 32 -> 64 -> 32 -> classes
typedef MlpShape<64, 0, 1, 32, 2, 64, 128, 0, 0> ShapeEnc64;        // pool2 64 -> 64, [. | feat 32] -> 128
typedef MlpShape<16, 0, 1, 8, 2, 16, 32, 0, 0> ShapeEnc16;          // pool2 16 -> 16, [. | feat 8] -> 32

// single narrow Linears over the finest levels (memory-bound: what matters is one pass, no 64-column tile padding)
typedef MlpShape<16, 0, -1, 0, 1, 8, 0, 0, 0> ShapeLin16x8;         // pool1.mlp of the first encoder layer
typedef MlpShape<8, 0, -1, 0, 1, 8, 0, 0, 0> ShapeLin8x8;           // mlp1 of the first encoder layer
typedef MlpShape<64, 0, -1, 0, 1, 32, 0, 0, 0> ShapeLin64x32;       // pool1.mlp of the second
typedef MlpShape<32, 0, -1, 0, 1, 32, 0, 0, 0> ShapeLin32x32;       // mlp1 of the second
typedef MlpShape<32, 0, -1, 0, 1, 64, 0, 0, 0> ShapeLin32x64;       // per-point score half (gscore) of the second

template <class S>
static bool mlp_shape_matches(const ChainArgs& a) {
    if (a.n_layers != S::NL || a.c0 != S::C0 || (a.a1 ? a.c1 : 0) != S::C1) return false;
    if (S::CATL >= 0 ? !(a.cat && a.cat_layer == S::CATL && a.cat_c == S::CATC) : (a.cat != nullptr)) return false;
    for (int l = 0; l < S::NL; ++l) {
        if (a.L[l].cin != S::cin(l)) return false;
        if (l + 1 < S::NL ? a.L[l].cout != S::n(l) : a.L[l].cout > S::n(l)) return false;
        if (l + 1 < S::NL && !(a.L[l].act && a.L[l].slope > 0.f && a.L[l].slope < 1.f)) return false;
    }
    const ChainLayer& last = a.L[S::NL - 1];
    if (last.act && !(last.slope > 0.f && last.slope < 1.f)) return false;
    if (a.m_total >= ((int64_t)1 << 30)) return false;
    if (S::C1 > 0 && a.gather && (a.rows_per_item < 32 || a.rows_per_item >= ((int64_t)1 << 30))) return false;
    return true;
}

template <class S, int NW>
static int launch_mlp_wave_s(const ChainArgs& a, hipStream_t st) {
    const size_t sm = sizeof(float) * ((size_t)S::w_total() + (size_t)NW * S::patch());
    if (sm > 48 * 1024 &&
        hipFuncSetAttribute((const void*)mlp_wave_s<S, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)
        return ML3D_E_LAUNCH;
    static const int cus = device_cu_count();
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)mlp_wave_s<S, NW>, NW * 64, sm) != hipSuccess || occ < 1) occ = 1;
    const int64_t tiles = (a.m_total + 31) / 32;
    const int64_t want = (tiles + NW - 1) / NW, cap = (int64_t)occ * cus;
    hipLaunchKernelGGL((mlp_wave_s<S, NW>), dim3((unsigned)(want < cap ? want : cap)), dim3(NW * 64), sm, st, a);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// mlp_chain_b3 (round 6) — the per-point chains of mlp_wave_s on the BF16 matrix pipe with the activations in REGISTERS from the first
// layer's input rows to the last layer's output: no LDS patch at all.
//   * a lane loads ITS row's 8 consecutive input channels of a 16-deep step straight from global memory (two 16-byte loads; the
//     decoder's [skip | interp[idx]] rows and the encoder's extra `cat` columns are just further steps), splits them three ways
//     (b3_split8): that IS the MFMA operand fragment (row = lane % 32, K = 16 step + 8 (lane / 32) ..+7);
//   * hidden layers run TRANSPOSED (C^T = W^T . X^T: a lane gets runs of four consecutive output channels of its own row), lrelu +
//     split + one v_permlane32_swap per dword pair turn the result into the next layer's fragments (lfa_attn_wave_b3's finish);
//   * the LAST layer runs untransposed on the same fragments (C = X . W: lane = output column), so the output rows are stored
//     as contiguous runs (76 bytes of a 19-class row, 128 bytes of a 128-wide one) like mlp_wave_s does;
//   * all weights sit in LDS once per workgroup as W^T planes [3][cout_pad][cin + 8] bf16 (both forms read them K-contiguous per
//     output column), biases as floats.
// Six bf16 MFMAs per 16-deep step instead of eight f32 MFMAs of twice the duration, and the VALU issue slots beside them stay
// free (f32 MFMAs block them on gfx950).  Shapes: the same compile-time MlpShape instances as mlp_wave_s, widths multiples of 16.
// ------------------------------------------------------------------------------------------------
template <class S>
struct ChainB3 {
    static constexpr int wp(int l) { return S::cin(l) + 8; }                                  // bf16 pitch of a weight row
    static constexpr int plane(int l) { return S::np(l) * wp(l); }                            // bf16 per plane
    static constexpr int w_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += 3 * plane(i); return o; }     // in bf16
    static constexpr int w_total() { return (w_off(S::NL) + 7) & ~7; }
    static constexpr int b_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += S::np(i); return o; }          // in floats
    static constexpr size_t smem_bytes() { return (size_t)w_total() * 2 + (size_t)b_off(S::NL) * 4; }
    static constexpr int maxk() { int m = 0; for (int l = 0; l < S::NL; ++l) m = S::cin(l) > m ? S::cin(l) : m; return m; }
    static constexpr bool ok() {
        for (int l = 0; l < S::NL; ++l) if (S::cin(l) % 16 != 0) return false;
        for (int l = 0; l + 1 < S::NL; ++l) if (S::n(l) % 32 != 0) return false;
        return S::C0 % 16 == 0 && S::C1 % 16 == 0 && (S::CATL < 0 || S::CATC % 16 == 0);
    }
};

// layer L of the chain on the fragments `in` (K = cin(L)); recursion instead of a loop: every layer has its own compile-time widths
template <class S, int L>
__device__ __forceinline__ void chain_b3_layer(const ChainArgs& A, const uint16_t* __restrict__ WT, const float* __restrict__ BS,
                                                ml3d_u32x4 (&in)[ChainB3<S>::maxk() / 16][3], uint32_t m_row0, uint32_t m_tot,
                                                int hi, int cl) {
    using B = ChainB3<S>;
    constexpr int KS = S::cin(L) / 16, NP = S::np(L), WP = B::wp(L), PL = B::plane(L);
    constexpr bool LAST = L == S::NL - 1;
    const ChainLayer& Ly = A.L[L];
    const uint16_t* W = WT + B::w_off(L);
    const float* bias = BS + B::b_off(L);
    if constexpr (LAST) {
        // C = X . W: lane = output column, registers = rows; stored as contiguous row runs
#pragma unroll
        for (int ct = 0; ct < NP / 32; ++ct) {
            const int col = ct * 32 + cl;
            f32x16 acc;
            const float b = bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = b;
            const uint16_t* br = W + col * WP + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                ml3d_u32x4 w[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) w[p] = *reinterpret_cast<const ml3d_u32x4*>(br + p * PL + 16 * ks);
                B3_PRODUCTS(mfma_bf16_32x32x16, acc, in[ks], w)
            }
            if (col < Ly.cout) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t m = m_row0 + mfma_row(r, hi);
                    float v = acc[r];
                    if (Ly.act) v = lrelu_max(v, Ly.slope);
                    if (m < m_tot) A.out[(int64_t)m * Ly.cout + col] = v;
                }
            }
        }
    } else {
        // C^T = W^T . X^T: lane = row, registers = channels 32 ct + 8 g + 4 hi .. + 3; -> the next layer's fragments
        ml3d_u32x4 out[B::maxk() / 16][3];
#pragma unroll
        for (int ct = 0; ct < NP / 32; ++ct) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b = *reinterpret_cast<const float4*>(bias + 32 * ct + 8 * g + 4 * hi);
                acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
            }
            const uint16_t* ar = W + (32 * ct + cl) * WP + 8 * hi;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                ml3d_u32x4 w[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) w[p] = *reinterpret_cast<const ml3d_u32x4*>(ar + p * PL + 16 * ks);
                B3_PRODUCTS(mfma_bf16_32x32x16, acc, w, in[ks])
            }
            uint2 pk[3][4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b3_split4(lrelu_max(acc[4 * g + 0], Ly.slope), lrelu_max(acc[4 * g + 1], Ly.slope), lrelu_max(acc[4 * g + 2], Ly.slope),
                          lrelu_max(acc[4 * g + 3], Ly.slope), pk[0][g], pk[1][g], pk[2][g]);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    uint32_t x0 = pk[p][2 * h2].x, x1 = pk[p][2 * h2].y, y0 = pk[p][2 * h2 + 1].x, y1 = pk[p][2 * h2 + 1].y;
                    lane32_swap(x0, y0);
                    lane32_swap(x1, y1);
                    out[2 * ct + h2][p] = (ml3d_u32x4){x0, x1, y0, y1};
                }
        }
        if constexpr (L + 1 == S::CATL) {
            // the next layer's extra input columns, straight from their rows into fragment form (requested here: they land under
            // nothing -- 32 columns of a layer that has 64 + 32 -- but need no LDS either)
            const uint32_t m = m_row0 + (uint32_t)cl;
#pragma unroll
            for (int e = 0; e < S::CATC / 16; ++e) {
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (m < m_tot) {
                    const float4* src = reinterpret_cast<const float4*>(A.cat + (int64_t)m * S::CATC + 16 * e + 8 * hi);
                    const float4 a = src[0], b = src[1];
                    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                }
                b3_split8(v, out[S::n(L) / 16 + e][0], out[S::n(L) / 16 + e][1], out[S::n(L) / 16 + e][2]);
            }
        }
        chain_b3_layer<S, L + 1>(A, WT, BS, out, m_row0, m_tot, hi, cl);
    }
}

template <class S, int NW>
__global__ void __launch_bounds__(NW * 64) mlp_chain_b3(ChainArgs A) {
    using B = ChainB3<S>;
    static_assert(B::ok(), "mlp_chain_b3: widths must be multiples of 16 (hidden: 32)");
    HIP_DYNAMIC_SHARED(float, smem)
    uint16_t* WT = reinterpret_cast<uint16_t*>(smem);
    float* BS = reinterpret_cast<float*>(WT + B::w_total());
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, cl = lane & 31;
    // ---- weights -> W^T planes (four consecutive K of one output column per item), biases ---------------------------------------
#pragma unroll
    for (int l = 0; l < S::NL; ++l) {
        const ChainLayer Ly = A.L[l];
        const int np = S::np(l), cin = S::cin(l), wp = B::wp(l), pl = B::plane(l);
        uint16_t* W = WT + B::w_off(l);
        for (int e = tid; e < np * (cin / 4); e += NW * 64) {
            const int c = e % np, k4 = (e / np) * 4;
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
            if (c < Ly.cout) {
                v0 = Ly.wt[(k4 + 0) * Ly.cout + c]; v1 = Ly.wt[(k4 + 1) * Ly.cout + c];
                v2 = Ly.wt[(k4 + 2) * Ly.cout + c]; v3 = Ly.wt[(k4 + 3) * Ly.cout + c];
            }
            uint2 h, m, lo;
            b3_split4(v0, v1, v2, v3, h, m, lo);
            uint16_t* d = W + c * wp + k4;
            *reinterpret_cast<uint2*>(d) = h;
            *reinterpret_cast<uint2*>(d + pl) = m;
            *reinterpret_cast<uint2*>(d + 2 * pl) = lo;
        }
        for (int c = tid; c < np; c += NW * 64) {
            float b = 0.f;
            if (c < Ly.cout) { if (Ly.bias) b = Ly.bias[c]; if (Ly.bias2) b += Ly.bias2[c]; }
            BS[B::b_off(l) + c] = b;
        }
    }
    __syncthreads();
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t m_tot = (uint32_t)A.m_total;
    const uint32_t tiles = (m_tot + 31) / 32;
    const uint32_t t_step = gridDim.x * NW;
    constexpr int K0 = S::C0 + S::C1, KS0 = K0 / 16, KA = S::C0 / 16;
    const uint32_t rpi = (uint32_t)A.rows_per_item, srpi = (uint32_t)A.a1_rows_per_item;

    // the lane's input row of the NEXT tile is requested while the current one is computed (gathered rows: their index one step
    // earlier still)
    float4 pre[KS0][2];
    int gix = -1;
    auto request_idx = [&](uint32_t t) {
        if constexpr (S::C1 > 0) {
            const uint32_t m = t * 32 + (uint32_t)cl;
            gix = -1;
            if (m < m_tot) gix = A.gather ? A.gather[m] : (int)m;
        }
    };
    auto request_rows = [&](uint32_t t) {
        const uint32_t m0 = t * 32, m = m0 + (uint32_t)cl;
        uint32_t it0 = 0, l0 = m0;
        if (S::C1 > 0 && A.gather) { it0 = m0 / rpi; l0 = m0 - it0 * rpi; }
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) {
            pre[ks][0] = make_float4(0.f, 0.f, 0.f, 0.f);
            pre[ks][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_tot) {
                const float4* src;
                if (ks < KA) {
                    src = reinterpret_cast<const float4*>(A.a0 + (int64_t)m * S::C0 + 16 * ks + 8 * hi);
                } else {
                    int64_t grow = gix;
                    if (A.gather) grow += (int64_t)(it0 + ((l0 + (uint32_t)cl >= rpi) ? 1u : 0u)) * srpi;   // a tile crosses items at most once
                    src = reinterpret_cast<const float4*>(A.a1 + grow * S::C1 + 16 * (ks - KA) + 8 * hi);
                }
                pre[ks][0] = src[0];
                pre[ks][1] = src[1];
            }
        }
    };
    uint32_t t = blockIdx.x * NW + swave;
    if (t < tiles) { request_idx(t); request_rows(t); }
    if (t + t_step < tiles) request_idx(t + t_step);
    for (; t < tiles; t += t_step) {
        ml3d_u32x4 in[B::maxk() / 16][3];
#pragma unroll
        for (int ks = 0; ks < KS0; ++ks) {
            const float v[8] = {pre[ks][0].x, pre[ks][0].y, pre[ks][0].z, pre[ks][0].w, pre[ks][1].x, pre[ks][1].y, pre[ks][1].z, pre[ks][1].w};
            b3_split8(v, in[ks][0], in[ks][1], in[ks][2]);
        }
        if (t + t_step < tiles) {
            request_rows(t + t_step);
            if (t + 2 * t_step < tiles) request_idx(t + 2 * t_step);
        }
        chain_b3_layer<S, 0>(A, WT, BS, in, t * 32, m_tot, hi, cl);
    }
}

template <class S, int NW>
static int launch_mlp_chain_b3(const ChainArgs& a, hipStream_t st) {
    const size_t sm = ChainB3<S>::smem_bytes();
    if (sm > 48 * 1024 &&
        hipFuncSetAttribute((const void*)mlp_chain_b3<S, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess)
        return ML3D_E_LAUNCH;
    static const int cus = device_cu_count();
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)mlp_chain_b3<S, NW>, NW * 64, sm) != hipSuccess || occ < 1) occ = 1;
    const int64_t tiles = (a.m_total + 31) / 32;
    const int64_t want = (tiles + NW - 1) / NW, cap = (int64_t)occ * cus;
    hipLaunchKernelGGL((mlp_chain_b3<S, NW>), dim3((unsigned)(want < cap ? want : cap)), dim3(NW * 64), sm, st, a);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

// multi-layer chains run fused iff their shape has a compiled per-wave instance (callers fall back to one Linear per layer)
static bool chain_compiled(const ChainArgs& a) {
    if (a.m_total <= 0 || !knobs().mlp_shaped) return false;
    return mlp_shape_matches<ShapeDecFc1>(a) || mlp_shape_matches<ShapeFc1>(a) || mlp_shape_matches<ShapeEnc64>(a) ||
           mlp_shape_matches<ShapeEnc16>(a);
}

static int launch_chain_auto(const ChainArgs& a, hipStream_t st) {
    if constexpr ((ML3D_CHAIN_B3) != 0) {
        // (16-byte loads of 8 consecutive channels: rows must be 16-byte aligned, which every [m, C] buffer of the forward is)
        const bool al = (((uintptr_t)a.a0 | (uintptr_t)a.a1 | (uintptr_t)a.cat) & 15) == 0;
        if (al && mlp_shape_matches<ShapeDecFc1>(a)) return launch_mlp_chain_b3<ShapeDecFc1, 4>(a, st);
        if (al && mlp_shape_matches<ShapeFc1>(a)) return launch_mlp_chain_b3<ShapeFc1, 4>(a, st);
        if (al && mlp_shape_matches<ShapeEnc64>(a)) return launch_mlp_chain_b3<ShapeEnc64, 8>(a, st);
    }
    if (mlp_shape_matches<ShapeDecFc1>(a)) return launch_mlp_wave_s<ShapeDecFc1, 8>(a, st);
    if (mlp_shape_matches<ShapeFc1>(a)) return launch_mlp_wave_s<ShapeFc1, 8>(a, st);
    if (mlp_shape_matches<ShapeEnc64>(a)) return launch_mlp_wave_s<ShapeEnc64, 4>(a, st);
    if (mlp_shape_matches<ShapeEnc16>(a)) return launch_mlp_wave_s<ShapeEnc16, 8>(a, st);
    return ML3D_E_UNSUPPORTED;
}

// one Linear described by LinArgs: a shape-compiled per-wave kernel or the tile GEMM; the VALU kernel for K < 8
static int launch_linear_auto(const LinArgs& a, hipStream_t st);

template <int D, int STAGE>
static size_t lfa_smem_bytes(int d_in) {
    using C = LfaCfg<D>;
    size_t f = (size_t)C::TP * C::XSLAB * (STAGE == 2 ? 2 : 1) + (size_t)C::TP * RK * 12 + (size_t)C::TP * D;
    if (STAGE == 2) f += (size_t)C::TP * D + (size_t)C::TP * ((d_in + 3) & ~3);
    return f * 4 + (size_t)C::TP * RK * 4;
}

template <int D>
static int launch_lfa(const LfaArgs& a1, const LfaArgs& a2, hipStream_t st, const ml3d_trace* tr, int tag1) {
    using C = LfaCfg<D>;
    int64_t tiles = (a1.m_total + C::TP - 1) / C::TP;
    unsigned grid = (unsigned)(tiles < 8192 ? tiles : 8192);
    size_t sm1 = lfa_smem_bytes<D, 1>(a1.d_in), sm2 = lfa_smem_bytes<D, 2>(a2.d_in);
    if (sm1 > 160 * 1024 || sm2 > 160 * 1024) return ML3D_E_UNSUPPORTED;
    if (sm1 > 48 * 1024 &&
        hipFuncSetAttribute((const void*)lfa_stage<D, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm1) != hipSuccess)
        return ML3D_E_LAUNCH;
    if (sm2 > 48 * 1024 &&
        hipFuncSetAttribute((const void*)lfa_stage<D, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm2) != hipSuccess)
        return ML3D_E_LAUNCH;
    if (tr && tr->tag == tag1 && tr->ev_start) (void)hipEventRecord((hipEvent_t)tr->ev_start, st);
    hipLaunchKernelGGL((lfa_stage<D, 1>), dim3(grid), dim3(LFA_THREADS), sm1, st, a1);
    if (tr && tr->tag == tag1 && tr->ev_stop) (void)hipEventRecord((hipEvent_t)tr->ev_stop, st);
    if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    if (tr && tr->tag == tag1 + 1 && tr->ev_start) (void)hipEventRecord((hipEvent_t)tr->ev_start, st);
    hipLaunchKernelGGL((lfa_stage<D, 2>), dim3(grid), dim3(LFA_THREADS), sm2, st, a2);
    if (tr && tr->tag == tag1 + 1 && tr->ev_stop) (void)hipEventRecord((hipEvent_t)tr->ev_stop, st);
    if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// random_sample (randlanet.py:300-327): out[b, i, c] = max_k feat[b, idx[b, i, k], c], i < n_out
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gather_max(const float* __restrict__ feat, const int32_t* __restrict__ nidx, float* __restrict__ out,
           int64_t n_in, int64_t n_out, int64_t batch, int c) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= batch * n_out * c) return;
    int ch = (int)(e % c);
    int64_t r = e / c;
    int64_t b = r / n_out, i = r - b * n_out;
    const int32_t* id = nidx + (b * n_in + i) * RK;
    float v = -__builtin_inff();           // (not a finite seed: a row of -inf features pools to -inf, like torch.max)
#pragma unroll
    for (int k = 0; k < RK; ++k) v = fmaxf(v, feat[(b * n_in + id[k]) * c + ch]);
    out[r * c + ch] = v;
}

// the same for c % 4 == 0 (every shipped width): a thread owns 4 channels of one kept point (16-byte gathers), the
// cloud is blockIdx.y and all index arithmetic is 32-bit -- the per-element kernel above spends ~3 64-bit divisions
// (~400 VALU instructions) per output float.  `order` (optional) = the kept points in the spatial order of the next
// level's grid, see LfaArgs::order.
__global__ void __launch_bounds__(256)
gather_max4(const float* __restrict__ feat, const int32_t* __restrict__ nidx, float* __restrict__ out, uint32_t n_in,
            uint32_t n_out, uint32_t cv /* c / 4 */, uint32_t rows_per_block, const int32_t* __restrict__ order) {
    const uint32_t rl = threadIdx.x / cv, q = threadIdx.x - rl * cv;
    const uint32_t j = blockIdx.x * rows_per_block + rl;              // position in the walk over the cloud's kept points
    if (rl >= rows_per_block || j >= n_out) return;
    const uint32_t b = blockIdx.y;
    const uint32_t i = order ? (uint32_t)order[(size_t)b * n_out + j] - b * n_out : j;
    const int32_t* id = nidx + ((size_t)b * n_in + i) * RK;
    const float4* base = reinterpret_cast<const float4*>(feat + (size_t)b * n_in * (4 * cv)) + q;
    const float ninf = -__builtin_inff();  // (not a finite seed: a row of -inf features pools to -inf, like torch.max)
    float4 v = make_float4(ninf, ninf, ninf, ninf);
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const float4 x = base[(uint32_t)id[k] * cv];
        v.x = fmaxf(v.x, x.x); v.y = fmaxf(v.y, x.y); v.z = fmaxf(v.z, x.z); v.w = fmaxf(v.w, x.w);
    }
    reinterpret_cast<float4*>(out + ((size_t)b * n_out + i) * (4 * cv))[q] = v;
}

// adjoint of random_sample for training: the gradient of an output element goes to the neighbour that held
// the maximum -- the FIRST of equal maxima, like torch.max(dim) -- with one float atomic (several kept points may share it)
__global__ void __launch_bounds__(256)
gather_max_adjoint(const float* __restrict__ feat, const int32_t* __restrict__ nidx, const float* __restrict__ gout,
                   float* __restrict__ gfeat, int64_t n_in, int64_t n_out, int64_t batch, int c) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= batch * n_out * c) return;
    const int ch = (int)(e % c);
    const int64_t r = e / c;
    const int64_t b = r / n_out, i = r - b * n_out;
    const int32_t* id = nidx + (b * n_in + i) * RK;
    // the first listed neighbour is the incumbent: a row whose K values are all NaN (or below every finite seed) still sends
    // its gradient to one of ITS neighbours, like torch.max's backward, never to row 0 of batch item 0
    int64_t arg = b * n_in + id[0];
    float v = feat[arg * c + ch];
#pragma unroll
    for (int k = 1; k < RK; ++k) {
        const int64_t row = b * n_in + id[k];
        const float x = feat[row * c + ch];
        if (x > v) { v = x; arg = row; }
    }
    atomicAdd(gfeat + arg * c + ch, gout[r * c + ch]);
}

// attentive pooling as a differentiable op for training (randlanet.py:622-637 without the trailing SharedMLP):
//   out[r, c] = sum_k softmax_k(scores[r, :, c])[k] x[r, k, c]
// one thread per (row, channel) -- lanes run along the channels, so every access is a coalesced row segment -- the K values of
// both operands in registers, max-subtracted exponentials like torch.softmax.  Backward recomputes the softmax p from the saved
// inputs (nothing of size [rows, K, C] is kept between the passes): dx = p g, dscores = p (x - out) g.
constexpr int AP_KMAX = 32;
template <bool BWD>
__global__ void __launch_bounds__(256)
attentive_pool_k(const float* __restrict__ scores, const float* __restrict__ x, const float* __restrict__ out_saved,
                 const float* __restrict__ gout, int64_t rows, int k, int c, float* __restrict__ out,
                 float* __restrict__ gscores, float* __restrict__ gx) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * c) return;
    const int ch = (int)(e % c);
    const int64_t r = e / c;
    const float* sp = scores + r * k * c + ch;
    const float* xp = x + r * k * c + ch;
    float sv[AP_KMAX], xv[AP_KMAX];
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < AP_KMAX; ++j) {
        if (j < k) { sv[j] = sp[(int64_t)j * c]; xv[j] = xp[(int64_t)j * c]; mx = fmaxf(mx, sv[j]); }
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < AP_KMAX; ++j) {
        if (j < k) { sv[j] = expf(sv[j] - mx); sum += sv[j]; }
    }
    if constexpr (!BWD) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < AP_KMAX; ++j) if (j < k) acc = fmaf(sv[j] / sum, xv[j], acc);
        out[e] = acc;
    } else {
        const float g = gout[e], o = out_saved[e];
#pragma unroll
        for (int j = 0; j < AP_KMAX; ++j) {
            if (j < k) {
                const float pg = sv[j] / sum * g;
                gx[r * k * c + (int64_t)j * c + ch] = pg;
                gscores[r * k * c + (int64_t)j * c + ch] = pg * (xv[j] - o);
            }
        }
    }
}

struct Tracer {
    const ml3d_trace* t;
    hipStream_t st;
    void begin(int tag) const {
        for (const ml3d_trace* r = t; r; r = r->next)
            if (r->tag == tag && r->ev_start) (void)hipEventRecord((hipEvent_t)r->ev_start, st);
    }
    void end(int tag) const {
        for (const ml3d_trace* r = t; r; r = r->next)
            if (r->tag == tag && r->ev_stop) (void)hipEventRecord((hipEvent_t)r->ev_stop, st);
    }
};

static int launch_linear(const LinArgs& a, hipStream_t st) {
    if (a.m_total <= 0) return 0;
    int64_t items = ((a.m_total + LIN_RM - 1) / LIN_RM) * a.cout;
    hipLaunchKernelGGL(linear_act, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

static int launch_linear_auto(const LinArgs& a, hipStream_t st) {
    // the register-prefetching tile GEMM of gemm.hip (lin_mode 2 = the scalar kernel: a settled A/B, Knobs::linear stays 0)
    const int lin_mode = knobs().linear;
    const bool shaped = knobs().mlp_shaped;
    if (lin_mode == 0 && shaped && !a.a1 && a.m_total >= knobs().fuse_rows) {     // (tests lower the row threshold)
        // narrow Linears over many rows: the barrier-free per-wave kernel with a compiled shape
        ChainArgs c = {};
        c.a0 = a.a0; c.c0 = a.c0; c.n_layers = 1;
        c.L[0].wt = a.wt; c.L[0].bias = a.bias; c.L[0].bias2 = a.bias2; c.L[0].cin = a.c0; c.L[0].cout = a.cout;
        c.L[0].act = a.act; c.L[0].slope = a.slope;
        c.out = a.out; c.m_total = a.m_total;
        if (a.cout == 8 && mlp_shape_matches<ShapeLin16x8>(c)) return launch_mlp_wave_s<ShapeLin16x8, 8>(c, st);
        if (a.cout == 8 && mlp_shape_matches<ShapeLin8x8>(c)) return launch_mlp_wave_s<ShapeLin8x8, 8>(c, st);
        if (a.cout == 32 && mlp_shape_matches<ShapeLin64x32>(c)) return launch_mlp_wave_s<ShapeLin64x32, 8>(c, st);
        if (a.cout == 32 && mlp_shape_matches<ShapeLin32x32>(c)) return launch_mlp_wave_s<ShapeLin32x32, 8>(c, st);
        if (a.cout == 64 && mlp_shape_matches<ShapeLin32x64>(c)) return launch_mlp_wave_s<ShapeLin32x64, 8>(c, st);
    }
    if ((ML3D_LIN_B3_MINK) > 0 && a.pack_ws && !a.gather && a.c0 % 32 == 0 && (a.a1 ? a.c1 : 0) % 32 == 0 &&
        a.c0 + (a.a1 ? a.c1 : 0) >= (ML3D_LIN_B3_MINK) && a.cout % 4 == 0) {
        const int K = a.c0 + (a.a1 ? a.c1 : 0);
        const size_t need = gemm_pack_bf16x3_bytes(K, a.cout);
        if (need > 0 && need <= a.pack_bytes && gemm_pack_bf16x3(a.wt, K, a.cout, a.pack_ws, st) == 0) {
            Epilogue ep = {a.bias, nullptr, 0, a.act ? 1 : 0, a.slope, 0, 0, 0, 0, a.bias2};
            const int rc = gemm_rows_bf16x3(a.a0, a.c0, a.c0, a.a1, a.c1, a.a1 ? a.c1 : 0, a.m_total, a.pack_ws, a.cout, ep, a.out,
                                            a.cout, nullptr, 0, st);
            if (rc != ML3D_E_UNSUPPORTED) return rc;
        }
    }
    if (lin_mode != 2 && a.c0 + a.c1 >= 8) {
        RowsA A;
        A.a = a.a0; A.lda = a.c0; A.k1 = a.c0;
        A.gather = a.a1 ? a.gather : nullptr; A.gather_stride = 1; A.a_rows = a.a1_rows_per_item;
        A.a2 = a.a1; A.lda2 = a.c1; A.k2 = a.a1 ? a.c1 : 0;
        A.gather_on_a2 = 1; A.g_rows_per_item = a.rows_per_item; A.g_src_rows_per_item = a.a1_rows_per_item;
        Epilogue ep = {a.bias, nullptr, 0, a.act ? 1 : 0, a.slope, 0, 0, 0, 0, a.bias2};
        return gemm_rows(A, a.wt, a.m_total, a.cout, a.c0 + A.k2, ep, a.out, a.cout, nullptr, 0, st);
    }
    return launch_linear(a, st);
}

// ---- parameter layout ---------------------------------------------------------------------------
struct Layout {
    int n_slots;
    int64_t off[2 + 18 * ML3D_RANDLA_MAX_LAYERS + 2 + 2 * ML3D_RANDLA_MAX_LAYERS + 6 + 1];
};

static bool desc_ok(const ml3d_randla_desc* d) {
    if (!d || d->num_layers < 1 || d->num_layers > ML3D_RANDLA_MAX_LAYERS) return false;
    if (d->in_channels < 1 || d->dim_features < 1 || d->num_classes < 1 || d->batch < 1 || d->num_points < 1)
        return false;
    for (int l = 0; l < d->num_layers; ++l)
        if (d->dim_output[l] < 2 || (d->dim_output[l] & 1) || d->sub_sampling_ratio[l] < 1) return false;
    return true;
}

static void enc_dims(const ml3d_randla_desc* d, int* ed /* L+1 */) {
    // encoder_dim_list of randlanet.py:81-91
    int n = 0;
    for (int l = 0; l < d->num_layers; ++l) {
        if (l == 0) ed[n++] = 2 * d->dim_output[0];
        ed[n++] = 2 * d->dim_output[l];
    }
}

static void make_layout(const ml3d_randla_desc* d, Layout* L) {
    int s = 0;
    int64_t o = 0;
    auto push = [&](int64_t count) { L->off[s++] = o; o += (count + 3) & ~(int64_t)3; };
    push((int64_t)d->in_channels * d->dim_features);
    push(d->dim_features);
    int d_in = d->dim_features;
    for (int l = 0; l < d->num_layers; ++l) {
        int64_t dd = d->dim_output[l], h = dd / 2;
        push(d_in * h); push(h);
        push(10 * h); push(h);
        push(dd * dd); push(dd);
        push(dd * h); push(h);
        push(h * h); push(h);
        push(dd * dd); push(dd);
        push(dd * dd); push(dd);
        push(dd * 2 * dd); push((int64_t)d_in * 2 * dd);   // mlp2_WT | shortcut_WT: one stacked [d + d_in][2d] matrix
        push(2 * dd); push(2 * dd);
        d_in = (int)(2 * dd);
    }
    int64_t Dm = d_in;
    push(Dm * Dm); push(Dm);
    int ed[ML3D_RANDLA_MAX_LAYERS + 1];
    enc_dims(d, ed);
    int64_t prev = Dm;
    for (int i = 0; i < d->num_layers; ++i) {
        int64_t skip = ed[d->num_layers + 1 - i - 2];
        push((skip + prev) * skip); push(skip);
        prev = skip;
    }
    push(prev * 64); push(64);
    push(64 * 32); push(32);
    push(32 * (int64_t)d->num_classes); push(d->num_classes);
    L->n_slots = s;
    L->off[s] = o;
}

// scratch for the bf16 planes of ONE weight matrix at a time (the pack kernel and the GEMM that reads it are stream-ordered; the
// next Linear's pack overwrites it after that GEMM): the largest [K, N] of pool2 / mlp2 | shortcut / the final mlp
static size_t lin_pack_bytes(const ml3d_randla_desc* d) {
    size_t m = 0;
    int d_in = d->dim_features;
    for (int l = 0; l < d->num_layers; ++l) {
        const int dd = d->dim_output[l];
        const size_t a = gemm_pack_bf16x3_bytes(dd, dd), b = gemm_pack_bf16x3_bytes(dd + d_in, 2 * dd), c = gemm_pack_bf16x3_bytes(dd, dd / 2);
        m = a > m ? a : m; m = b > m ? b : m; m = c > m ? c : m;
        d_in = 2 * dd;
    }
    const size_t e = gemm_pack_bf16x3_bytes(d_in, d_in);
    return (e > m ? e : m) + 256;
}

static size_t fwd_ws_floats(const ml3d_randla_desc* d) {
    int64_t n[ML3D_RANDLA_MAX_LAYERS + 1];
    n[0] = d->num_points;
    for (int l = 0; l < d->num_layers; ++l) n[l + 1] = n[l] / d->sub_sampling_ratio[l];
    auto al = [](int64_t x) { return (x + 63) & ~(int64_t)63; };
    int64_t B = d->batch, f = 0;
    f += al(B * n[0] * d->dim_features);
    for (int l = 0; l < d->num_layers; ++l) {
        int64_t dd = d->dim_output[l], h = dd / 2;
        f += 2 * al(B * n[l] * h) + al(B * n[l] * 2 * dd) + al(B * n[l + 1] * 2 * dd) + 2 * al(B * n[l] * dd);
    }
    int64_t Dm = 2 * d->dim_output[d->num_layers - 1];
    f += al(B * n[d->num_layers] * Dm);
    int ed[ML3D_RANDLA_MAX_LAYERS + 1];
    enc_dims(d, ed);
    for (int i = 0; i < d->num_layers; ++i) {
        const int lev = d->num_layers - 1 - i, skip_c = ed[d->num_layers + 1 - i - 2];
        f += al(B * n[lev] * skip_c) + al(B * n[lev + 1] * skip_c);    // stage output + per-coarse-point half (decoder split)
    }
    f += al(B * n[0] * 64) + al(B * n[0] * 32);
    f += al((int64_t)(lin_pack_bytes(d) + 3) / 4) + 64;              // three bf16 planes of the largest Linear's weights
    return (size_t)f;
}

}  // namespace ml3d

using namespace ml3d;

extern "C" int ml3d_randla_param_layout(const ml3d_randla_desc* desc, int64_t* offsets_out, int max_slots) {
    if (!desc_ok(desc) || !offsets_out) return ML3D_E_INVALID;
    Layout L;
    make_layout(desc, &L);
    if (max_slots < L.n_slots + 1) return ML3D_E_INVALID;
    for (int i = 0; i <= L.n_slots; ++i) offsets_out[i] = L.off[i];
    return L.n_slots;
}

extern "C" size_t ml3d_randla_forward_workspace_bytes(const ml3d_randla_desc* desc) {
    if (!desc_ok(desc)) return 0;
    return fwd_ws_floats(desc) * 4 + 256;
}

extern "C" int ml3d_randla_forward(const ml3d_randla_desc* d, const float* params, const float* features,
                                   const float* points, const int32_t* const* neighbor_idx,
                                   const int32_t* const* interp_idx, float* out_scores, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    return ml3d_randla_forward_traced(d, params, features, points, neighbor_idx, interp_idx, out_scores, workspace,
                                      workspace_bytes, stream, nullptr);
}

extern "C" int ml3d_randla_forward_traced(const ml3d_randla_desc* d, const float* params, const float* features,
                                          const float* points, const int32_t* const* neighbor_idx,
                                          const int32_t* const* interp_idx, float* out_scores, void* workspace,
                                          size_t workspace_bytes, void* stream, const ml3d_trace* trace) {
    return ml3d_randla_forward_ordered(d, params, features, points, neighbor_idx, interp_idx, nullptr, out_scores,
                                       workspace, workspace_bytes, stream, trace);
}

extern "C" int ml3d_randla_forward_ordered(const ml3d_randla_desc* d, const float* params, const float* features,
                                           const float* points, const int32_t* const* neighbor_idx,
                                           const int32_t* const* interp_idx, const int32_t* const* tile_order,
                                           float* out_scores, void* workspace, size_t workspace_bytes, void* stream,
                                           const ml3d_trace* trace) {
    if (!desc_ok(d) || !params || !features || !points || !neighbor_idx || !interp_idx || !out_scores)
        return ML3D_E_INVALID;
    if (d->num_neighbors != RK) return ML3D_E_UNSUPPORTED;
    if (workspace_bytes < ml3d_randla_forward_workspace_bytes(d)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const Tracer T = {trace, st};
    const bool force_valu = knobs().force_valu;                // (settled A/B switches of rounds 1-3, fixed false: the generic VALU
    const bool no_fuse = knobs().no_fuse;                      //  kernels / one launch per Linear)
    const int64_t fuse_rows = knobs().fuse_rows;               // tuning/test knob: rows from which pool2+mlp2 fuse
    const int Lr = d->num_layers;
    const int64_t B = d->batch;
    int64_t n[ML3D_RANDLA_MAX_LAYERS + 1];
    n[0] = d->num_points;
    for (int l = 0; l < Lr; ++l) n[l + 1] = n[l] / d->sub_sampling_ratio[l];
    if (n[Lr] < 1) return ML3D_E_INVALID;
    if (n[Lr - 1] < RK) return ML3D_E_UNSUPPORTED;  // every level needs >= 16 points for a full neighbour row
    Layout L;
    make_layout(d, &L);
    auto P = [&](int slot) { return params + L.off[slot]; };
    float* wsf = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    auto take = [&](int64_t count) { float* p = wsf; wsf += (count + 63) & ~(int64_t)63; return p; };

    // scratch for the bf16 planes of one weight matrix (deep Linears on the bf16 matrix pipe: launch_linear_auto)
    const size_t pack_bytes = lin_pack_bytes(d);
    void* pack_ws = take((int64_t)(pack_bytes + 3) / 4);
    // fc0 + bn0 + lrelu(0.2)            (randlanet.py:266-271)
    float* feat = take(B * n[0] * d->dim_features);
    // every reference config: 8 features into a 16-wide first layer -> fc0 and that layer's mlp1 share one launch
    float* f1_head = nullptr;
    if (!force_valu && !no_fuse && d->dim_features == 8 && d->dim_output[0] == 16) {
        f1_head = take(B * n[0] * 8);
        const int64_t m = B * n[0];
        T.begin(1000);
        hipLaunchKernelGGL(head_fc0_mlp1, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, features, d->in_channels, P(0), P(1),
                           P(2), P(3), m, feat, f1_head);
        T.end(1000);
        if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    } else {
        LinArgs a = {}; a.pack_ws = pack_ws; a.pack_bytes = pack_bytes;
        a.a0 = features; a.c0 = d->in_channels; a.wt = P(0); a.bias = P(1); a.out = feat;
        a.m_total = B * n[0]; a.cout = d->dim_features; a.act = 1; a.slope = 0.2f;
        T.begin(1000); int rc = launch_linear_auto(a, st); T.end(1000); if (rc) return rc;
    }
    int d_in = d->dim_features;
    float* enc_keep[ML3D_RANDLA_MAX_LAYERS + 1];   // encoder_feat_list (randlanet.py:274-283)
    for (int l = 0; l < Lr; ++l) {
        const int dd = d->dim_output[l], h = dd / 2, sb = 2 + 18 * l;
        const int64_t M = B * n[l];
        float* f1 = (l == 0 && f1_head) ? f1_head : take(M * h);
        float* p1 = take(M * h);
        float* enc = take(M * 2 * dd);
        float* samp = take(B * n[l + 1] * 2 * dd);
        if (!(l == 0 && f1_head)) {   // mlp1: SharedMLP(d_in, d/2) lrelu 0.2   (randlanet.py:680)
            LinArgs a = {}; a.pack_ws = pack_ws; a.pack_bytes = pack_bytes;
            a.a0 = feat; a.c0 = d_in; a.wt = P(sb + 0); a.bias = P(sb + 1); a.out = f1;
            a.m_total = M; a.cout = h; a.act = 1; a.slope = 0.2f;
            T.begin(8 * l); int rc = launch_linear_auto(a, st); T.end(8 * l); if (rc) return rc;
        }
        LfaArgs s1 = {};
        s1.xyz = points; s1.nidx = neighbor_idx[l]; s1.n = n[l]; s1.n0 = n[0]; s1.m_total = M;
        s1.gfeat = f1; s1.lse1_wt = P(sb + 2); s1.lse1_b = P(sb + 3);
        s1.score_wt = P(sb + 4); s1.score_b = P(sb + 5); s1.pool_wt = P(sb + 6); s1.pool_b = P(sb + 7);
        s1.d_in = d_in; s1.out = p1;
        s1.order = tile_order ? tile_order[l] : nullptr;
        LfaArgs s2 = s1;
        s2.gfeat = p1; s2.lse2_wt = P(sb + 8); s2.lse2_b = P(sb + 9);
        s2.score_wt = P(sb + 10); s2.score_b = P(sb + 11); s2.pool_wt = P(sb + 12); s2.pool_b = P(sb + 13);
        s2.mlp2_wt = P(sb + 14); s2.mlp2_b = P(sb + 16); s2.short_wt = P(sb + 15); s2.short_b = P(sb + 17);
        s2.feat_in = feat; s2.out = enc;
        int rc = 0;
        // (the MFMA kernels keep point indices in 32 bits; a > 2^30-point batch level takes the generic VALU kernel)
        const bool mfma_ok = !force_valu && (dd == 16 ? attn_mfma_fits<16>(s1) : dd == 32 ? attn_mfma_fits<32>(s1) :
                                             dd == 64 ? attn_mfma_fits<64>(s1) : dd == 128 ? attn_mfma_fits<128>(s1) :
                                             dd == 256 ? attn_mfma_fits<256>(s1) : false);
        if (mfma_ok) {
            float* agg = take(M * dd);
            float* p2 = take(M * dd);
            LfaArgs q1 = s1; q1.out = agg;
            // first layer of every reference config (16 wide, 8 features in): the per-point Linears behind both stages run
            // inside the attention kernels (lfa_attn_mfma16<.., EPI>), `agg` is never written
            const bool epi16 = dd == 16 && d_in == 8 && !no_fuse;
            if (epi16) q1.out = p1;
            // D >= 128: the feature half of the score Linear once per POINT (gscore = f . W_top^T, into the p2 scratch,
            // which is otherwise idle until pool2), gathered by the attention kernel instead of recomputed per neighbour
            const bool split_on = knobs().attn_split;
            const bool split = split_on && dd >= 32 && dd <= 256 && M * dd * 4 < ((int64_t)1 << 32) &&
                               M < ((int64_t)1 << 30);
            // (the per-wave kernels of D <= 64 and the bf16x3 kernels of D = 128 / 256 take the score bias inside gscore, the f32
            //  workgroup kernels add it themselves)
            const bool b3_attn = ((ML3D_ATTN_B3) & 1) != 0 && split && ((dd == 128 && n[l] >= B3Cfg<128, 1>::TP) || (dd == 256 && n[l] >= B3Cfg<256, 1>::TP));
            auto point_scores = [&](const float* gfeat, const float* score_wt, const float* score_b, int tag) -> int {
                LinArgs ga = {}; ga.pack_ws = pack_ws; ga.pack_bytes = pack_bytes;
                ga.a0 = gfeat; ga.c0 = h; ga.wt = score_wt; ga.bias = (dd <= 64 || b3_attn) ? score_b : nullptr; ga.out = p2;   // first h rows of [d][d]
                ga.m_total = M; ga.cout = dd; ga.act = 0;
                T.begin(tag);
                const int r = launch_linear_auto(ga, st);
                T.end(tag);
                return r;
            };
            if (split) { rc = point_scores(f1, s1.score_wt, s1.score_b, 8 * l + 7); if (rc) return rc; q1.gscore = p2; }
            T.begin(8 * l + 1);
            switch (dd) {
                case 16: rc = launch_attn_mfma16<1>(q1, st, epi16); break;
                case 32: rc = launch_attn_mfma<32, 1>(q1, st); break;
                case 64: rc = launch_attn_mfma<64, 1>(q1, st); break;
                case 128: rc = launch_attn_mfma<128, 1>(q1, st); break;
                default: rc = launch_attn_mfma<256, 1>(q1, st); break;
            }
            T.end(8 * l + 1);
            if (rc) return rc;
            if (!epi16) {   // pool1.mlp: SharedMLP(d, d/2) lrelu 0.2            (randlanet.py:639)
                LinArgs a = {}; a.pack_ws = pack_ws; a.pack_bytes = pack_bytes;
                a.a0 = agg; a.c0 = dd; a.wt = P(sb + 6); a.bias = P(sb + 7); a.out = p1;
                a.m_total = M; a.cout = h; a.act = 1; a.slope = 0.2f;
                T.begin(8 * l + 4); rc = launch_linear_auto(a, st); T.end(8 * l + 4); if (rc) return rc;
            }
            LfaArgs q2 = s2; q2.out = epi16 ? enc : agg;
            if (split) { rc = point_scores(p1, s2.score_wt, s2.score_b, 8 * l + 7); if (rc) return rc; q2.gscore = p2; }
            T.begin(8 * l + 2);
            switch (dd) {
                case 16: rc = launch_attn_mfma16<2>(q2, st, epi16); break;
                case 32: rc = launch_attn_mfma<32, 2>(q2, st); break;
                case 64: rc = launch_attn_mfma<64, 2>(q2, st); break;
                case 128: rc = launch_attn_mfma<128, 2>(q2, st); break;
                default: rc = launch_attn_mfma<256, 2>(q2, st); break;
            }
            T.end(8 * l + 2);
            if (rc) return rc;
            // pool2.mlp: SharedMLP(d, d) lrelu 0.2, then lrelu_0.01(mlp2(p2) + shortcut(feat)) as ONE linear over
            // [p2 | feat] (randlanet.py:639, 692).  With enough rows to fill the chip the two run back to back in
            // one launch (p2 stays in LDS); small levels keep two launches so the column passes spread over CUs.
            ChainArgs ch = {};
            ch.a0 = agg; ch.c0 = dd; ch.n_layers = 2;
            ch.L[0].wt = P(sb + 12); ch.L[0].bias = P(sb + 13); ch.L[0].cin = dd; ch.L[0].cout = dd;
            ch.L[0].act = 1; ch.L[0].slope = 0.2f;
            ch.cat = feat; ch.cat_c = d_in; ch.cat_layer = 1;
            ch.L[1].wt = P(sb + 14); ch.L[1].bias = P(sb + 16); ch.L[1].bias2 = P(sb + 17);
            ch.L[1].cin = dd + d_in; ch.L[1].cout = 2 * dd; ch.L[1].act = 1; ch.L[1].slope = 0.01f;
            ch.out = enc; ch.m_total = M;
            // (fused when the shape has a compiled per-wave instance: the first two encoder layers of every reference
            //  config; wide layers run faster as two tile GEMMs than through any chain kernel: 0.69 -> 0.35 ms at 128 channels)
            if (epi16) {
                // (done inside stage 2)
            } else if (!no_fuse && M >= fuse_rows && chain_compiled(ch)) {
                T.begin(8 * l + 5); rc = launch_chain_auto(ch, st); T.end(8 * l + 5); if (rc) return rc;
            } else {
                {
                    LinArgs a = {}; a.pack_ws = pack_ws; a.pack_bytes = pack_bytes;
                    a.a0 = agg; a.c0 = dd; a.wt = P(sb + 12); a.bias = P(sb + 13); a.out = p2;
                    a.m_total = M; a.cout = dd; a.act = 1; a.slope = 0.2f;
                    T.begin(8 * l + 5); rc = launch_linear_auto(a, st); T.end(8 * l + 5); if (rc) return rc;
                }
                {
                    LinArgs a = {}; a.pack_ws = pack_ws; a.pack_bytes = pack_bytes;
                    a.a0 = p2; a.c0 = dd; a.a1 = feat; a.c1 = d_in; a.wt = P(sb + 14);
                    a.bias = P(sb + 16); a.bias2 = P(sb + 17); a.out = enc;
                    a.m_total = M; a.cout = 2 * dd; a.act = 1; a.slope = 0.01f;
                    T.begin(8 * l + 6); rc = launch_linear_auto(a, st); T.end(8 * l + 6); if (rc) return rc;
                }
            }
        } else {
            switch (dd) {
                case 8: rc = launch_lfa<8>(s1, s2, st, trace, 8 * l + 1); break;
                case 16: rc = launch_lfa<16>(s1, s2, st, trace, 8 * l + 1); break;
                case 32: rc = launch_lfa<32>(s1, s2, st, trace, 8 * l + 1); break;
                case 64: rc = launch_lfa<64>(s1, s2, st, trace, 8 * l + 1); break;
                case 128: rc = launch_lfa<128>(s1, s2, st, trace, 8 * l + 1); break;
                case 256: rc = launch_lfa<256>(s1, s2, st, trace, 8 * l + 1); break;
                case 512: rc = launch_lfa<512>(s1, s2, st, trace, 8 * l + 1); break;
                default: return ML3D_E_UNSUPPORTED;
            }
        }
        if (rc) return rc;
        {   // random_sample onto the kept prefix      (randlanet.py:278)
            int64_t items = B * n[l + 1] * 2 * dd;
            T.begin(8 * l + 3);
            const int c2 = 2 * dd;
            if ((c2 & 3) == 0 && c2 / 4 <= 256 && n[l] * c2 < ((int64_t)1 << 32) && B < 65536 && !force_valu) {
                const uint32_t cv = (uint32_t)(c2 / 4), rpb = 256u / cv;
                // (the kept points of level l are level l+1: its grid's order is their spatial order)
                const int32_t* ord = (tile_order && l + 1 < Lr) ? tile_order[l + 1] : nullptr;
                hipLaunchKernelGGL(gather_max4, dim3((unsigned)((n[l + 1] + rpb - 1) / rpb), (unsigned)B), dim3(256), 0, st, enc,
                                   neighbor_idx[l], samp, (uint32_t)n[l], (uint32_t)n[l + 1], cv, rpb, ord);
            } else {
                hipLaunchKernelGGL(gather_max, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, enc,
                                   neighbor_idx[l], samp, n[l], n[l + 1], B, 2 * dd);
            }
            T.end(8 * l + 3);
            if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
        }
        if (l == 0) enc_keep[0] = enc;
        enc_keep[l + 1] = samp;
        feat = samp;
        d_in = 2 * dd;
    }
    int slot = 2 + 18 * Lr;
    const int Dm = d_in;
    float* cur = take(B * n[Lr] * Dm);
    {   // mlp: SharedMLP(D, D) lrelu 0.2               (randlanet.py:285)
        LinArgs a = {}; a.pack_ws = pack_ws; a.pack_bytes = pack_bytes;
        a.a0 = feat; a.c0 = Dm; a.wt = P(slot); a.bias = P(slot + 1); a.out = cur;
        a.m_total = B * n[Lr]; a.cout = Dm; a.act = 1; a.slope = 0.2f;
        T.begin(1001); int rc = launch_linear_auto(a, st); T.end(1001); if (rc) return rc;
        slot += 2;
    }
    int ed[ML3D_RANDLA_MAX_LAYERS + 1];
    enc_dims(d, ed);
    int cprev = Dm;
    for (int i = 0; i < Lr; ++i) {
        // nearest_interpolation + cat + ConvTranspose2d 1x1 + BN + lrelu 0.2   (randlanet.py:288-293)
        const int lev = Lr - 1 - i;                 // output level
        const int skip_c = ed[Lr + 1 - i - 2];
        float* outp = take(B * n[lev] * skip_c);
        LinArgs a = {}; a.pack_ws = pack_ws; a.pack_bytes = pack_bytes;
        a.a0 = enc_keep[Lr + 1 - i - 2]; a.c0 = skip_c;
        a.a1 = cur; a.c1 = cprev; a.gather = interp_idx[lev];
        a.rows_per_item = n[lev]; a.a1_rows_per_item = n[lev + 1];
        a.wt = P(slot); a.bias = P(slot + 1); a.out = outp;
        a.m_total = B * n[lev]; a.cout = skip_c; a.act = 1; a.slope = 0.2f;
        if (i == Lr - 1 && !no_fuse && !force_valu) {
            // the last decoder stage feeds only fc1: run [skip | interp] -> decoder -> fc1.0 -> fc1.1 -> fc1.3 as ONE
            // per-wave chain when the shape has a compiled instance (the N0 x 32 decoder output never exists in HBM)
            ChainArgs ch = {};
            ch.a0 = a.a0; ch.c0 = a.c0; ch.a1 = a.a1; ch.c1 = a.c1; ch.gather = a.gather;
            ch.rows_per_item = a.rows_per_item; ch.a1_rows_per_item = a.a1_rows_per_item;
            ch.n_layers = 4;
            ch.L[0].wt = a.wt; ch.L[0].bias = a.bias; ch.L[0].cin = a.c0 + a.c1; ch.L[0].cout = skip_c; ch.L[0].act = 1; ch.L[0].slope = 0.2f;
            ch.L[1].wt = P(slot + 2); ch.L[1].bias = P(slot + 3); ch.L[1].cin = skip_c; ch.L[1].cout = 64; ch.L[1].act = 1; ch.L[1].slope = 0.2f;
            ch.L[2].wt = P(slot + 4); ch.L[2].bias = P(slot + 5); ch.L[2].cin = 64; ch.L[2].cout = 32; ch.L[2].act = 1; ch.L[2].slope = 0.2f;
            ch.L[3].wt = P(slot + 6); ch.L[3].bias = P(slot + 7); ch.L[3].cin = 32; ch.L[3].cout = d->num_classes; ch.L[3].act = 0;
            ch.out = out_scores; ch.m_total = a.m_total;
            const bool dec_fuse = knobs().dec_fc1;
            if (dec_fuse && knobs().mlp_shaped && mlp_shape_matches<ShapeDecFc1>(ch)) {
                T.begin(1200); int rc = launch_chain_auto(ch, st); T.end(1200);
                return rc;
            }
        }
        // W . [skip ; up[idx]] = W_skip . skip + (W_up . up)[idx]: the upsampled half is linear in a per-COARSE-point
        // quantity, computed once per coarse point (1/ratio of the rows) and added back through the interpolation index
        // as a gathered residual of the skip GEMM
        const bool dec_split = knobs().dec_split;
        if (dec_split && !force_valu && n[lev] >= 64 && (skip_c & 3) == 0 && (cprev & 3) == 0 && skip_c >= 8) {
            float* up = take(B * n[lev + 1] * skip_c);
            RowsA Au = {};
            Au.a = cur; Au.lda = cprev; Au.k1 = cprev;
            Epilogue e0 = {};
            T.begin(1100 + i);
            int rc = gemm_rows(Au, a.wt + (int64_t)skip_c * skip_c, B * n[lev + 1], skip_c, cprev, e0, up, skip_c, nullptr, 0, st);
            if (rc) return rc;
            RowsA As = {};
            As.a = a.a0; As.lda = skip_c; As.k1 = skip_c;
            Epilogue e1 = {a.bias, up, skip_c, 1, 0.2f, 0, 0, 0, 0, nullptr, a.gather, n[lev], n[lev + 1], 0, n[lev + 1]};
            rc = gemm_rows(As, a.wt, a.m_total, skip_c, skip_c, e1, outp, skip_c, nullptr, 0, st);
            T.end(1100 + i);
            if (rc) return rc;
        } else {
            T.begin(1100 + i); int rc = launch_linear_auto(a, st); T.end(1100 + i); if (rc) return rc;
        }
        slot += 2;
        cur = outp;
        cprev = skip_c;
    }
    float* t0 = take(B * n[0] * 64);
    float* t1 = take(B * n[0] * 32);
    {   // fc1                                            (randlanet.py:93-96, 296)
        LinArgs a = {}; a.pack_ws = pack_ws; a.pack_bytes = pack_bytes;
        a.a0 = cur; a.c0 = cprev; a.wt = P(slot); a.bias = P(slot + 1); a.out = t0;
        a.m_total = B * n[0]; a.cout = 64; a.act = 1; a.slope = 0.2f;
        ChainArgs ch = {};
        ch.a0 = cur; ch.c0 = cprev; ch.n_layers = 3;
        ch.L[0].wt = P(slot); ch.L[0].bias = P(slot + 1); ch.L[0].cin = cprev; ch.L[0].cout = 64; ch.L[0].act = 1; ch.L[0].slope = 0.2f;
        ch.L[1].wt = P(slot + 2); ch.L[1].bias = P(slot + 3); ch.L[1].cin = 64; ch.L[1].cout = 32; ch.L[1].act = 1; ch.L[1].slope = 0.2f;
        ch.L[2].wt = P(slot + 4); ch.L[2].bias = P(slot + 5); ch.L[2].cin = 32; ch.L[2].cout = d->num_classes; ch.L[2].act = 0;
        ch.out = out_scores; ch.m_total = B * n[0];
        if (!no_fuse && !force_valu && chain_compiled(ch)) {
            // fc1.0 -> fc1.1 -> fc1.3 back to back: the 64- and 32-wide activations never leave LDS
            T.begin(1200); int rc = launch_chain_auto(ch, st); T.end(1200); if (rc) return rc;
        } else {
            T.begin(1200); int rc = launch_linear_auto(a, st); T.end(1200); if (rc) return rc;
            a.a0 = t0; a.c0 = 64; a.wt = P(slot + 2); a.bias = P(slot + 3); a.out = t1; a.cout = 32;
            T.begin(1201); rc = launch_linear_auto(a, st); T.end(1201); if (rc) return rc;
            a.a0 = t1; a.c0 = 32; a.wt = P(slot + 4); a.bias = P(slot + 5); a.out = out_scores;
            a.cout = d->num_classes; a.act = 0;
            T.begin(1202); rc = launch_linear_auto(a, st); T.end(1202); if (rc) return rc;
        }
    }
    return 0;
}

// ---- random_sample as a stand-alone differentiable op (training side) ------------------------------------
extern "C" int ml3d_randla_gather_max(const float* features, const int32_t* pool_idx, int64_t batch, int64_t n_in, int64_t n_out,
                                      int channels, float* out, void* stream) {
    if (batch < 0 || n_in < 0 || n_out < 0 || n_out > n_in || channels <= 0) return ML3D_E_INVALID;
    if (batch * n_out == 0) return 0;
    if (!features || !pool_idx || !out) return ML3D_E_INVALID;
    const int64_t total = batch * n_out * channels;
    hipLaunchKernelGGL(gather_max, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, features, pool_idx, out,
                       n_in, n_out, batch, channels);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_randla_gather_max_backward(const float* features, const int32_t* pool_idx, const float* grad_out, int64_t batch,
                                               int64_t n_in, int64_t n_out, int channels, float* grad_features, void* stream) {
    if (batch < 0 || n_in < 0 || n_out < 0 || n_out > n_in || channels <= 0) return ML3D_E_INVALID;
    if (batch * n_in == 0) return 0;
    if (!grad_features) return ML3D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_features, 0, sizeof(float) * (size_t)(batch * n_in) * (size_t)channels, st) != hipSuccess) return ML3D_E_LAUNCH;
    if (n_out == 0) return 0;
    if (!features || !pool_idx || !grad_out) return ML3D_E_INVALID;
    const int64_t total = batch * n_out * channels;
    hipLaunchKernelGGL(gather_max_adjoint, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, features, pool_idx, grad_out,
                       grad_features, n_in, n_out, batch, channels);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

// ---- attentive pooling as a stand-alone differentiable op (training side) ---------------------------------
extern "C" int ml3d_randla_attentive_pool(const float* scores, const float* x, int64_t rows, int k, int channels, float* out,
                                          void* stream) {
    if (rows < 0 || k <= 0 || channels <= 0) return ML3D_E_INVALID;
    if (k > AP_KMAX) return ML3D_E_UNSUPPORTED;
    if (rows == 0) return 0;
    if (!scores || !x || !out) return ML3D_E_INVALID;
    const int64_t total = rows * channels;
    hipLaunchKernelGGL((attentive_pool_k<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, x,
                       nullptr, nullptr, rows, k, channels, out, nullptr, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_randla_attentive_pool_backward(const float* scores, const float* x, const float* out, const float* grad_out,
                                                   int64_t rows, int k, int channels, float* grad_scores, float* grad_x,
                                                   void* stream) {
    if (rows < 0 || k <= 0 || channels <= 0) return ML3D_E_INVALID;
    if (k > AP_KMAX) return ML3D_E_UNSUPPORTED;
    if (rows == 0) return 0;
    if (!scores || !x || !out || !grad_out || !grad_scores || !grad_x) return ML3D_E_INVALID;
    const int64_t total = rows * channels;
    hipLaunchKernelGGL((attentive_pool_k<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, scores, x,
                       out, grad_out, rows, k, channels, nullptr, grad_scores, grad_x);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}
