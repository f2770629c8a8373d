// kpconv.hip — KPConv (rigid) inference blocks for gfx950.
//
// Replaces the PyTorch op chains of
//   KPConv.forward, non-deformable branch        ml3d/torch/models/kpconv.py:1048-1068, 1105-1118, 1139-1159
//   UnaryBlock / BatchNormBlock (eval)            kpconv.py:1213-1300
//   max_pool / closest_pool                       kpconv.py:821-858
//   NearestUpsampleBlock + torch.cat + UnaryBlock kpconv.py:283-286, 1468-1481  (one fused GEMM)
// with BatchNorm folded into the weights by the host.
//
// KPConv = two kernels:
//  (1) kp_weighted — "gather" half.  For each query point the influence of its H neighbours on the 15
//      kernel points, w[k][h] = max(0, 1 - |s_h - q - kp_k| / extent), is computed once per (query,
//      neighbour) pair into an LDS tile shared by the lanes that own the query's channels; the lanes
//      then stream the neighbours' feature rows (contiguous [Cin] bursts, L2-resident) and accumulate
//      wf[k][c] = sum_h w[k][h] * x[idx_h][c] in registers (15 x Cin/G accumulators per lane).  The
//      reference materialises [N,H,15,3] differences, [N,15,H] weights and an [N,H,Cin] gather in HBM
//      (47% of its forward is aten::gather); here only wf [N, 15*Cin] is written.
//  (2) the [Nq, 15*Cin] x [15*Cin, Cout] contraction with the kernel weights = one f32 MFMA GEMM
//      (gemm.hip) with bias (folded BN) + LeakyReLU in the epilogue, split along K for the deep layers.
// Roofline: f32 vector/matrix peak 157.3 TFLOP/s; algorithmic flops per block
//   Nq * (2*15*H*Cin + 2*15*Cin*Cout)   (SURVEY.md §8d).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gemm.h"
#include "grid.h"
#include <gfx950_ops.h>
#include "ml3d_hip.h"

namespace ml3d {

constexpr int KP_K = 15;          // kernel points (every reference config)
constexpr int KP_WP = 16;         // LDS pitch of a neighbour's 15 weights
constexpr int KP_HC = 32;         // neighbours per LDS chunk

struct KpArgs {
    const float* q_pts; const float* s_pts;
    const int32_t* inds; int64_t nq, ns; int h;
    const float* x; int cin;
    const float* kp;              // [15, 3]
    float inv_extent; int influence;   // 0 constant, 1 linear, 2 gaussian (sigma = 0.3 * extent)
    float gauss_den;              // 2 * sigma^2 + eps
    float* wf;                    // [nq, 15 * cin]
    // deformable KPConv (kpconv.py:1011-1066): the inner convolution's output per query, [nq, off_dim]: 15 x 3 offsets in
    // units of the extent (+ 15 modulation logits when off_dim = 60); null = rigid
    const float* off; int off_dim; float extent;
    // kp_agg_mfma works on a SLICE of 16 NT channels starting at c_off of rows that are c_total floats wide (cin = 512 runs
    // as two slices of 256); c_total = cin, c_off = 0 otherwise
    int c_total, c_off;
};

__device__ __forceinline__ float kp_influence(float d2, const KpArgs& A) {
    if (A.influence == 0) return 1.0f;
    if (A.influence == 1) { float w = 1.0f - sqrtf(d2) * A.inv_extent; return w > 0.f ? w : 0.f; }
    return expf(-d2 / A.gauss_den);
}

typedef float kp_v2f __attribute__((ext_vector_type(2)));

// the influences of TWO kernel points at once on packed f32 (v_pk_add / v_pk_mul / v_pk_fma): the 15 x (3 sub, 3 mul, 2 add,
// sqrt, fma, max) per (query, neighbour) pair are 40 % of the gather kernel's instructions.
// NOTE: d2 and 1 - d / extent are FUSED here (explicit fma, unaffected by -ffp-contract=off) while the scalar
// kp_influence above rounds every product: the two differ in the last ulp, so which kernel a layer takes (kp_small_fused /
// kp_weighted on this form, kp_weighted_small on the scalar one) shows in the last bit of the influence weights.  KPConv
// features are a FLOAT row of SURVEY.md §8 (tolerance 1e-4 on the logits, tests/test_gpu_kpconv.py), never a bit-exact one
// -- unlike the index ops, whose d2 must stay unfused.
__device__ __forceinline__ kp_v2f kp_influence2(kp_v2f dx, kp_v2f dy, kp_v2f dz, const KpArgs& A) {
    const kp_v2f d2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
    if (A.influence == 0) return (kp_v2f){1.0f, 1.0f};
    if (A.influence == 1) {
        const kp_v2f d = (kp_v2f){sqrtf(d2.x), sqrtf(d2.y)};
        const kp_v2f w = __builtin_elementwise_fma(d, (kp_v2f){-A.inv_extent, -A.inv_extent}, (kp_v2f){1.0f, 1.0f});
        return (kp_v2f){w.x > 0.f ? w.x : 0.f, w.y > 0.f ? w.y : 0.f};
    }
    return (kp_v2f){expf(-d2.x / A.gauss_den), expf(-d2.y / A.gauss_den)};
}

// G lanes per query (G in {16, 32, 64}); a wave serves 64 / G queries; J = channels per lane.
template <int G, int J>
__global__ void __launch_bounds__(256) kp_weighted(KpArgs A) {
    constexpr int QW = 64 / G;
    __shared__ __attribute__((aligned(16))) float W[4][QW][KP_HC][KP_WP];
    __shared__ int NI[4][QW][KP_HC];
    __shared__ float KPs[KP_WP * 3];                 // x[16] | y[16] | z[16]; the sixteenth point is padding
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < KP_WP * 3) KPs[tid] = (tid % KP_WP) < KP_K ? A.kp[3 * (tid % KP_WP) + tid / KP_WP] : 0.f;
    __syncthreads();
    const int qi = lane / G, cg = lane % G;
    const int64_t q0 = ((int64_t)blockIdx.x * 4 + wave) * QW;   // first query of this wave
    if (q0 >= A.nq) return;                                      // wave-uniform, no block barrier below
    const int64_t q = q0 + qi;
    const bool q_ok = q < A.nq;
    // accumulators as PAIRS of kernel points (v_pk_fma_f32: two multiply-adds per lane and instruction -- the f32 matrix
    // rate without the MFMA's 16 / 32-row granularity; the sixteenth slot of the last pair is padding, never stored)
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f acc[KP_WP / 2][J];
#pragma unroll
    for (int k = 0; k < KP_WP / 2; ++k)
#pragma unroll
        for (int j = 0; j < J; ++j) acc[k][j] = (v2f){0.f, 0.f};

    for (int h0 = 0; h0 < A.h; h0 += KP_HC) {
        // ---- phase 1: influence weights of the chunk's (query, neighbour) pairs -> LDS -------------------
        for (int pr = lane; pr < QW * KP_HC; pr += 64) {
            const int pq = pr / KP_HC, ph = pr % KP_HC;
            const int64_t qq = q0 + pq;
            const int hh = h0 + ph;
            int idx = -1;
            if (qq < A.nq && hh < A.h) {
                idx = A.inds[qq * A.h + hh];
                if (idx < 0 || idx >= A.ns) idx = -1;            // shadow neighbour (kpconv.py:1048-1051)
            }
            NI[wave][pq][ph] = idx;
            if (idx >= 0) {
                const float* sp = A.s_pts + 3 * (int64_t)idx;
                const float* qp = A.q_pts + 3 * qq;
                const float nx = sp[0] - qp[0], ny = sp[1] - qp[1], nz = sp[2] - qp[2];
                const kp_v2f* kx = reinterpret_cast<const kp_v2f*>(KPs);
                const kp_v2f* ky = reinterpret_cast<const kp_v2f*>(KPs + KP_WP);
                const kp_v2f* kz = reinterpret_cast<const kp_v2f*>(KPs + 2 * KP_WP);
                kp_v2f* wrow = reinterpret_cast<kp_v2f*>(&W[wave][pq][ph][0]);
#pragma unroll
                for (int k = 0; k < KP_WP / 2; ++k)
                    wrow[k] = kp_influence2((kp_v2f){nx, nx} - kx[k], (kp_v2f){ny, ny} - ky[k], (kp_v2f){nz, nz} - kz[k], A);
                W[wave][pq][ph][KP_K] = 0.f;                     // the padding slot of the last pair
            }
        }
        wave_sync();
        // ---- phase 2: stream the neighbours' feature rows --------------------------------------------------
        const int hn = (A.h - h0) < KP_HC ? (A.h - h0) : KP_HC;
        if (q_ok) {
            for (int ph = 0; ph < hn; ++ph) {
                const int idx = NI[wave][qi][ph];
                if (idx < 0) continue;
                const float* xr = A.x + (int64_t)idx * A.cin;
                float xv[J];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    const int c = cg + j * G;
                    xv[j] = c < A.cin ? xr[c] : 0.f;
                }
                const float4* wp = reinterpret_cast<const float4*>(&W[wave][qi][ph][0]);
                const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
                const v2f wk[8] = {(v2f){w0.x, w0.y}, (v2f){w0.z, w0.w}, (v2f){w1.x, w1.y}, (v2f){w1.z, w1.w},
                                   (v2f){w2.x, w2.y}, (v2f){w2.z, w2.w}, (v2f){w3.x, w3.y}, (v2f){w3.z, w3.w}};
#pragma unroll
                for (int k = 0; k < KP_WP / 2; ++k)
#pragma unroll
                    for (int j = 0; j < J; ++j) acc[k][j] = __builtin_elementwise_fma(wk[k], (v2f){xv[j], xv[j]}, acc[k][j]);
            }
        }
        wave_sync();
    }
    if (q_ok) {
        float* o = A.wf + q * (int64_t)(KP_K * A.cin);
#pragma unroll
        for (int k = 0; k < KP_K; ++k)
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int c = cg + j * G;
                if (c < A.cin) o[k * A.cin + c] = (k & 1) ? acc[k >> 1][j].y : acc[k >> 1][j].x;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// kp_agg_mfma<NT> -- the same weighted sum (Cin = 16 NT) on v_mfma_f32_16x16x4_f32, one WAVE per query:
//   wf[k][c] = sum_h w[k][h] * x[idx_h][c]   =   A [16 kernel-point slots x 4 neighbours]  x  B [4 neighbours x 16 channels]
// per group of four neighbours and per 16-channel tile.  Lane (k = lane & 15, j = lane >> 4) owns neighbour j of the group:
// it computes the ONE influence w[k][j] the A operand wants from it (its kernel point k in three registers, the neighbour's
// position one broadcast load per 16 lanes) and loads the NT channels k NT .. k NT + NT - 1 of that neighbour's feature row (the
// 16 lanes of a neighbour read one contiguous row) as its B values -- tile nt holds channels {k NT + nt}.  No LDS at all.
// kp_weighted above gives every lane of a query ALL sixteen weights of every neighbour through LDS: four ds_read_b128 per
// neighbour and wave, 32 cycles of the CU's one LDS return path against 32 cycles of packed FMAs on each of FOUR SIMDs -- the
// Cin = 32 block ran LDS-bound at 1.1 ms for 640 000 queries (profiles/r03_pmc_kp_fetch.csv).  Here the weights never leave
// the lane that computes them, and the product runs on the matrix unit at the same f32 rate.
// The row's indices are read once (lane = column, two registers for widths up to 128); the walk ends at the row's last real
// neighbour (dense rows list the real ones first: ~30 of 73 columns at the Toronto3D bench size, the rest is padding that the
// lockstep loops of kp_weighted iterate over), wherever the shadows sit.
// ------------------------------------------------------------------------------------------------------------------
template <int MODE>       // 0 constant, 1 linear, 2 gaussian: compile-time here (no branch per group of four neighbours)
__device__ __forceinline__ float kp_influence1(float dx, float dy, float dz, const KpArgs& A) {
    // (the arithmetic of kp_influence2, one kernel point: fused d2, fused 1 - d / extent; v_sqrt_f32 itself -- 1 ulp -- instead
    //  of sqrtf()'s 20-instruction correctly rounded sequence: the kernel is issue-bound and the weights are a float row)
    if constexpr (MODE == 0) return 1.0f;
    const float d2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    if constexpr (MODE == 1) { const float w = fmaf(__builtin_amdgcn_sqrtf(d2), -A.inv_extent, 1.0f); return w > 0.f ? w : 0.f; }
    return expf(-d2 / A.gauss_den);
}

__device__ __forceinline__ uint32_t ml3d_umul24(uint32_t a, uint32_t b) {       // v_mul_u32_u24: full rate (v_mul_lo_u32 is quarter rate)
#ifdef ML3D_HIPEMU
    return a * b;
#else
    return __umul24(a, b);
#endif
}

template <int NT>
struct KpRow {                       // one neighbour as lane (k, j) sees it: its influence on kernel point k + its NT channels
    float w;                         // (PSH = false: w, wy, wz hold the neighbour's position until it is consumed)
    float wy, wz;
    float xv[NT];
    bool real;
};

template <int NT>
__device__ __forceinline__ void kp_row_load(KpRow<NT>& r, const KpArgs& A, int idx, int k) {
    // UNCONDITIONAL load (a shadow lane reads row 0 and is zeroed when consumed): loads inside an `if (idx >= 0)` leave the
    // number of outstanding requests unknown, and the compiler then waits for ALL of them before the previous group's use
    r.real = idx >= 0;
    // (32-bit BYTE offset from the scalar base -- one 24-bit multiply-add instead of a 64-bit multiply and a 64-bit shift-add;
    //  agg_mfma_ok: fewer than 2^24 support rows, a row shorter than 2^24 bytes, the feature matrix below 4 GB)
    const uint32_t i = (uint32_t)(idx < 0 ? 0 : idx);
    const float* xr = reinterpret_cast<const float*>(reinterpret_cast<const char*>(A.x) +
                                                     (ml3d_umul24(i, (uint32_t)A.c_total * 4u) + (uint32_t)(A.c_off + k * NT) * 4u));
    if constexpr (NT == 1) r.xv[0] = xr[0];
    else if constexpr (NT == 2) { const float2 v = *reinterpret_cast<const float2*>(xr); r.xv[0] = v.x; r.xv[1] = v.y; }
    else {
#pragma unroll
        for (int t = 0; t < NT / 4; ++t) {
            const float4 v = reinterpret_cast<const float4*>(xr)[t];
            r.xv[4 * t] = v.x; r.xv[4 * t + 1] = v.y; r.xv[4 * t + 2] = v.z; r.xv[4 * t + 3] = v.w;
        }
    }
}

// one query's weighted sum on the matrix unit: lane (k = lane & 15, j = lane >> 4) ends with acc[n][r] = wf[kernel point 4 j + r]
// [channel k NT + n].  `ia`, `ib`: the lane's column of the query's index row (columns lane and 64 + lane, -1 past the row),
// requested by the caller one query ahead.
template <int NT, int MODE, bool DEF, bool PSH>
__device__ __forceinline__ void kp_agg_query(const KpArgs& A, int64_t q, int lane, int ia, int ib, float qx, float qy, float qz,
                                             float kx, float ky, float kz, bool kreal,
                                             float __attribute__((ext_vector_type(4))) (&acc)[NT]) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int k = lane & 15, j = lane >> 4;
    const int32_t* row_cur = A.inds + q * A.h;
    // DEF: this query's kernel points = the layer's + its offsets (lane k owns point k; the 16 lanes j of a point agree)
    float kxq = kx, kyq = ky, kzq = kz, modq = 1.0f;
    if constexpr (DEF) {
        if (kreal) {
            const float* o = A.off + q * (int64_t)A.off_dim;
            kxq = fmaf(o[3 * k], A.extent, kx); kyq = fmaf(o[3 * k + 1], A.extent, ky); kzq = fmaf(o[3 * k + 2], A.extent, kz);
            if (A.off_dim > 3 * KP_K) modq = 2.0f / (1.0f + expf(-o[3 * KP_K + k]));      // 2 sigmoid (kpconv.py:1023-1024)
        }
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // rows wider than 128 columns (the deformable layers search with deform_radius) are walked 128 columns at a time; the
    // first block's indices were requested under the previous query, later blocks are read here
    for (int cb = 0; cb < A.h; cb += 128) {
        if (cb > 0) {
            ia = cb + lane < A.h ? row_cur[cb + lane] : -1;
            ib = cb + 64 + lane < A.h ? row_cur[cb + 64 + lane] : -1;
        }
        if (ia < 0 || ia >= A.ns) ia = -1;                           // shadow neighbour (kpconv.py:1048-1051)
        if (ib < 0 || ib >= A.ns) ib = -1;
        const unsigned long long ma = __ballot(ia >= 0), mb = __ballot(ib >= 0);
        const int count = mb ? 128 - __builtin_clzll(mb) : (ma ? 64 - __builtin_clzll(ma) : 0);   // last real column + 1
        const int groups = (count + 3) >> 2;
        // PSH: the neighbours' positions relative to the query are loaded ONE column per lane (6 independent loads per 128
        // columns) and the 16 lanes that weigh a neighbour fetch them with cross-lane reads.  !PSH: every lane loads its
        // group's neighbour itself -- three more dependent loads per group (16 lanes reading the same 12 bytes), no LDS-pipe
        // traffic.  Measured: the fused 32 -> 32 kernel (4 waves per SIMD to hide latency with) 1.00 -> 0.90 ms with PSH; the
        // KPConv step with PSH in the stand-alone aggregations too (6-8 waves per SIMD) 9.17 against 8.65 ms.
        float nxa = 0.f, nya = 0.f, nza = 0.f, nxb = 0.f, nyb = 0.f, nzb = 0.f;
        if constexpr (PSH) {
            const float* pa = A.s_pts + 3 * (int64_t)(ia < 0 ? 0 : ia);
            const float* pb = A.s_pts + 3 * (int64_t)(ib < 0 ? 0 : ib);
            nxa = pa[0] - qx; nya = pa[1] - qy; nza = pa[2] - qz;
            nxb = pb[0] - qx; nyb = pb[1] - qy; nzb = pb[2] - qz;
        }
        // The 128 columns are walked as two halves of 16 groups (columns 0..63 from the lane's first register set, 64..127 from
        // its second): inside a half the cross-lane source of group g is simply lane 4 g + j -- no per-group select between the
        // two sets (five VALU instructions per group in the fused kernel, which is issue-bound: profiles/r04_pmc_kp_sq1.csv).
        for (int half = 0; half < 2; ++half) {
            const int gh = min(16, groups - 16 * half);               // groups of this half (wave-uniform)
            if (gh <= 0) break;
            const int ci = half ? ib : ia;
            const float cnx = half ? nxb : nxa, cny = half ? nyb : nya, cnz = half ? nzb : nza;
            // this lane's neighbour of group g of the half; idx -1 past its last group (the prefetch runs two groups ahead)
            auto fetch = [&](KpRow<NT>& r, int g) {
                const int src = (4 * g + j) & 63;
                const int v = __shfl(ci, src);
                const int idx = g < gh ? v : -1;
                if constexpr (PSH) {
                    const float nx = __shfl(cnx, src), ny = __shfl(cny, src), nz = __shfl(cnz, src);
                    kp_row_load<NT>(r, A, idx, k);
                    const float w = kp_influence1<MODE>(nx - kxq, ny - kyq, nz - kzq, A);
                    r.w = (kreal && idx >= 0) ? w : 0.f;
                } else {
                    const float* sp = A.s_pts + 3 * (int64_t)(idx < 0 ? 0 : idx);
                    r.w = sp[0]; r.wy = sp[1]; r.wz = sp[2];
                    kp_row_load<NT>(r, A, idx, k);
                }
            };
            auto consume = [&](const KpRow<NT>& r) {
                float w = r.w;
                if constexpr (!PSH) {
                    const float nx = r.w - qx, ny = r.wy - qy, nz = r.wz - qz;
                    w = kp_influence1<MODE>(nx - kxq, ny - kyq, nz - kzq, A);
                    w = (kreal && r.real) ? w : 0.f;
                }
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, r.real ? r.xv[n] : 0.f, acc[n], 0, 0, 0);
            };
            // two STATIC row buffers, the loop unrolled by two: the loads of group g + 1 are in flight while group g is
            // consumed (a rotating `cur = nxt` form made the compiler wait for the loads it had just issued)
            // (a third row buffer -- two groups in flight -- measured slower in the fused kernel: 0.94 against 0.89 ms)
            KpRow<NT> ra, rb;
            fetch(ra, 0);
            for (int g = 0; g < gh; g += 2) {
                // (sched_barrier: the scheduler otherwise issues both rows' loads together and waits for both before the first
                //  MFMA -- the request counter retires in order, so a use may only wait for the OLDER row while the younger is
                //  in flight)
                fetch(rb, g + 1);
                __builtin_amdgcn_sched_barrier(0);
                consume(ra);
                __builtin_amdgcn_sched_barrier(0);
                fetch(ra, g + 2);
                __builtin_amdgcn_sched_barrier(0);
                consume(rb);              // (unconditional: past the last group the index is -1, a zero MFMA -- under an `if`
                                          //  the compiler sinks rb's loads into the branch, right in front of their use)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (DEF) {
        // modulations scale kernel point kk's row of the weighted features (kpconv.py:1147-1149); this lane holds rows
        // 4 j .. 4 j + 3, lane kk holds modulation kk
        if (A.off_dim > 3 * KP_K) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m = __shfl(modq, 4 * j + r);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n][r] *= m;
            }
        }
    }
}

template <int NT, int MODE, bool DEF = false>
__global__ void __launch_bounds__(256) kp_agg_mfma(KpArgs A) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane & 15, j = lane >> 4;
    const bool kreal = k < KP_K;                                  // slot 15 is padding: weight 0, never stored
    const float kx = kreal ? A.kp[3 * k] : 0.f, ky = kreal ? A.kp[3 * k + 1] : 0.f, kz = kreal ? A.kp[3 * k + 2] : 0.f;
    // XCD-contiguous walk (workgroup b runs on XCD b & 7): each XCD takes one contiguous eighth of the queries, so the feature
    // rows that neighbouring queries share stay in that XCD's L2
    const int64_t per_xcd = (A.nq + 7) / 8;
    const int64_t qbase = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t stride = (int64_t)(gridDim.x >> 3) * 4;
    // the NEXT query's index row and position are requested while this one is aggregated (a wave walks its queries serially:
    // without this every query starts with two dependent round trips -- row, then first neighbours -- that nothing covers);
    // unconditional, clamped loads again: a wave past its last query re-reads query 0 and drops it
    int ia_n, ib_n;
    float qx_n, qy_n, qz_n;
    auto request = [&](int64_t tt) {
        const int64_t qq = qbase + tt;
        const int64_t qc = (tt < per_xcd && qq < A.nq) ? qq : 0;
        const int32_t* row = A.inds + qc * A.h;
        ia_n = row[lane < A.h ? lane : A.h - 1];
        ib_n = row[64 + lane < A.h ? 64 + lane : A.h - 1];
        const float* qp = A.q_pts + 3 * qc;
        qx_n = qp[0]; qy_n = qp[1]; qz_n = qp[2];
    };
    int64_t t = (int64_t)(blockIdx.x >> 3) * 4 + wave;
    request(t);
    for (; t < per_xcd; t += stride) {
        const int64_t q = qbase + t;
        if (q >= A.nq) break;
        const int ia = lane < A.h ? ia_n : -1, ib = 64 + lane < A.h ? ib_n : -1;
        const float qx = qx_n, qy = qy_n, qz = qz_n;
        request(t + stride);
        f32x4 acc[NT];
        kp_agg_query<NT, MODE, DEF, false>(A, q, lane, ia, ib, qx, qy, qz, kx, ky, kz, kreal, acc);
        // D: lane (column k, j) holds kernel points 4 j .. 4 j + 3 of channels k NT .. k NT + NT - 1
        float* o = A.wf + q * (int64_t)(KP_K * A.c_total) + A.c_off + k * NT;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = 4 * j + r;
            if (kk < KP_K) {
                float* dst = o + kk * A.c_total;
                if constexpr (NT == 1) dst[0] = acc[0][r];
                else if constexpr (NT == 2) *reinterpret_cast<float2*>(dst) = make_float2(acc[0][r], acc[1][r]);
                else {
#pragma unroll
                    for (int t4 = 0; t4 < NT / 4; ++t4)
                        reinterpret_cast<float4*>(dst)[t4] = make_float4(acc[4 * t4][r], acc[4 * t4 + 1][r], acc[4 * t4 + 2][r],
                                                                         acc[4 * t4 + 3][r]);
                }
            }
        }
    }
}

// the [15 * Cin] x [Cout] product that follows the aggregation, for the kernels that carry it themselves
struct KpOut {
    const float* weights;         // [15 * cin][cout]
    const float* bias;            // [cout] or null
    int act; float slope; int cout;
    float* out;                   // [nq, cout]
};

// ---- the Cin = 32 -> Cout = 32 block in ONE kernel (the KPConv of the full-resolution resnet bottlenecks of every reference
// config: 640 000 queries per 64-sphere step) -- north_star's "LDS-tiled gather-GEMM", built on the MFMA aggregation.
// The two-kernel form writes wf [Nq, 480] to HBM and reads it back in the GEMM: 2 x 1.23 GB per launch, ~9x the op's algorithmic
// bytes (profiles/r04_pmc_kp_*.csv).  Here a workgroup of four waves owns a tile of 16 queries: every wave aggregates four of
// them with kp_agg_query and parks its D registers -- which ARE wf -- in an LDS tile [16][480] (pitch 516: conflict-free 8-byte
// reads); after one LDS barrier the tile is the A operand of `tile x W`: wave w multiplies K slice [120 w, 120 w + 120) against
// ITS rows of the kernel weights, 60 values per lane that live in registers for the whole (persistent) kernel; the four partial
// [16 x 32] blocks cross LDS (in the tile's own memory), are summed with bias + activation and leave as 128-byte rows.
// Round 3's kp_fused32 had the same second half on a packed-FMA aggregation (168 registers, 50 KB of LDS: 3 waves per SIMD,
// slower than two kernels); the MFMA aggregation needs a third of the registers.
constexpr int KF_TQ = 16;                 // queries per tile
constexpr int KF_PITCH = 516;             // floats per tile row: 516 = 8 * 64 + 4 -> rows land 4 banks apart

template <int MODE>
__global__ void __launch_bounds__(256) ML3D_WAVES_PER_SIMD(4) kp_agg_gemm32(KpArgs A, KpOut O) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float tile[KF_TQ * KF_PITCH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = lane & 15, j = lane >> 4;
    const bool kreal = k < KP_K;
    const float kx = kreal ? A.kp[3 * k] : 0.f, ky = kreal ? A.kp[3 * k + 1] : 0.f, kz = kreal ? A.kp[3 * k + 2] : 0.f;
    // this lane's share of the kernel weights: B operand of step s, column tile t = W[120 wave + 30 j + s][16 t + k]
    // (K index 120 wave + 30 j + s: lane (i, j) of the A operand reads 30 CONSECUTIVE tile floats -- any bijection of the K
    //  indices onto (step, j) sums the same products)
    float wb[2][30];
#pragma unroll
    for (int s = 0; s < 30; ++s) {
        const float* wr = O.weights + (int64_t)(120 * wave + 30 * j + s) * 32 + k;
        wb[0][s] = wr[0];
        wb[1][s] = wr[16];
    }
    const int64_t tiles = (A.nq + KF_TQ - 1) / KF_TQ;
    const int64_t per_xcd = (tiles + 7) / 8;
    const int64_t tbase = (int64_t)(blockIdx.x & 7) * per_xcd;
    const int64_t stride = (int64_t)(gridDim.x >> 3);
    int ia_n, ib_n;
    float qx_n, qy_n, qz_n;
    auto request = [&](int64_t q) {
        const int64_t qc = q < A.nq ? q : 0;
        const int32_t* row = A.inds + qc * A.h;
        ia_n = row[lane < A.h ? lane : A.h - 1];
        ib_n = row[64 + lane < A.h ? 64 + lane : A.h - 1];
        const float* qp = A.q_pts + 3 * qc;
        qx_n = qp[0]; qy_n = qp[1]; qz_n = qp[2];
    };
    int64_t tt = (int64_t)(blockIdx.x >> 3);
    if (tt < per_xcd && tbase + tt < tiles) request((tbase + tt) * KF_TQ + 4 * wave);
    for (; tt < per_xcd && tbase + tt < tiles; tt += stride) {
        const int64_t q0 = (tbase + tt) * KF_TQ;
        // ---- aggregate: wave w fills rows 4 w .. 4 w + 3 of the tile
#pragma unroll 1
        for (int u = 0; u < 4; ++u) {
            const int64_t q = q0 + 4 * wave + u;
            const int ia = lane < A.h ? ia_n : -1, ib = 64 + lane < A.h ? ib_n : -1;
            const float qx = qx_n, qy = qy_n, qz = qz_n;
            // the next query of this wave: in this tile, or the first one of its next tile
            const int64_t qn = u < 3 ? q + 1 : (tbase + tt + stride) * KF_TQ + 4 * wave;
            request((u < 3 || (tt + stride < per_xcd && tbase + tt + stride < tiles)) ? qn : 0);
            f32x4 acc[2];
            if (q < A.nq) kp_agg_query<2, MODE, false, true>(A, q, lane, ia, ib, qx, qy, qz, kx, ky, kz, kreal, acc);
            else { acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[1] = acc[0]; }
            float* trow = tile + (4 * wave + u) * KF_PITCH + 2 * k;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = 4 * j + r;
                if (kk < KP_K) *reinterpret_cast<float2*>(trow + kk * 32) = make_float2(acc[0][r], acc[1][r]);
            }
        }
        block_sync_lds();
        // ---- multiply: [16 x 120] slice of the tile x this wave's [120 x 32] rows of W
        f32x4 c0 = (f32x4){0.f, 0.f, 0.f, 0.f}, c1 = c0;
        {
            const float* arow = tile + k * KF_PITCH + 120 * wave + 30 * j;          // A operand: lane (i = k, j)
            // (read in chunks of 10: all 30 A values at once, on top of the 60 weights, spill)
#pragma unroll
            for (int s0 = 0; s0 < 30; s0 += 10) {
                float av[10];
#pragma unroll
                for (int s = 0; s < 10; s += 2) {
                    const float2 v = *reinterpret_cast<const float2*>(arow + s0 + s);
                    av[s] = v.x; av[s + 1] = v.y;
                }
#pragma unroll
                for (int s = 0; s < 10; ++s) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], wb[0][s0 + s], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s], wb[1][s0 + s], c1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        block_sync_lds();               // every wave has read the tile: its memory takes the four partial blocks
        {
            // D: lane (col k, j) holds rows 4 j .. 4 j + 3 -> part[wave][row][col], pitch 36
            float* part = tile + wave * (KF_TQ * 36);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                part[(4 * j + r) * 36 + k] = c0[r];
                part[(4 * j + r) * 36 + 16 + k] = c1[r];
            }
        }
        block_sync_lds();
        {
            // 512 outputs, two per thread: (row = tid / 16, cols 2 (tid % 16), + 1)
            const int row = threadIdx.x >> 4, col = 2 * (threadIdx.x & 15);
            float2 v = make_float2(0.f, 0.f);
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) {
                const float2 p = *reinterpret_cast<const float2*>(tile + w4 * (KF_TQ * 36) + row * 36 + col);
                v.x += p.x; v.y += p.y;
            }
            if (O.bias) { v.x += O.bias[col]; v.y += O.bias[col + 1]; }
            if (O.act == 1) { v.x = v.x > 0.f ? v.x : v.x * O.slope; v.y = v.y > 0.f ? v.y : v.y * O.slope; }
            else if (O.act == 2) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
            const int64_t q = q0 + row;
            if (q < A.nq) *reinterpret_cast<float2*>(O.out + q * 32 + col) = v;
        }
        block_sync_lds();               // the partial blocks are consumed before the next tile's rows are written
    }
}

// tiny Cin (the first layer, in_features_dim in {1, 2, 4, 5}): one thread per query, weights on the fly
template <int CIN>
__global__ void __launch_bounds__(256) kp_weighted_small(KpArgs A) {
    __shared__ float KPs[KP_K * 3];
    if (threadIdx.x < KP_K * 3) KPs[threadIdx.x] = A.kp[threadIdx.x];
    __syncthreads();
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= A.nq) return;
    float acc[KP_K][CIN];
#pragma unroll
    for (int k = 0; k < KP_K; ++k)
#pragma unroll
        for (int c = 0; c < CIN; ++c) acc[k][c] = 0.f;
    const float qx = A.q_pts[3 * q], qy = A.q_pts[3 * q + 1], qz = A.q_pts[3 * q + 2];
    for (int hh = 0; hh < A.h; ++hh) {
        const int idx = A.inds[q * A.h + hh];
        if (idx < 0 || idx >= A.ns) continue;
        const float* sp = A.s_pts + 3 * (int64_t)idx;
        const float nx = sp[0] - qx, ny = sp[1] - qy, nz = sp[2] - qz;
        float xv[CIN];
#pragma unroll
        for (int c = 0; c < CIN; ++c) xv[c] = A.x[(int64_t)idx * CIN + c];
#pragma unroll
        for (int k = 0; k < KP_K; ++k) {
            const float dx = nx - KPs[3 * k], dy = ny - KPs[3 * k + 1], dz = nz - KPs[3 * k + 2];
            const float w = kp_influence(dx * dx + dy * dy + dz * dz, A);
#pragma unroll
            for (int c = 0; c < CIN; ++c) acc[k][c] = fmaf(w, xv[c], acc[k][c]);
        }
    }
    float* o = A.wf + q * (int64_t)(KP_K * CIN);
#pragma unroll
    for (int k = 0; k < KP_K; ++k)
#pragma unroll
        for (int c = 0; c < CIN; ++c) o[k * CIN + c] = acc[k][c];
}

// ------------------------------------------------------------------------------------------------------------------
// kp_small_fused — the whole KPConv of the FIRST layer (in_features_dim in {1 .. 5}, 64 output channels in the reference
// configs) in one kernel: the [Nq, 15 * Cin] intermediate never exists in HBM and there is no K = 15 GEMM launch.
// lane = (query of the wave's 8, PAIR of kernel points): per neighbour the lane computes the two influences of its pair
// (packed f32) and multiplies them into Cin packed accumulators; neighbour index / position / feature loads are the same
// address for the 8 lanes of a query (broadcast) and are requested U neighbours ahead of their use.  (The thread-per-query
// kernel above walks 64 different index rows per instruction: 0.44 ms for 320 000 queries, bound by the address coalescer.)
// Then the 15 * Cin weighted features of the wave's 8 queries go through LDS to the output layout -- lane = (query, 1/8 of
// the output channels) -- for the [15 * Cin] x [Cout] product against the LDS-resident kernel weights, bias (folded BN) and
// activation included.
// ------------------------------------------------------------------------------------------------------------------
template <int CIN, int NV>       // NV = cout / 32 float4 column groups per lane
__global__ void __launch_bounds__(256) kp_small_fused(KpArgs A, KpOut O) {
    constexpr int KC = KP_K * CIN;                    // rows of the weight matrix
    constexpr int WFP = KP_WP * CIN + 4;              // LDS pitch of a query's weighted features
    constexpr int U = 4;
    HIP_DYNAMIC_SHARED(float, smem)
    float* WT = smem;                                 // [KC][cout]
    float* WF = smem + KC * O.cout;                   // [4 waves][8 queries][WFP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < KC * O.cout; i += 256) WT[i] = O.weights[i];
    __syncthreads();
    const int qi = lane >> 3, kp = lane & 7;
    const int64_t q0 = ((int64_t)blockIdx.x * 4 + wave) * 8;
    if (q0 >= A.nq) return;                           // wave-uniform, no block barrier below
    const int64_t q = q0 + qi < A.nq ? q0 + qi : A.nq - 1;     // (the tail lanes redo the last query and do not store)
    const bool q_ok = q0 + qi < A.nq;
    typedef float v2f __attribute__((ext_vector_type(2)));
    // the lane's two kernel points (the sixteenth is padding: its accumulator is never read)
    const int k0 = 2 * kp, k1 = 2 * kp + 1 < KP_K ? 2 * kp + 1 : 2 * kp;
    const v2f kx = (v2f){A.kp[3 * k0], A.kp[3 * k1]}, ky = (v2f){A.kp[3 * k0 + 1], A.kp[3 * k1 + 1]},
              kz = (v2f){A.kp[3 * k0 + 2], A.kp[3 * k1 + 2]};
    const float qx = A.q_pts[3 * q], qy = A.q_pts[3 * q + 1], qz = A.q_pts[3 * q + 2];
    v2f acc[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[c] = (v2f){0.f, 0.f};
    const int32_t* irow = A.inds + q * A.h;
    // columns in use: up to the last real neighbour of any of the wave's 8 queries.  Dense rows list the real neighbours first
    // and pad to the widest row of the BATCH (73 columns at the Toronto3D bench size, ~30 real on average): the lockstep walk
    // below paid full price for every shadow column.  The 8 lanes of a query scan its row once (column = lane & 7 + 8 i; the
    // same lines the walk reads next), then a wave-wide max -- exact wherever the shadows sit.
    int hend = 0;
    for (int c0 = kp; c0 < A.h; c0 += 8) {
        const int v = irow[c0];
        if (v >= 0 && v < A.ns && q_ok) hend = c0 + 1;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(hend, m); hend = o > hend ? o : hend; }
    for (int h0 = 0; h0 < hend; h0 += U) {
        int idx[U];
        float sx[U], sy[U], sz[U], xv[U][CIN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            idx[u] = h0 + u < A.h ? irow[h0 + u] : -1;
            if (idx[u] < 0 || idx[u] >= A.ns) idx[u] = -1;       // shadow neighbour (kpconv.py:1048-1051)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = idx[u] < 0 ? 0 : idx[u];
            const bool ok = idx[u] >= 0;
            sx[u] = ok ? A.s_pts[3 * r] : 0.f; sy[u] = ok ? A.s_pts[3 * r + 1] : 0.f; sz[u] = ok ? A.s_pts[3 * r + 2] : 0.f;
#pragma unroll
            for (int c = 0; c < CIN; ++c) xv[u][c] = ok ? A.x[r * CIN + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float nx = sx[u] - qx, ny = sy[u] - qy, nz = sz[u] - qz;
            const v2f w = kp_influence2((v2f){nx, nx} - kx, (v2f){ny, ny} - ky, (v2f){nz, nz} - kz, A);
#pragma unroll
            for (int c = 0; c < CIN; ++c) acc[c] = __builtin_elementwise_fma(w, (v2f){xv[u][c], xv[u][c]}, acc[c]);
        }
    }
    // weighted features -> LDS in [k][c] order (row index of the weight matrix = k * CIN + c)
    float* wf = WF + (wave * 8 + qi) * WFP;
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
        wf[k0 * CIN + c] = acc[c].x;
        if (2 * kp + 1 < KP_K) wf[(2 * kp + 1) * CIN + c] = acc[c].y;
    }
    wave_sync();
    // out[q][co] = act(bias[co] + sum_kk wf[kk] * WT[kk][co]); lane = (query, column group kp): float4 v covers
    // channels (v * 8 + kp) * 4 .. + 3, so the 8 lanes of a query write 128 contiguous bytes per v
    float4 o[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int co = (v * 8 + kp) * 4;
        o[v] = O.bias ? *reinterpret_cast<const float4*>(O.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll 5
    for (int kk = 0; kk < KC; ++kk) {
        const float f = wf[kk];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const float4 w4 = *reinterpret_cast<const float4*>(WT + kk * O.cout + (v * 8 + kp) * 4);
            o[v].x = fmaf(f, w4.x, o[v].x); o[v].y = fmaf(f, w4.y, o[v].y);
            o[v].z = fmaf(f, w4.z, o[v].z); o[v].w = fmaf(f, w4.w, o[v].w);
        }
    }
    if (q_ok) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float4 r = o[v];
            if (O.act == 1) {
                r.x = r.x > 0.f ? r.x : r.x * O.slope; r.y = r.y > 0.f ? r.y : r.y * O.slope;
                r.z = r.z > 0.f ? r.z : r.z * O.slope; r.w = r.w > 0.f ? r.w : r.w * O.slope;
            } else if (O.act == 2) {
                r.x = r.x > 0.f ? r.x : 0.f; r.y = r.y > 0.f ? r.y : 0.f; r.z = r.z > 0.f ? r.z : 0.f; r.w = r.w > 0.f ? r.w : 0.f;
            }
            *reinterpret_cast<float4*>(O.out + q * (int64_t)O.cout + (v * 8 + kp) * 4) = r;
        }
    }
}

template <int CIN>
static bool launch_small_fused(const KpArgs& a, const KpOut& o, hipStream_t st) {
    const int nv = o.cout / 32;
    const size_t sm = sizeof(float) * ((size_t)KP_K * CIN * o.cout + 4 * 8 * (KP_WP * CIN + 4));
    const unsigned nb = (unsigned)((a.nq + 31) / 32);
    auto go = [&](auto kern) {
        if (sm > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        hipLaunchKernelGGL(kern, dim3(nb), dim3(256), sm, st, a, o);
    };
    switch (nv) {
        case 1: go(kp_small_fused<CIN, 1>); break;
        case 2: go(kp_small_fused<CIN, 2>); break;
        case 3: go(kp_small_fused<CIN, 3>); break;
        default: go(kp_small_fused<CIN, 4>); break;
    }
    return true;
}

// the fused first-layer kernel takes cin <= 5 with 32 | cout <= 128 and 16-byte aligned weights / bias / out
static bool small_fused_ok(const KpArgs& a, const KpOut& o) {
    return a.cin <= 5 && o.cout % 32 == 0 && o.cout <= 128 &&
           ((((uintptr_t)o.weights) | ((uintptr_t)o.bias) | ((uintptr_t)o.out)) & 15) == 0;
}

// kp_row_load addresses a neighbour's feature row with a 24-bit x 24-bit multiply into a 32-bit byte offset
static bool kp_rows_32bit(const KpArgs& a) {
    return a.ns < (1 << 24) && (int64_t)a.c_total * 4 < (1 << 24) && (int64_t)a.ns * a.c_total * 4 < ((int64_t)1 << 32);
}

// the one-kernel block takes cin = cout = 32, rigid, 16-byte aligned features and 8-byte aligned outputs
static bool agg_gemm32_ok(const KpArgs& a, const KpOut& o) {
    return a.cin == 32 && o.cout == 32 && !a.off && a.h > 0 && a.ns > 0 && a.nq > 0 && kp_rows_32bit(a) &&
           ((((uintptr_t)a.x) & 15) | (((uintptr_t)o.out) & 7)) == 0;
}

static void launch_agg_gemm32(const KpArgs& a, const KpOut& o, hipStream_t st) {
    const int64_t tiles = (a.nq + KF_TQ - 1) / KF_TQ;
    int64_t nb = (tiles + 7) / 8 * 8;
    if (nb > 256 * 4) nb = 256 * 4;                               // persistent: 33 KB of LDS -> 4 workgroups per CU
    if (a.influence == 0) hipLaunchKernelGGL((kp_agg_gemm32<0>), dim3((unsigned)nb), dim3(256), 0, st, a, o);
    else if (a.influence == 1) hipLaunchKernelGGL((kp_agg_gemm32<1>), dim3((unsigned)nb), dim3(256), 0, st, a, o);
    else hipLaunchKernelGGL((kp_agg_gemm32<2>), dim3((unsigned)nb), dim3(256), 0, st, a, o);
}

// max over the listed neighbours (shadow rows are zeros) / feature of the first listed neighbour
__global__ void gather_pool_k(const float* __restrict__ x, int64_t ns, int c, const int32_t* __restrict__ inds,
                              int64_t nq, int h, int mode, float* __restrict__ out) {
    const int64_t total = nq * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = i / c;
        const int ch = (int)(i - q * c);
        float v;
        if (mode == 1) {
            const int idx = inds[q * h];
            v = (idx >= 0 && idx < ns) ? x[(int64_t)idx * c + ch] : 0.f;
        } else {
            v = -3.0e38f;
            for (int hh = 0; hh < h; ++hh) {
                const int idx = inds[q * h + hh];
                const float xv = (idx >= 0 && idx < ns) ? x[(int64_t)idx * c + ch] : 0.f;
                v = xv > v ? xv : v;
            }
        }
        out[i] = v;
    }
}

// the same for C % 4 == 0: a thread owns FOUR channels of one query (16-byte gathers; the neighbour index is read once per
// four outputs instead of once per output), the query's threads are adjacent lanes reading one contiguous row
__global__ void __launch_bounds__(256)
gather_pool_v4(const float4* __restrict__ x, int64_t ns, int c4, const int32_t* __restrict__ inds, int64_t nq, int h, int mode,
               float4* __restrict__ out) {
    const int64_t total = nq * c4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t q;                                      // (32-bit division whenever it fits: the 64-bit one is ~100 instructions)
        if (total < 0x7fffffffll) q = (int64_t)((unsigned)i / (unsigned)c4); else q = i / c4;
        const int cq = (int)(i - q * c4);
        const int32_t* row = inds + q * h;
        float4 v;
        if (mode == 1) {
            const int idx = row[0];
            v = (idx >= 0 && idx < ns) ? x[(int64_t)idx * c4 + cq] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            v = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            auto take = [&](const float4& xv) {
                v.x = xv.x > v.x ? xv.x : v.x; v.y = xv.y > v.y ? xv.y : v.y;
                v.z = xv.z > v.z ? xv.z : v.z; v.w = xv.w > v.w ? xv.w : v.w;
            };
            // four neighbours per trip: their indices, then their 16-byte rows, are requested together (one neighbour per trip
            // is two DEPENDENT round trips that only other waves can cover: 0.43 ms for the 141 000 x 128 pool of the first layer)
            int hh = 0;
            const uint32_t c4u = (uint32_t)c4, cqu = (uint32_t)cq;        // (launcher: ns * c4 < 2^32 on this path)
            for (; hh + 4 <= h; hh += 4) {
                const int i0 = row[hh], i1 = row[hh + 1], i2 = row[hh + 2], i3 = row[hh + 3];
                float4 x0 = zero, x1 = zero, x2 = zero, x3 = zero;
                if (i0 >= 0 && i0 < ns) x0 = x[(uint32_t)i0 * c4u + cqu];
                if (i1 >= 0 && i1 < ns) x1 = x[(uint32_t)i1 * c4u + cqu];
                if (i2 >= 0 && i2 < ns) x2 = x[(uint32_t)i2 * c4u + cqu];
                if (i3 >= 0 && i3 < ns) x3 = x[(uint32_t)i3 * c4u + cqu];
                take(x0); take(x1); take(x2); take(x3);
            }
            for (; hh < h; ++hh) {
                const int idx = row[hh];
                take((idx >= 0 && idx < ns) ? x[(int64_t)idx * c4 + cq] : zero);
            }
        }
        out[i] = v;
    }
}

template <int G, int J>
static void launch_kpw(const KpArgs& a, hipStream_t st) {
    const int qw = 64 / G;
    unsigned nb = (unsigned)((a.nq + 4 * qw - 1) / (4 * qw));
    hipLaunchKernelGGL((kp_weighted<G, J>), dim3(nb), dim3(256), 0, st, a);
}

template <int NT>
static void launch_agg_mfma(const KpArgs& a, hipStream_t st) {
    int64_t nb = ((a.nq + 3) / 4 + 7) / 8 * 8;                    // four queries (waves) per workgroup, a multiple of 8 workgroups
    if (nb > 256 * 8) nb = 256 * 8;
    if (a.off) { hipLaunchKernelGGL((kp_agg_mfma<NT, 1, true>), dim3((unsigned)nb), dim3(256), 0, st, a); return; }   // (linear only)
    if (a.influence == 0) hipLaunchKernelGGL((kp_agg_mfma<NT, 0>), dim3((unsigned)nb), dim3(256), 0, st, a);
    else if (a.influence == 1) hipLaunchKernelGGL((kp_agg_mfma<NT, 1>), dim3((unsigned)nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((kp_agg_mfma<NT, 2>), dim3((unsigned)nb), dim3(256), 0, st, a);
}

// the MFMA aggregation takes cin in {16, 32, 64, 128, 256} and 16-byte aligned features / wf
static bool agg_mfma_ok(const KpArgs& a) {
    const int c = a.cin;
    return (c == 16 || c == 32 || c == 64 || c == 128 || c == 256 || (c == 512 && a.off)) && a.h > 0 && a.ns > 0 &&
           a.nq > 0 && kp_rows_32bit(a) &&
           ((((uintptr_t)a.x) | ((uintptr_t)a.wf)) & 15) == 0;
}

static int launch_weighted(const KpArgs& a, hipStream_t st) {
    const int c = a.cin;
    if (agg_mfma_ok(a)) {
        switch (c) {
            case 16: launch_agg_mfma<1>(a, st); break;
            case 32: launch_agg_mfma<2>(a, st); break;
            case 64: launch_agg_mfma<4>(a, st); break;
            case 128: launch_agg_mfma<8>(a, st); break;
            case 256: launch_agg_mfma<16>(a, st); break;
            default: {                      // 512 (deformable only: the rigid form has kp_weighted<64, 8>): two slices of 256
                KpArgs h = a;
                launch_agg_mfma<16>(h, st);
                h.c_off = 256;
                launch_agg_mfma<16>(h, st);
                break;
            }
        }
        return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
    }
    const unsigned nbs = (unsigned)((a.nq + 255) / 256);
    if (c == 1) hipLaunchKernelGGL(kp_weighted_small<1>, dim3(nbs), dim3(256), 0, st, a);
    else if (c == 2) hipLaunchKernelGGL(kp_weighted_small<2>, dim3(nbs), dim3(256), 0, st, a);
    else if (c == 3) hipLaunchKernelGGL(kp_weighted_small<3>, dim3(nbs), dim3(256), 0, st, a);
    else if (c == 4) hipLaunchKernelGGL(kp_weighted_small<4>, dim3(nbs), dim3(256), 0, st, a);
    else if (c == 5) hipLaunchKernelGGL(kp_weighted_small<5>, dim3(nbs), dim3(256), 0, st, a);
    else if (c <= 16) launch_kpw<16, 1>(a, st);
    else if (c <= 32) launch_kpw<32, 1>(a, st);     // (16 lanes x 2 channels per query measured slower: 1.67 vs 1.49 ms at Cin = 32)
    else if (c <= 64) launch_kpw<64, 1>(a, st);
    else if (c <= 128) launch_kpw<64, 2>(a, st);
    else if (c <= 256) launch_kpw<64, 4>(a, st);
    else if (c <= 512) launch_kpw<64, 8>(a, st);
    else return ML3D_E_UNSUPPORTED;
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

static inline size_t kp_align(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace ml3d

using namespace ml3d;

extern "C" size_t ml3d_kpconv_workspace_bytes(int64_t n_queries, int cin, int cout, int num_kernel_points) {
    if (n_queries < 0 || cin <= 0 || cout <= 0 || num_kernel_points != KP_K) return 0;
    size_t b = kp_align(sizeof(float) * (size_t)(n_queries > 0 ? n_queries : 1) * KP_K * (size_t)cin);
    const size_t p32 = gemm_partial_bytes(n_queries, cout, KP_K * cin), pbf = gemm_partial_bytes_bf16x3(n_queries, cout, KP_K * cin);
    b += kp_align(p32 > pbf ? p32 : pbf);
    return b + 512;
}

static int kpconv_run(const float* q_pts, const float* s_pts, const int32_t* neighb_inds, int64_t n_queries,
                      int64_t n_supports, int64_t max_neighbors, const float* features, int cin,
                      const float* kernel_points, int num_kernel_points, float kp_extent, int kp_influence_mode,
                      const float* offset_features, int offset_dim, const float* weights, const float* bias, int act,
                      float slope, int cout, float* out, void* workspace, size_t workspace_bytes, void* stream,
                      const void* packed = nullptr) {
    if (n_queries < 0 || n_supports < 0 || max_neighbors < 0 || cin <= 0 || cout <= 0 || !(kp_extent > 0.f) ||
        kp_influence_mode < 0 || kp_influence_mode > 2 || max_neighbors > 0x7fffffff)
        return ML3D_E_INVALID;
    if (num_kernel_points != KP_K || cin > 512) return ML3D_E_UNSUPPORTED;
    if (n_queries == 0) return 0;
    if (!q_pts || !kernel_points || !weights || !out || (max_neighbors > 0 && (!neighb_inds || !features || !s_pts)))
        return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_kpconv_workspace_bytes(n_queries, cin, cout, num_kernel_points)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* wf = (float*)p;
    p += kp_align(sizeof(float) * (size_t)n_queries * KP_K * (size_t)cin);
    KpArgs a;
    a.q_pts = q_pts; a.s_pts = s_pts; a.inds = neighb_inds; a.nq = n_queries; a.ns = n_supports; a.h = (int)max_neighbors;
    a.x = features; a.cin = cin; a.kp = kernel_points;
    a.inv_extent = 1.0f / kp_extent; a.influence = kp_influence_mode;
    const float sigma = kp_extent * 0.3f;                       // kpconv.py:1122-1125 + radius_gaussian eps
    a.gauss_den = 2.0f * sigma * sigma + 1e-9f;
    a.wf = wf;
    a.off = offset_features; a.off_dim = offset_dim; a.extent = kp_extent;
    a.c_total = cin; a.c_off = 0;
    const KpOut ko = {weights, bias, act, slope, cout, out};
    if (offset_features) {
        // the deformed kernel points live in the MFMA aggregation only; the neighbour pruning of kpconv.py:1071-1103 drops
        // neighbours whose LINEAR influence is zero anyway -- with another influence function it would change the sum
        if (kp_influence_mode != 1 || !agg_mfma_ok(a)) return ML3D_E_UNSUPPORTED;
    } else if (max_neighbors > 0 && small_fused_ok(a, ko)) {
        switch (cin) {
            case 1: launch_small_fused<1>(a, ko, st); break;
            case 2: launch_small_fused<2>(a, ko, st); break;
            case 3: launch_small_fused<3>(a, ko, st); break;
            case 4: launch_small_fused<4>(a, ko, st); break;
            default: launch_small_fused<5>(a, ko, st); break;
        }
        return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
    }
    if (max_neighbors > 0 && agg_gemm32_ok(a, ko)) {
        launch_agg_gemm32(a, ko, st);
        return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
    }
    int rc = launch_weighted(a, st);
    if (rc) return rc;
    RowsA A;
    A.a = wf; A.lda = (int64_t)KP_K * cin; A.k1 = KP_K * cin;
    A.gather = nullptr; A.gather_stride = 0; A.a_rows = n_queries;
    A.a2 = nullptr; A.lda2 = 0; A.k2 = 0;
    A.gather_on_a2 = 0; A.g_rows_per_item = 0; A.g_src_rows_per_item = 0;
    Epilogue ep = {bias, nullptr, 0, act, slope, 0, 0, 0, 0};
    if (packed) {           // the contraction on the bf16 matrix pipe (gemm.h: three-way split of both operands); ineligible -> f32 below
        const int rcb = gemm_rows_bf16x3(wf, (int64_t)KP_K * cin, KP_K * cin, nullptr, 0, 0, n_queries, packed, cout, ep, out, cout, p,
                                         gemm_partial_bytes_bf16x3(n_queries, cout, KP_K * cin), st);
        if (rcb != ML3D_E_UNSUPPORTED) return rcb;
    }
    return gemm_rows(A, weights, n_queries, cout, KP_K * cin, ep, out, cout, p,
                     gemm_partial_bytes(n_queries, cout, KP_K * cin), st);
}

// ---- training side of the rigid KPConv ---------------------------------------------------------------
// out = wf . W with wf[q, k, c] = sum_h w[q, k, h] x[inds[q, h], c] (kpconv.py:1105-1159); the influences w depend on geometry
// only (kernel points are not trained).  Forward for autograd = the aggregation kernels above writing wf for the caller;
// backward: dW = wf^T . g and dwf = g . W^T are plain GEMMs (the caller's), the adjoint of the aggregation is a scatter:
//   dx[inds[q, h], c] += sum_k w[q, k, h] dwf[q, k, c]
// One wave per query, lane = channel (chunks of 64): the query's 15 x 64 slice of dwf sits in registers, lanes 0..14 compute the
// influences of a neighbour once, every lane forms its channel's sum over the kernel points from 15 broadcasts and adds it
// to the neighbour's row with one atomic.  (Summation order over the queries that share a neighbour is the hardware's:
// gradients are a float row, tolerance 1e-4.)
namespace ml3d {

__global__ void __launch_bounds__(256) kp_weighted_adjoint(KpArgs A, const float* __restrict__ dwf, float* __restrict__ dx) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= A.nq) return;
    const float qx = A.q_pts[3 * q], qy = A.q_pts[3 * q + 1], qz = A.q_pts[3 * q + 2];
    const int k = lane < KP_K ? lane : 0;
    const float kx = A.kp[3 * k], ky = A.kp[3 * k + 1], kz = A.kp[3 * k + 2];
    const int32_t* row = A.inds + q * A.h;
    for (int c0 = 0; c0 < A.cin; c0 += 64) {
        const int c = c0 + lane;
        const bool live = c < A.cin;
        float g[KP_K];
#pragma unroll
        for (int kk = 0; kk < KP_K; ++kk) g[kk] = live ? dwf[(q * KP_K + kk) * A.cin + c] : 0.f;
        for (int h = 0; h < A.h; ++h) {
            const int idx = row[h];
            if (idx < 0 || idx >= A.ns) continue;               // shadow neighbour (wave-uniform: the row is the wave's)
            const float* sp = A.s_pts + 3 * (int64_t)idx;
            const float dxk = (sp[0] - qx) - kx, dyk = (sp[1] - qy) - ky, dzk = (sp[2] - qz) - kz;
            float w = kp_influence(dxk * dxk + dyk * dyk + dzk * dzk, A);
            if (lane >= KP_K) w = 0.f;
            float v = 0.f;
#pragma unroll
            for (int kk = 0; kk < KP_K; ++kk) v = fmaf(__shfl(w, kk), g[kk], v);
            if (live) atomicAdd(dx + (int64_t)idx * A.cin + c, v);
        }
    }
}

}  // namespace ml3d

static int kp_geometry_args(KpArgs& a, const float* q_pts, const float* s_pts, const int32_t* neighb_inds, int64_t n_queries,
                            int64_t n_supports, int64_t max_neighbors, int cin, const float* kernel_points, int num_kernel_points,
                            float kp_extent, int kp_influence_mode) {
    if (n_queries < 0 || n_supports < 0 || max_neighbors < 0 || cin <= 0 || !(kp_extent > 0.f) || kp_influence_mode < 0 ||
        kp_influence_mode > 2 || max_neighbors > 0x7fffffff)
        return ML3D_E_INVALID;
    if (num_kernel_points != KP_K || cin > 512) return ML3D_E_UNSUPPORTED;
    if (n_queries > 0 && (!q_pts || !kernel_points || (max_neighbors > 0 && (!neighb_inds || !s_pts)))) return ML3D_E_INVALID;
    a.q_pts = q_pts; a.s_pts = s_pts; a.inds = neighb_inds; a.nq = n_queries; a.ns = n_supports; a.h = (int)max_neighbors;
    a.x = nullptr; a.cin = cin; a.kp = kernel_points;
    a.inv_extent = 1.0f / kp_extent; a.influence = kp_influence_mode;
    const float sigma = kp_extent * 0.3f;
    a.gauss_den = 2.0f * sigma * sigma + 1e-9f;
    a.wf = nullptr;
    a.off = nullptr; a.off_dim = 0; a.extent = kp_extent;
    a.c_total = cin; a.c_off = 0;
    return 0;
}

extern "C" int ml3d_kpconv_weighted(const float* q_pts, const float* s_pts, const int32_t* neighb_inds, int64_t n_queries,
                                    int64_t n_supports, int64_t max_neighbors, const float* features, int cin,
                                    const float* kernel_points, int num_kernel_points, float kp_extent, int kp_influence_mode,
                                    float* out_wf, void* stream) {
    KpArgs a;
    const int rc = kp_geometry_args(a, q_pts, s_pts, neighb_inds, n_queries, n_supports, max_neighbors, cin, kernel_points,
                                    num_kernel_points, kp_extent, kp_influence_mode);
    if (rc) return rc;
    if (n_queries == 0) return 0;
    if (!out_wf || (max_neighbors > 0 && !features)) return ML3D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (max_neighbors == 0)
        return hipMemsetAsync(out_wf, 0, sizeof(float) * (size_t)n_queries * KP_K * (size_t)cin, st) == hipSuccess ? 0 : ML3D_E_LAUNCH;
    a.x = features; a.wf = out_wf;
    return launch_weighted(a, st);
}

extern "C" int ml3d_kpconv_weighted_backward(const float* q_pts, const float* s_pts, const int32_t* neighb_inds,
                                             int64_t n_queries, int64_t n_supports, int64_t max_neighbors, int cin,
                                             const float* kernel_points, int num_kernel_points, float kp_extent,
                                             int kp_influence_mode, const float* grad_wf, float* grad_features, void* stream) {
    KpArgs a;
    const int rc = kp_geometry_args(a, q_pts, s_pts, neighb_inds, n_queries, n_supports, max_neighbors, cin, kernel_points,
                                    num_kernel_points, kp_extent, kp_influence_mode);
    if (rc) return rc;
    if (n_supports == 0) return 0;
    if (!grad_features) return ML3D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(grad_features, 0, sizeof(float) * (size_t)n_supports * (size_t)cin, st) != hipSuccess) return ML3D_E_LAUNCH;
    if (n_queries == 0 || max_neighbors == 0) return 0;
    if (!grad_wf) return ML3D_E_INVALID;
    hipLaunchKernelGGL(kp_weighted_adjoint, dim3((unsigned)((n_queries + 3) / 4)), dim3(256), 0, st, a, grad_wf, grad_features);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_kpconv_rigid(const float* q_pts, const float* s_pts, const int32_t* neighb_inds, int64_t n_queries,
                                 int64_t n_supports, int64_t max_neighbors, const float* features, int cin,
                                 const float* kernel_points, int num_kernel_points, float kp_extent, int kp_influence_mode,
                                 const float* weights, const float* bias, int act, float slope, int cout, float* out,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    return kpconv_run(q_pts, s_pts, neighb_inds, n_queries, n_supports, max_neighbors, features, cin, kernel_points,
                      num_kernel_points, kp_extent, kp_influence_mode, nullptr, 0, weights, bias, act, slope, cout, out,
                      workspace, workspace_bytes, stream);
}

// ml3d_kpconv_rigid with the [15 cin, cout] contraction on the bf16 matrix pipe: `packed` = ml3d_gemm_pack_bf16x3 of `weights`
// (both are passed: the fused small-channel kernels and ineligible shapes keep the float matrix)
extern "C" int ml3d_kpconv_rigid_bf16x3(const float* q_pts, const float* s_pts, const int32_t* neighb_inds, int64_t n_queries,
                                        int64_t n_supports, int64_t max_neighbors, const float* features, int cin,
                                        const float* kernel_points, int num_kernel_points, float kp_extent, int kp_influence_mode,
                                        const float* weights, const void* packed, const float* bias, int act, float slope,
                                        int cout, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!packed) return ML3D_E_INVALID;
    return kpconv_run(q_pts, s_pts, neighb_inds, n_queries, n_supports, max_neighbors, features, cin, kernel_points,
                      num_kernel_points, kp_extent, kp_influence_mode, nullptr, 0, weights, bias, act, slope, cout, out,
                      workspace, workspace_bytes, stream, packed);
}

extern "C" int ml3d_kpconv_deformable(const float* q_pts, const float* s_pts, const int32_t* neighb_inds,
                                      int64_t n_queries, int64_t n_supports, int64_t max_neighbors, const float* features,
                                      int cin, const float* kernel_points, int num_kernel_points, float kp_extent,
                                      int kp_influence_mode, const float* offset_features, int offset_dim,
                                      const float* weights, const float* bias, int act, float slope, int cout, float* out,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    if (!offset_features || (offset_dim != 3 * KP_K && offset_dim != 4 * KP_K)) return ML3D_E_INVALID;
    if (max_neighbors <= 0) return ML3D_E_INVALID;
    return kpconv_run(q_pts, s_pts, neighb_inds, n_queries, n_supports, max_neighbors, features, cin, kernel_points,
                      num_kernel_points, kp_extent, kp_influence_mode, offset_features, offset_dim, weights, bias, act, slope,
                      cout, out, workspace, workspace_bytes, stream);
}

extern "C" size_t ml3d_linear_workspace_bytes(int64_t m, int n, int k) {
    if (m < 0 || n <= 0 || k <= 0) return 0;
    return gemm_partial_bytes(m, n, k) + 512;
}

extern "C" int ml3d_linear(const float* a, int64_t lda, int k1, const int32_t* a_gather, int64_t a_gather_stride,
                           int64_t a_rows, const float* a2, int64_t lda2, int k2, const float* weights_t,
                           const float* bias, const float* residual, int64_t ldr, const int32_t* residual_gather,
                           int64_t residual_gather_stride, int64_t residual_rows, int act, float slope, float* out,
                           int64_t ldc, int64_t m, int n, void* workspace, size_t workspace_bytes, void* stream) {
    if (m < 0 || n <= 0 || k1 < 0 || k2 < 0 || k1 + k2 <= 0 || act < 0 || act > 2) return ML3D_E_INVALID;
    if (m == 0) return 0;
    if (!weights_t || !out || (k1 > 0 && !a) || (k2 > 0 && !a2) || lda < k1 || (k2 > 0 && lda2 < k2) || ldc < n ||
        (residual && ldr < n) || (residual_gather && (!residual || residual_gather_stride < 1 || residual_rows < 0)))
        return ML3D_E_INVALID;
    RowsA A;
    A.a = a; A.lda = lda; A.k1 = k1;
    A.gather = a_gather; A.gather_stride = a_gather_stride; A.a_rows = a_rows;
    A.a2 = a2; A.lda2 = lda2; A.k2 = k2;
    A.gather_on_a2 = 0; A.g_rows_per_item = 0; A.g_src_rows_per_item = 0;
    Epilogue ep = {bias, residual, ldr, act, slope, 0, 0, 0, 0};
    if (residual_gather) {      // global row indices: one "item" spanning every row
        ep.res_gather = residual_gather; ep.rg_rows_per_item = (int64_t)1 << 62; ep.rg_src_rows_per_item = 0;
        ep.rg_stride = residual_gather_stride; ep.rg_limit = residual_rows;
    }
    char* p = workspace ? (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255) : nullptr;
    size_t avail = workspace ? (workspace_bytes > 256 ? workspace_bytes - 256 : 0) : 0;
    return gemm_rows(A, weights_t, m, n, k1 + k2, ep, out, ldc, p, avail, (hipStream_t)stream);
}

extern "C" int ml3d_gather_pool(const float* features, int64_t n_supports, int channels, const int32_t* inds,
                                int64_t n_queries, int64_t max_neighbors, int mode, float* out, void* stream) {
    if (n_supports < 0 || channels <= 0 || n_queries < 0 || max_neighbors <= 0 || (mode != 0 && mode != 1) ||
        max_neighbors > 0x7fffffff)
        return ML3D_E_INVALID;
    if (n_queries == 0) return 0;
    if (!features || !inds || !out) return ML3D_E_INVALID;
    if ((channels & 3) == 0 && (((uintptr_t)features | (uintptr_t)out) & 15) == 0 &&
        n_supports * (int64_t)(channels / 4) < ((int64_t)1 << 32)) {
        const int64_t total = n_queries * (channels / 4);
        unsigned nb = (unsigned)((total + 255) / 256 < 262144 ? (total + 255) / 256 : 262144);
        hipLaunchKernelGGL(gather_pool_v4, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float4*)features, n_supports,
                           channels / 4, inds, n_queries, (int)max_neighbors, mode, (float4*)out);
        return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
    }
    int64_t total = n_queries * channels;
    unsigned nb = (unsigned)((total + 255) / 256 < 65535 ? (total + 255) / 256 : 65535);
    hipLaunchKernelGGL(gather_pool_k, dim3(nb), dim3(256), 0, (hipStream_t)stream, features, n_supports, channels, inds,
                       n_queries, (int)max_neighbors, mode, out);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}
