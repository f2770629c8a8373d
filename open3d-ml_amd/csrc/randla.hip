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

#include "ml3d_hip.h"

namespace ml3d {

constexpr int RK = 16;        // neighbours per point (num_neighbors in every reference config)
constexpr int XROW = 20;      // LDS row pitch of the K-slab: 16 + 4 pad floats keeps b128 reads conflict-free
constexpr int LFA_THREADS = 256;

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

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
    float* out;                          // [m][cout]
    int64_t m_total;
    int cout;
    int act;                             // 0 none, 1 leaky relu
    float slope;
};

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
        acc[r] = A.bias[o];
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
};

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
    float v = -3.0e38f;
#pragma unroll
    for (int k = 0; k < RK; ++k) v = fmaxf(v, feat[(b * n_in + id[k]) * c + ch]);
    out[r * c + ch] = v;
}

struct Tracer {
    const ml3d_trace* t;
    hipStream_t st;
    void begin(int tag) const { if (t && t->tag == tag && t->ev_start) (void)hipEventRecord((hipEvent_t)t->ev_start, st); }
    void end(int tag) const { if (t && t->tag == tag && t->ev_stop) (void)hipEventRecord((hipEvent_t)t->ev_stop, st); }
};

static int launch_linear(const LinArgs& a, hipStream_t st) {
    if (a.m_total <= 0) return 0;
    int64_t items = ((a.m_total + LIN_RM - 1) / LIN_RM) * a.cout;
    hipLaunchKernelGGL(linear_act, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
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
        push(dd * 2 * dd); push(2 * dd);
        push((int64_t)d_in * 2 * dd); push(2 * dd);
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

static size_t fwd_ws_floats(const ml3d_randla_desc* d) {
    int64_t n[ML3D_RANDLA_MAX_LAYERS + 1];
    n[0] = d->num_points;
    for (int l = 0; l < d->num_layers; ++l) n[l + 1] = n[l] / d->sub_sampling_ratio[l];
    auto al = [](int64_t x) { return (x + 63) & ~(int64_t)63; };
    int64_t B = d->batch, f = 0;
    f += al(B * n[0] * d->dim_features);
    for (int l = 0; l < d->num_layers; ++l) {
        int64_t dd = d->dim_output[l], h = dd / 2;
        f += 2 * al(B * n[l] * h) + al(B * n[l] * 2 * dd) + al(B * n[l + 1] * 2 * dd);
    }
    int64_t Dm = 2 * d->dim_output[d->num_layers - 1];
    f += al(B * n[d->num_layers] * Dm);
    int ed[ML3D_RANDLA_MAX_LAYERS + 1];
    enc_dims(d, ed);
    for (int i = 0; i < d->num_layers; ++i) f += al(B * n[d->num_layers - 1 - i] * ed[d->num_layers + 1 - i - 2]);
    f += al(B * n[0] * 64) + al(B * n[0] * 32);
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
    if (!desc_ok(d) || !params || !features || !points || !neighbor_idx || !interp_idx || !out_scores)
        return ML3D_E_INVALID;
    if (d->num_neighbors != RK) return ML3D_E_UNSUPPORTED;
    if (workspace_bytes < ml3d_randla_forward_workspace_bytes(d)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const Tracer T = {trace, st};
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

    // fc0 + bn0 + lrelu(0.2)            (randlanet.py:266-271)
    float* feat = take(B * n[0] * d->dim_features);
    {
        LinArgs a = {};
        a.a0 = features; a.c0 = d->in_channels; a.wt = P(0); a.bias = P(1); a.out = feat;
        a.m_total = B * n[0]; a.cout = d->dim_features; a.act = 1; a.slope = 0.2f;
        T.begin(1000); int rc = launch_linear(a, st); T.end(1000); if (rc) return rc;
    }
    int d_in = d->dim_features;
    float* enc_keep[ML3D_RANDLA_MAX_LAYERS + 1];   // encoder_feat_list (randlanet.py:274-283)
    for (int l = 0; l < Lr; ++l) {
        const int dd = d->dim_output[l], h = dd / 2, sb = 2 + 18 * l;
        const int64_t M = B * n[l];
        float* f1 = take(M * h);
        float* p1 = take(M * h);
        float* enc = take(M * 2 * dd);
        float* samp = take(B * n[l + 1] * 2 * dd);
        {   // mlp1: SharedMLP(d_in, d/2) lrelu 0.2   (randlanet.py:680)
            LinArgs a = {};
            a.a0 = feat; a.c0 = d_in; a.wt = P(sb + 0); a.bias = P(sb + 1); a.out = f1;
            a.m_total = M; a.cout = h; a.act = 1; a.slope = 0.2f;
            T.begin(8 * l); int rc = launch_linear(a, st); T.end(8 * l); if (rc) return rc;
        }
        LfaArgs s1 = {};
        s1.xyz = points; s1.nidx = neighbor_idx[l]; s1.n = n[l]; s1.n0 = n[0]; s1.m_total = M;
        s1.gfeat = f1; s1.lse1_wt = P(sb + 2); s1.lse1_b = P(sb + 3);
        s1.score_wt = P(sb + 4); s1.score_b = P(sb + 5); s1.pool_wt = P(sb + 6); s1.pool_b = P(sb + 7);
        s1.d_in = d_in; s1.out = p1;
        LfaArgs s2 = s1;
        s2.gfeat = p1; s2.lse2_wt = P(sb + 8); s2.lse2_b = P(sb + 9);
        s2.score_wt = P(sb + 10); s2.score_b = P(sb + 11); s2.pool_wt = P(sb + 12); s2.pool_b = P(sb + 13);
        s2.mlp2_wt = P(sb + 14); s2.mlp2_b = P(sb + 15); s2.short_wt = P(sb + 16); s2.short_b = P(sb + 17);
        s2.feat_in = feat; s2.out = enc;
        int rc;
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
        if (rc) return rc;
        {   // random_sample onto the kept prefix      (randlanet.py:278)
            int64_t items = B * n[l + 1] * 2 * dd;
            T.begin(8 * l + 3);
            hipLaunchKernelGGL(gather_max, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, enc,
                               neighbor_idx[l], samp, n[l], n[l + 1], B, 2 * dd);
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
        LinArgs a = {};
        a.a0 = feat; a.c0 = Dm; a.wt = P(slot); a.bias = P(slot + 1); a.out = cur;
        a.m_total = B * n[Lr]; a.cout = Dm; a.act = 1; a.slope = 0.2f;
        T.begin(1001); int rc = launch_linear(a, st); T.end(1001); if (rc) return rc;
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
        LinArgs a = {};
        a.a0 = enc_keep[Lr + 1 - i - 2]; a.c0 = skip_c;
        a.a1 = cur; a.c1 = cprev; a.gather = interp_idx[lev];
        a.rows_per_item = n[lev]; a.a1_rows_per_item = n[lev + 1];
        a.wt = P(slot); a.bias = P(slot + 1); a.out = outp;
        a.m_total = B * n[lev]; a.cout = skip_c; a.act = 1; a.slope = 0.2f;
        T.begin(1100 + i); int rc = launch_linear(a, st); T.end(1100 + i); if (rc) return rc;
        slot += 2;
        cur = outp;
        cprev = skip_c;
    }
    float* t0 = take(B * n[0] * 64);
    float* t1 = take(B * n[0] * 32);
    {   // fc1                                            (randlanet.py:93-96, 296)
        LinArgs a = {};
        a.a0 = cur; a.c0 = cprev; a.wt = P(slot); a.bias = P(slot + 1); a.out = t0;
        a.m_total = B * n[0]; a.cout = 64; a.act = 1; a.slope = 0.2f;
        T.begin(1200); int rc = launch_linear(a, st); T.end(1200); if (rc) return rc;
        a.a0 = t0; a.c0 = 64; a.wt = P(slot + 2); a.bias = P(slot + 3); a.out = t1; a.cout = 32;
        T.begin(1201); rc = launch_linear(a, st); T.end(1201); if (rc) return rc;
        a.a0 = t1; a.c0 = 32; a.wt = P(slot + 4); a.bias = P(slot + 5); a.out = out_scores;
        a.cout = d->num_classes; a.act = 0;
        T.begin(1202); rc = launch_linear(a, st); T.end(1202); if (rc) return rc;
    }
    return 0;
}
