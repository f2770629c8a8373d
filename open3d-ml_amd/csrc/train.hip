// train.hip — the training side of the hot path on hand-written HIP, forward AND backward:
//   * ml3d_gemm_tn                      C = A^T B over the rows (weight gradients of every Linear / 1x1 convolution and of KPConv's
//                                       [15 cin, cout] contraction: dW = x^T dy), f32 MFMA straight from global rows, split over rows
//   * ml3d_batchnorm_train_*            BatchNorm on the batch statistics (+ LeakyReLU), forward and backward
//                                       (SharedMLP.forward randlanet.py:503-518, BatchNormBlock.forward kpconv.py:1238-1249)
//   * ml3d_gather_rows / ml3d_scatter_add_rows / ml3d_gather_pool_backward
//                                       nearest_interpolation (randlanet.py:329-350), closest_pool / max_pool (kpconv.py:821-858)
//   * ml3d_randla_attention_stage[_backward]
//                                       gather + concat + score Linear + softmax over K + weighted sum of one attentive pooling
//                                       (randlanet.py:596-605, 622-637) as ONE kernel each way: the [B, N, K, d] tensors of the
//                                       reference formulation (gathered features, scores, probabilities and their three gradients)
//                                       are never materialised -- only the encoded relative positions [B, N, K, d/2], which the
//                                       reference keeps as well.
// Everything here is float32; the products run on the f32 MFMA (v_mfma_f32_32x32x2_f32).  Reductions across workgroups use float
// atomics (the order of the partial sums varies run to run at the 1e-7 level, like every GPU training backward).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <gfx950_ops.h>
#include "grid.h"
#include "ml3d_hip.h"

namespace ml3d {

typedef float tr_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ tr_f32x16 tr_zero16() {
    tr_f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
// row of element r of a 32 x 32 MFMA result held by lane (hi, cl): (r & 3) + 8 (r >> 2) + 4 hi; its column is cl
__device__ __forceinline__ int tr_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---------------------------------------------------------------------------------------------------------------------
// C[i, j] += sum_r A[r, i] B[r, j]   (A [m, lda] uses k columns, B [m, ldb] uses n columns, C [k, ldc] zeroed by the host call)
// One wave per (64 x 64 tile of C, slice of the rows): lane (hi, cl) feeds A[r + hi][i0 + cl] and B[r + hi][j0 + cl] -- both
// coalesced 128-byte row segments -- into four 32 x 32 x 2 MFMAs per two rows; no LDS, no barrier.  The slices meet in C through
// float atomics.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gemm_tn_k(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb, int64_t m, int k, int n,
          int64_t rows_per_slice, int tiles_j, int64_t n_units, float* __restrict__ c, int64_t ldc) {
    const int lane = threadIdx.x & 63, hi = lane >> 5, cl = lane & 31;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (unit >= n_units) return;                                     // (wave-uniform)
    const int tiles_i = (k + 63) / 64;
    const int64_t ntiles = (int64_t)tiles_i * tiles_j;
    const int64_t slice = unit / ntiles;
    const int tile = (int)(unit - slice * ntiles);
    const int i0 = (tile / tiles_j) * 64, j0 = (tile % tiles_j) * 64;
    const int64_t r0 = slice * rows_per_slice;
    const int64_t r1 = r0 + rows_per_slice < m ? r0 + rows_per_slice : m;
    const bool ia0 = i0 + cl < k, ia1 = i0 + 32 + cl < k, jb0 = j0 + cl < n, jb1 = j0 + 32 + cl < n;
    const float* ap = a + i0 + cl;
    const float* bp = b + j0 + cl;
    tr_f32x16 acc00 = tr_zero16(), acc01 = tr_zero16(), acc10 = tr_zero16(), acc11 = tr_zero16();
#pragma unroll 4
    for (int64_t r = r0; r < r1; r += 2) {
        const int64_t rr = r + hi;
        const bool ok = rr < r1;
        const float a0 = ok && ia0 ? ap[rr * lda] : 0.f, a1 = ok && ia1 ? ap[rr * lda + 32] : 0.f;
        const float b0 = ok && jb0 ? bp[rr * ldb] : 0.f, b1 = ok && jb1 ? bp[rr * ldb + 32] : 0.f;
        acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + tr_row(r, hi);
        if (i < k) {
            if (jb0) atomicAdd(c + (int64_t)i * ldc + j0 + cl, acc00[r]);
            if (jb1) atomicAdd(c + (int64_t)i * ldc + j0 + 32 + cl, acc01[r]);
        }
        if (i + 32 < k) {
            if (jb0) atomicAdd(c + (int64_t)(i + 32) * ldc + j0 + cl, acc10[r]);
            if (jb1) atomicAdd(c + (int64_t)(i + 32) * ldc + j0 + 32 + cl, acc11[r]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Per-column reductions over the rows of a [m, c] matrix (row stride ld).  A workgroup takes a slice of the rows; with c <= 256 it
// is cut into 256 / c row groups of c threads (thread = one column of one group: coalesced row segments) that meet in LDS, one
// atomic per (workgroup, column) leaves.  MODE 0: sum x (float, bias gradients); MODE 1: sum x, sum x^2 in double (BatchNorm
// statistics); MODE 2: sum g', sum g' xhat in double with g' = gy * act'(y), xhat = (x - mean) invstd (BatchNorm backward).
// ---------------------------------------------------------------------------------------------------------------------
struct BnRef { const float* y; const float* gy; const float* mean; const float* invstd; int act; float slope; };

__device__ __forceinline__ float act_grad(float gy, float y, int act, float slope) {
    return act == 0 ? gy : (y > 0.f ? gy : gy * slope);
}

template <int MODE>
__global__ void __launch_bounds__(256)
col_reduce_k(const float* __restrict__ x, int64_t ld, int64_t m, int c, int64_t rows_per_block, BnRef R, float* __restrict__ out_f,
             double* __restrict__ out_d) {
    __shared__ double red[2][256];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < m ? r0 + rows_per_block : m;
    const int t = threadIdx.x;
    const int cols_per = c <= 256 ? c : 256;
    const int groups = 256 / cols_per;
    const int g = t / cols_per, col0 = t - g * cols_per;
    const bool live = g < groups;
    for (int cb = 0; cb < c; cb += cols_per) {                        // (one pass when c <= 256)
        const int col = cb + col0;
        double s0 = 0.0, s1 = 0.0;
        float f0 = 0.f;
        if (live && col < c) {
            float mu = 0.f, is = 0.f;
            if (MODE == 2) { mu = R.mean[col]; is = R.invstd[col]; }
            for (int64_t r = r0 + g; r < r1; r += groups) {
                const float v = x[r * ld + col];
                if (MODE == 0) f0 += v;
                if (MODE == 1) { s0 += (double)v; s1 += (double)v * (double)v; }
                if (MODE == 2) {
                    const float gp = act_grad(R.gy[r * (int64_t)c + col], R.y[r * (int64_t)c + col], R.act, R.slope);
                    s0 += (double)gp;
                    s1 += (double)gp * (double)((v - mu) * is);
                }
            }
        }
        if (MODE == 0) s0 = (double)f0;
        red[0][t] = s0;
        red[1][t] = s1;
        __syncthreads();
        if (t < cols_per && cb + t < c) {
            double a0 = 0.0, a1 = 0.0;
            for (int q = 0; q < groups; ++q) { a0 += red[0][q * cols_per + t]; a1 += red[1][q * cols_per + t]; }
            if (MODE == 0) atomicAdd(out_f + cb + t, (float)a0);
            else { atomicAdd(out_d + cb + t, a0); atomicAdd(out_d + c + cb + t, a1); }
        }
        __syncthreads();
    }
}

// mean / biased variance / 1 / sqrt(var + eps) from the double sums
__global__ void bn_finalize_k(const double* __restrict__ sums, int64_t m, int c, float eps, float* __restrict__ mean,
                              float* __restrict__ var, float* __restrict__ invstd) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const double mu = sums[ch] / (double)m;
    double v = sums[c + ch] / (double)m - mu * mu;
    if (v < 0.0) v = 0.0;
    mean[ch] = (float)mu;
    var[ch] = (float)v;
    invstd[ch] = (float)(1.0 / sqrt(v + (double)eps));
}

__global__ void __launch_bounds__(256)
bn_apply_k(const float* __restrict__ x, int64_t total, int c, const float* __restrict__ gamma, const float* __restrict__ beta,
           const float* __restrict__ mean, const float* __restrict__ invstd, int act, float slope, float* __restrict__ y) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int ch = (int)(e % c);
        float v = (x[e] - mean[ch]) * invstd[ch];
        v = v * (gamma ? gamma[ch] : 1.f) + (beta ? beta[ch] : 0.f);
        y[e] = act == 0 ? v : (v > 0.f ? v : v * slope);
    }
}

// gx = gamma invstd (g' - mean(g') - xhat mean(g' xhat));  ggamma = sum g' xhat, gbeta = sum g' (written by the first threads)
__global__ void __launch_bounds__(256)
bn_backward_apply_k(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy, int64_t m, int c,
                    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                    const double* __restrict__ sums, int act, float slope, float* __restrict__ gx, float* __restrict__ ggamma,
                    float* __restrict__ gbeta) {
    const int64_t total = m * (int64_t)c;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (first < c) {
        if (ggamma) ggamma[first] = (float)sums[c + first];
        if (gbeta) gbeta[first] = (float)sums[first];
    }
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = first; e < total; e += step) {
        const int ch = (int)(e % c);
        const float is = invstd[ch];
        const float xh = (x[e] - mean[ch]) * is;
        const float gp = act_grad(gy[e], y[e], act, slope);
        const float m0 = (float)(sums[ch] / (double)m), m1 = (float)(sums[c + ch] / (double)m);
        gx[e] = (gamma ? gamma[ch] : 1.f) * is * (gp - m0 - xh * m1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// row gathers and their adjoints
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
gather_rows_k(const float* __restrict__ x, int64_t n_src, int c, const int32_t* __restrict__ idx, int64_t idx_stride, int64_t m,
              float* __restrict__ out) {
    const int64_t total = m * (int64_t)c, step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int64_t r = e / c;
        const int ch = (int)(e - r * c);
        const int64_t s = idx[r * idx_stride];
        out[e] = (s >= 0 && s < n_src) ? x[s * c + ch] : 0.f;           // (a shadow index reads the zero row)
    }
}

__global__ void __launch_bounds__(256)
scatter_add_rows_k(const float* __restrict__ g, int64_t n_src, int c, const int32_t* __restrict__ idx, int64_t idx_stride, int64_t m,
                   float* __restrict__ gx) {
    const int64_t total = m * (int64_t)c, step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int64_t r = e / c;
        const int ch = (int)(e - r * c);
        const int64_t s = idx[r * idx_stride];
        if (s >= 0 && s < n_src) atomicAdd(gx + s * c + ch, g[e]);
    }
}

// max_pool's adjoint (kpconv.py:841-858: max over the listed neighbours of the features padded with one zero row): the gradient of
// (q, ch) goes to the FIRST maximal neighbour in list order (torch.max's CPU rule); if that is the shadow row it is dropped
__global__ void __launch_bounds__(256)
max_pool_adjoint_k(const float* __restrict__ feat, int64_t ns, int c, const int32_t* __restrict__ inds, int64_t nq, int64_t H,
                   const float* __restrict__ g, float* __restrict__ gfeat) {
    const int64_t total = nq * (int64_t)c, step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int64_t q = e / c;
        const int ch = (int)(e - q * c);
        int64_t arg = -1;
        float best = 0.f;
        for (int64_t h = 0; h < H; ++h) {
            const int64_t s = inds[q * H + h];
            const bool real = s >= 0 && s < ns;
            const float v = real ? feat[s * c + ch] : 0.f;
            if (h == 0 || v > best) { best = v; arg = real ? s : -1; }
        }
        if (arg >= 0) atomicAdd(gfeat + arg * c + ch, g[e]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The attention stage of RandLA-Net's LocalFeatureAggregation, training form.  Per point p (K = 16 neighbours, d = c1 + c2):
//   x[k, :]  = [ f[idx[p, k], :c1] | enc[p, k, :c2] ]                      (LocalSpatialEncoding's concat, randlanet.py:596-605)
//   s[k, :]  = x[k, :] W^T + bias                                           (score_fn's Linear, randlanet.py:617)
//   out[c]   = sum_k softmax_k(s[:, c])[k] x[k, c]                          (randlanet.py:631-637)
// A workgroup stages the x rows of TP points (R = 16 TP rows, all d columns) in LDS once, then walks the output columns in blocks
// of 64: scores of the block on the MFMA (A from LDS, W^T rows from global / L2, coalesced), softmax + weighted sum with one thread
// per (point, column).  Backward recomputes x, s and the probabilities p, forms gs[k, c] = g[c] p[k, c] (x[k, c] - out[c]) in
// place of s, and takes
//   gx = gs W (+ g p on the block's own columns)   -> scattered with atomics into grad_f (through idx) and grad_enc
//   gW[c, :] += gs[:, c]^T x                        -> 64 x d accumulator tiles in registers across ALL tiles of the workgroup,
//                                                      one atomic pass per workgroup at the end
//   gbias[c] += sum_k gs[k, c]                      (zero up to rounding: softmax is shift invariant; kept for fidelity)
// Workgroup (g, cb) owns column block cb of a persistent share g of the tiles, so the gW tile stays in registers.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int AS_K = 16;                 // neighbours per point (num_neighbors of every in-scope configuration)
constexpr int AS_CB = 64;                // score columns per pass
constexpr int AS_DMAX = 256;             // widest stage (d = dim_output[l]); wider stages stay on the unfused path
constexpr int AS_LDS_FLOATS = 64 * (128 + 1) + 64 * (AS_CB + 1);      // >= 32 * (256 + 1) + 32 * (AS_CB + 1)
constexpr int AS_MAXT = 4;               // 32 x 32 accumulator tiles per wave

struct AttnArgs {
    const float* f; const float* enc; const int32_t* idx; const float* w; const float* wt; const float* bias;
    int64_t batch, n; int c1, c2;
    float* out;                                       // forward result / saved forward result
    const float* gout; float* gf; float* genc; float* gw; float* gbias;
};

// stage the R x d rows of tile `tile` (points tile * TP ...) into X (row pitch ldx); rows past the last point are zeros
template <int R>
__device__ __forceinline__ void attn_stage_x(const AttnArgs& A, int64_t tile, float* X, int ldx) {
    constexpr int TP = R / AS_K;
    const int d = A.c1 + A.c2;
    const int64_t npts = A.batch * A.n;
    for (int e = threadIdx.x; e < R * d; e += 256) {
        const int row = e / d, col = e - row * d;
        const int64_t pt = tile * TP + row / AS_K;
        float v = 0.f;
        if (pt < npts) {
            const int kk = row % AS_K;
            if (col < A.c1) {
                const int64_t b = pt / A.n;
                const int64_t src = A.idx[pt * AS_K + kk];
                if (src >= 0 && src < A.n) v = A.f[(b * A.n + src) * A.c1 + col];
            } else {
                v = A.enc[(pt * AS_K + kk) * A.c2 + (col - A.c1)];
            }
        }
        X[row * ldx + col] = v;
    }
}

// S[R x 64] <- X[R x d] . Wt[:, cb0 .. cb0 + 63] + bias   (wave tiles: R = 64: 2 x 2, R = 32: 1 x 2 on waves 0, 1)
template <int R>
__device__ __forceinline__ void attn_scores(const AttnArgs& A, const float* X, int ldx, int cb0, float* S) {
    const int d = A.c1 + A.c2;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hi = lane >> 5, cl = lane & 31;
    const int rt = R == 64 ? (wave & 1) : 0, ct = R == 64 ? (wave >> 1) : wave;
    if (ct < 2) {                                                     // (wave-uniform)
        const int col = cb0 + ct * 32 + cl;
        const bool cok = col < d;
        const float* xr = X + (rt * 32 + cl) * ldx + hi;
        const float* wp = A.wt + (int64_t)hi * d + col;
        tr_f32x16 acc = tr_zero16();
        for (int kk = 0; kk < d; kk += 2) {
            const float av = xr[kk];
            const float bv = cok ? wp[(int64_t)kk * d] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        const float bs = (cok && A.bias) ? A.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[(rt * 32 + tr_row(r, hi)) * (AS_CB + 1) + ct * 32 + cl] = acc[r] + bs;
    }
}

template <int R>
__global__ void __launch_bounds__(256)
attn_stage_fwd_k(AttnArgs A, int64_t n_tiles) {
    constexpr int TP = R / AS_K;
    __shared__ float lds[AS_LDS_FLOATS];
    const int d = A.c1 + A.c2, ldx = d | 1;
    float* X = lds;
    float* S = lds + R * ldx;
    const int64_t npts = A.batch * A.n;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        attn_stage_x<R>(A, tile, X, ldx);
        __syncthreads();
        for (int cb0 = 0; cb0 < d; cb0 += AS_CB) {
            attn_scores<R>(A, X, ldx, cb0, S);
            __syncthreads();
            const int t = threadIdx.x;
            if (t < TP * AS_CB) {
                const int tp = t / AS_CB, c = t - tp * AS_CB, col = cb0 + c;
                const int64_t pt = tile * TP + tp;
                if (col < d && pt < npts) {
                    float sv[AS_K], mx = -3.0e38f;
#pragma unroll
                    for (int k = 0; k < AS_K; ++k) { sv[k] = S[(tp * AS_K + k) * (AS_CB + 1) + c]; mx = fmaxf(mx, sv[k]); }
                    float sum = 0.f;
#pragma unroll
                    for (int k = 0; k < AS_K; ++k) { sv[k] = expf(sv[k] - mx); sum += sv[k]; }
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < AS_K; ++k) acc = fmaf(sv[k] / sum, X[(tp * AS_K + k) * ldx + col], acc);
                    A.out[pt * d + col] = acc;
                }
            }
            __syncthreads();
        }
    }
}

template <int R>
__global__ void __launch_bounds__(256)
attn_stage_bwd_k(AttnArgs A, int64_t n_tiles, int groups) {
    constexpr int TP = R / AS_K;
    __shared__ float lds[AS_LDS_FLOATS];
    const int d = A.c1 + A.c2, ldx = d | 1;
    float* X = lds;
    float* S = lds + R * ldx;
    const int64_t npts = A.batch * A.n;
    const int n_cb = (d + AS_CB - 1) / AS_CB;
    const int cbi = blockIdx.x % n_cb, grp = blockIdx.x / n_cb;
    const int cb0 = cbi * AS_CB;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hi = lane >> 5, cl = lane & 31;
    const int t = threadIdx.x;
    const int n_jt = (d + 31) / 32;                                   // 32-column tiles across d
    // gW block [64 x d]: tiles (it in 0..1, jt in 0..n_jt-1), tile q = it * n_jt + jt on wave q % 4, slot q / 4
    tr_f32x16 gw[AS_MAXT];
#pragma unroll
    for (int q = 0; q < AS_MAXT; ++q) gw[q] = tr_zero16();
    float gb = 0.f;                                                   // thread t < TP * 64 owns score column cb0 + t % 64
    for (int64_t tile = grp; tile < n_tiles; tile += groups) {
        attn_stage_x<R>(A, tile, X, ldx);
        __syncthreads();
        attn_scores<R>(A, X, ldx, cb0, S);
        __syncthreads();
        if (t < TP * AS_CB) {
            const int tp = t / AS_CB, c = t - tp * AS_CB, col = cb0 + c;
            const int64_t pt = tile * TP + tp;
            const bool on = col < d && pt < npts;
            float sv[AS_K], mx = -3.0e38f;
#pragma unroll
            for (int k = 0; k < AS_K; ++k) { sv[k] = S[(tp * AS_K + k) * (AS_CB + 1) + c]; mx = fmaxf(mx, sv[k]); }
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < AS_K; ++k) { sv[k] = expf(sv[k] - mx); sum += sv[k]; }
            const float g = on ? A.gout[pt * d + col] : 0.f, o = on ? A.out[pt * d + col] : 0.f;
            const int64_t b = on ? pt / A.n : 0;
#pragma unroll
            for (int k = 0; k < AS_K; ++k) {
                const float pg = on ? sv[k] / sum * g : 0.f;
                const float gs = pg * (X[(tp * AS_K + k) * ldx + (on ? col : 0)] - o);
                S[(tp * AS_K + k) * (AS_CB + 1) + c] = on ? gs : 0.f;
                gb += on ? gs : 0.f;
                if (on) {                                              // the direct term d out / d x = p g on the block's own columns
                    if (col < A.c1) {
                        const int64_t src = A.idx[pt * AS_K + k];
                        if (src >= 0 && src < A.n) atomicAdd(A.gf + (b * A.n + src) * A.c1 + col, pg);
                    } else {
                        atomicAdd(A.genc + (pt * AS_K + k) * A.c2 + (col - A.c1), pg);
                    }
                }
            }
        }
        __syncthreads();
        // ---- gW[cb0 + i, j] += sum_rows gs[row, i] x[row, j]
#pragma unroll
        for (int q = 0; q < AS_MAXT; ++q) {
            const int tq = q * 4 + wave;
            if (tq < 2 * n_jt) {                                      // (wave-uniform)
                const int it = tq / n_jt, jt = tq - it * n_jt;
                const bool jok = jt * 32 + cl < d;
                const float* sp = S + hi * (AS_CB + 1) + it * 32 + cl;
                const float* xp = X + hi * ldx + (jok ? jt * 32 + cl : 0);
                tr_f32x16 acc = gw[q];
                for (int r = 0; r < R; r += 2) {
                    const float av = sp[r * (AS_CB + 1)];
                    const float bv = jok ? xp[r * ldx] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
                }
                gw[q] = acc;
            }
        }
        // ---- gx[row, j] = sum_c gs[row, c] W[cb0 + c, j]  -> grad_f (through idx) / grad_enc, atomics
        constexpr int RT = R / 32;
#pragma unroll
        for (int q = 0; q < AS_MAXT; ++q) {
            const int tq = q * 4 + wave;
            if (tq < RT * n_jt) {                                     // (wave-uniform)
                const int rt = tq / n_jt, jt = tq - rt * n_jt;
                const int j = jt * 32 + cl;
                const bool jok = j < d;
                const float* sp = S + (rt * 32 + cl) * (AS_CB + 1) + hi;
                tr_f32x16 acc = tr_zero16();
                for (int cc = 0; cc < AS_CB; cc += 2) {
                    const float av = sp[cc];
                    const bool wok = jok && cb0 + cc + hi < d;
                    const float bv = wok ? A.w[(int64_t)(cb0 + cc + hi) * d + j] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
                }
                if (jok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rt * 32 + tr_row(r, hi);
                        const int64_t pt = tile * TP + row / AS_K;
                        if (pt < npts) {
                            const int kk = row % AS_K;
                            if (j < A.c1) {
                                const int64_t src = A.idx[pt * AS_K + kk];
                                if (src >= 0 && src < A.n) atomicAdd(A.gf + ((pt / A.n) * A.n + src) * A.c1 + j, acc[r]);
                            } else {
                                atomicAdd(A.genc + (pt * AS_K + kk) * A.c2 + (j - A.c1), acc[r]);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- the workgroup's share of gW / gbias
#pragma unroll
    for (int q = 0; q < AS_MAXT; ++q) {
        const int tq = q * 4 + wave;
        if (tq < 2 * n_jt) {
            const int it = tq / n_jt, jt = tq - it * n_jt;
            const int j = jt * 32 + cl;
            if (j < d) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = cb0 + it * 32 + tr_row(r, hi);
                    if (i < d) atomicAdd(A.gw + (int64_t)i * d + j, gw[q][r]);
                }
            }
        }
    }
    if (A.gbias && t < TP * AS_CB && cb0 + t % AS_CB < d) atomicAdd(A.gbias + cb0 + t % AS_CB, gb);
}

static inline unsigned tr_blocks(int64_t total, int per, unsigned cap) {
    int64_t b = (total + per - 1) / per;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace ml3d

using namespace ml3d;

extern "C" int ml3d_gemm_tn(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m, int k, int n, float* c, int64_t ldc,
                            float* col_sums_a, void* stream) {
    if (m < 0 || k <= 0 || n <= 0 || lda < k || ldb < n || ldc < n || !c) return ML3D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (ldc == n) zero_async(c, sizeof(float) * (size_t)k * (size_t)n, st);
    else for (int i = 0; i < k; ++i) zero_async(c + (int64_t)i * ldc, sizeof(float) * (size_t)n, st);
    if (col_sums_a) zero_async(col_sums_a, sizeof(float) * (size_t)k, st);
    if (m == 0) return 0;
    if (!a || !b) return ML3D_E_INVALID;
    const int tiles_i = (k + 63) / 64, tiles_j = (n + 63) / 64;
    const int64_t ntiles = (int64_t)tiles_i * tiles_j;
    int64_t slices = (4096 + ntiles - 1) / ntiles;                      // ~4096 waves in flight
    const int64_t max_slices = (m + 63) / 64;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    int64_t rps = (m + slices - 1) / slices;
    rps = (rps + 1) & ~(int64_t)1;
    slices = (m + rps - 1) / rps;
    const int64_t units = slices * ntiles;
    hipLaunchKernelGGL(gemm_tn_k, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, a, lda, b, ldb, m, k, n, rps, tiles_j, units, c, ldc);
    if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    if (col_sums_a) {
        const unsigned nb = tr_blocks(m, 256, 512);
        const int64_t rpb = (m + nb - 1) / nb;
        BnRef none = {};
        hipLaunchKernelGGL((col_reduce_k<0>), dim3(nb), dim3(256), 0, st, a, lda, m, k, rpb, none, col_sums_a, (double*)nullptr);
        if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    }
    return 0;
}

extern "C" size_t ml3d_batchnorm_train_workspace_bytes(int channels) {
    return channels > 0 ? sizeof(double) * 2 * (size_t)channels + 256 : 0;
}

static double* bn_ws(void* workspace) { return (double*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255); }

extern "C" int ml3d_batchnorm_train_forward(const float* x, int64_t rows, int channels, const float* gamma, const float* beta, float eps,
                                            int act, float slope, float* y, float* save_mean, float* save_var, float* save_invstd,
                                            void* workspace, size_t workspace_bytes, void* stream) {
    if (rows <= 0 || channels <= 0 || (act != 0 && act != 1)) return ML3D_E_INVALID;
    if (!x || !y || !save_mean || !save_var || !save_invstd || !workspace) return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_batchnorm_train_workspace_bytes(channels)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* sums = bn_ws(workspace);
    zero_async(sums, sizeof(double) * 2 * (size_t)channels, st);
    const unsigned nb = tr_blocks(rows, 256, 512);
    const int64_t rpb = (rows + nb - 1) / nb;
    BnRef none = {};
    hipLaunchKernelGGL((col_reduce_k<1>), dim3(nb), dim3(256), 0, st, x, (int64_t)channels, rows, channels, rpb, none, (float*)nullptr, sums);
    hipLaunchKernelGGL(bn_finalize_k, dim3((unsigned)((channels + 255) / 256)), dim3(256), 0, st, sums, rows, channels, eps, save_mean, save_var,
                       save_invstd);
    const int64_t total = rows * (int64_t)channels;
    hipLaunchKernelGGL(bn_apply_k, dim3(tr_blocks(total, 1024, 4096)), dim3(256), 0, st, x, total, channels, gamma, beta, save_mean, save_invstd,
                       act, slope, y);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_batchnorm_train_backward(const float* x, const float* y, const float* grad_y, int64_t rows, int channels, const float* gamma,
                                             const float* save_mean, const float* save_invstd, int act, float slope, float* grad_x,
                                             float* grad_gamma, float* grad_beta, void* workspace, size_t workspace_bytes, void* stream) {
    if (rows <= 0 || channels <= 0 || (act != 0 && act != 1)) return ML3D_E_INVALID;
    if (!x || !y || !grad_y || !save_mean || !save_invstd || !grad_x || !workspace) return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_batchnorm_train_workspace_bytes(channels)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double* sums = bn_ws(workspace);
    zero_async(sums, sizeof(double) * 2 * (size_t)channels, st);
    const unsigned nb = tr_blocks(rows, 256, 512);
    const int64_t rpb = (rows + nb - 1) / nb;
    BnRef ref = {y, grad_y, save_mean, save_invstd, act, slope};
    hipLaunchKernelGGL((col_reduce_k<2>), dim3(nb), dim3(256), 0, st, x, (int64_t)channels, rows, channels, rpb, ref, (float*)nullptr, sums);
    const int64_t total = rows * (int64_t)channels;
    hipLaunchKernelGGL(bn_backward_apply_k, dim3(tr_blocks(total, 1024, 4096)), dim3(256), 0, st, x, y, grad_y, rows, channels, gamma, save_mean,
                       save_invstd, sums, act, slope, grad_x, grad_gamma, grad_beta);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_gather_rows(const float* x, int64_t n_src, int channels, const int32_t* index, int64_t index_stride, int64_t m, float* out,
                                void* stream) {
    if (n_src < 0 || channels <= 0 || m < 0 || index_stride < 1) return ML3D_E_INVALID;
    if (m == 0) return 0;
    if (!index || !out || (n_src > 0 && !x)) return ML3D_E_INVALID;
    const int64_t total = m * (int64_t)channels;
    hipLaunchKernelGGL(gather_rows_k, dim3(tr_blocks(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, n_src, channels, index, index_stride,
                       m, out);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_scatter_add_rows(const float* grad_out, int64_t n_src, int channels, const int32_t* index, int64_t index_stride, int64_t m,
                                     float* grad_x, void* stream) {
    if (n_src < 0 || channels <= 0 || m < 0 || index_stride < 1) return ML3D_E_INVALID;
    if (n_src == 0) return 0;
    if (!grad_x) return ML3D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    zero_async(grad_x, sizeof(float) * (size_t)n_src * (size_t)channels, st);
    if (m == 0) return 0;
    if (!grad_out || !index) return ML3D_E_INVALID;
    const int64_t total = m * (int64_t)channels;
    hipLaunchKernelGGL(scatter_add_rows_k, dim3(tr_blocks(total, 256, 8192)), dim3(256), 0, st, grad_out, n_src, channels, index, index_stride, m,
                       grad_x);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_gather_pool_backward(const float* features, int64_t n_supports, int channels, const int32_t* inds, int64_t n_queries,
                                         int64_t max_neighbors, int mode, const float* grad_out, float* grad_features, void* stream) {
    if (n_supports < 0 || channels <= 0 || n_queries < 0 || max_neighbors < 0 || (mode != 0 && mode != 1)) return ML3D_E_INVALID;
    if (n_supports == 0) return 0;
    if (!grad_features) return ML3D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    zero_async(grad_features, sizeof(float) * (size_t)n_supports * (size_t)channels, st);
    if (n_queries == 0 || max_neighbors == 0) return 0;
    if (!inds || !grad_out || (mode == 0 && !features)) return ML3D_E_INVALID;
    const int64_t total = n_queries * (int64_t)channels;
    if (mode == 0)
        hipLaunchKernelGGL(max_pool_adjoint_k, dim3(tr_blocks(total, 256, 8192)), dim3(256), 0, st, features, n_supports, channels, inds, n_queries,
                           max_neighbors, grad_out, grad_features);
    else
        hipLaunchKernelGGL(scatter_add_rows_k, dim3(tr_blocks(total, 256, 8192)), dim3(256), 0, st, grad_out, n_supports, channels, inds,
                           max_neighbors, n_queries, grad_features);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

static int attn_check(int64_t batch, int64_t n, int k, int c1, int c2) {
    if (batch < 0 || n < 0 || c1 <= 0 || c2 <= 0) return ML3D_E_INVALID;
    const int d = c1 + c2;
    if (k != AS_K || d > AS_DMAX || (d & 1)) return ML3D_E_UNSUPPORTED;
    return 0;
}

extern "C" int ml3d_randla_attention_stage(const float* f, const float* enc, const int32_t* neighbor_idx, const float* weight_t, const float* bias,
                                           int64_t batch, int64_t n, int k, int c1, int c2, float* out, void* stream) {
    const int rc = attn_check(batch, n, k, c1, c2);
    if (rc) return rc;
    if (batch * n == 0) return 0;
    if (!f || !enc || !neighbor_idx || !weight_t || !out) return ML3D_E_INVALID;
    AttnArgs A = {};
    A.f = f; A.enc = enc; A.idx = neighbor_idx; A.wt = weight_t; A.bias = bias; A.batch = batch; A.n = n; A.c1 = c1; A.c2 = c2; A.out = out;
    const int d = c1 + c2;
    const int64_t npts = batch * n;
    if (d <= 128) {
        const int64_t tiles = (npts + 3) / 4;
        hipLaunchKernelGGL((attn_stage_fwd_k<64>), dim3(tr_blocks(tiles, 1, 2048)), dim3(256), 0, (hipStream_t)stream, A, tiles);
    } else {
        const int64_t tiles = (npts + 1) / 2;
        hipLaunchKernelGGL((attn_stage_fwd_k<32>), dim3(tr_blocks(tiles, 1, 2048)), dim3(256), 0, (hipStream_t)stream, A, tiles);
    }
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_randla_attention_stage_backward(const float* f, const float* enc, const int32_t* neighbor_idx, const float* weight,
                                                    const float* weight_t, const float* bias, const float* out, const float* grad_out,
                                                    int64_t batch, int64_t n, int k, int c1, int c2, float* grad_f, float* grad_enc,
                                                    float* grad_weight, float* grad_bias, void* stream) {
    const int rc = attn_check(batch, n, k, c1, c2);
    if (rc) return rc;
    if (!grad_weight) return ML3D_E_INVALID;
    const int d = c1 + c2;
    hipStream_t st = (hipStream_t)stream;
    zero_async(grad_weight, sizeof(float) * (size_t)d * (size_t)d, st);
    if (grad_bias) zero_async(grad_bias, sizeof(float) * (size_t)d, st);
    const int64_t npts = batch * n;
    if (npts == 0) return 0;
    if (!f || !enc || !neighbor_idx || !weight || !weight_t || !out || !grad_out || !grad_f || !grad_enc) return ML3D_E_INVALID;
    zero_async(grad_f, sizeof(float) * (size_t)npts * (size_t)c1, st);
    zero_async(grad_enc, sizeof(float) * (size_t)npts * AS_K * (size_t)c2, st);
    AttnArgs A = {};
    A.f = f; A.enc = enc; A.idx = neighbor_idx; A.w = weight; A.wt = weight_t; A.bias = bias; A.batch = batch; A.n = n; A.c1 = c1; A.c2 = c2;
    A.out = const_cast<float*>(out); A.gout = grad_out; A.gf = grad_f; A.genc = grad_enc; A.gw = grad_weight; A.gbias = grad_bias;
    const int n_cb = (d + AS_CB - 1) / AS_CB;
    const int64_t tiles = d <= 128 ? (npts + 3) / 4 : (npts + 1) / 2;
    int groups = (int)(tiles < 512 ? tiles : 512);
    if (groups < 1) groups = 1;
    if (d <= 128) hipLaunchKernelGGL((attn_stage_bwd_k<64>), dim3((unsigned)(groups * n_cb)), dim3(256), 0, st, A, tiles, groups);
    else hipLaunchKernelGGL((attn_stage_bwd_k<32>), dim3((unsigned)(groups * n_cb)), dim3(256), 0, st, A, tiles, groups);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}
