// train.hip — the training side of the hot path on hand-written HIP, forward AND backward:
//   * ml3d_gemm_tn                      C = A^T B over the rows (weight gradients of every Linear / 1x1 convolution and of KPConv's
//                                       [15 cin, cout] contraction: dW = x^T dy), f32 MFMA straight from global rows, split over rows
//   * ml3d_batchnorm_train_*            BatchNorm on the batch statistics (+ LeakyReLU), forward and backward
//                                       (SharedMLP.forward randlanet.py:503-518, BatchNormBlock.forward kpconv.py:1238-1249)
//   * ml3d_gather_rows / ml3d_scatter_add_rows / ml3d_gather_pool_backward
//                                       nearest_interpolation (randlanet.py:329-350), closest_pool / max_pool (kpconv.py:821-858)
//   * ml3d_kpconv_deformed_weighted[_backward]
//                                       the deformed KPConv's aggregation with per-query kernel points and its adjoint with respect
//                                       to the features AND the kernel points (kpconv.py:1011-1066, 1105-1137)
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
// coalesced 128-byte row segments -- into four 32 x 32 x 2 MFMAs per two rows; no LDS, no barrier.  The <= 2048 / tiles row slices
// meet in C (zeroed by the host call) through float atomics.  (Per-slice partial matrices + a summing kernel measured slower:
// profiles/r05_train_randlanet_hip_kernel_stats_v2.csv.)
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
    const bool two_i = i0 + 32 < k, two_j = j0 + 32 < n;             // does the tile have a second 32-row / 32-column half at all
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
        if (two_j) acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc01, 0, 0, 0);       // (wave-uniform: thin products -- 10 -> 8
        if (two_i) {                                                                            //  channels over 2.9 M rows -- issue one
            acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc10, 0, 0, 0);               //  MFMA per row pair, not four)
            if (two_j) acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc11, 0, 0, 0);
        }
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
// (row, channel) of the flat element index e = row * c + channel, advanced by a fixed stride WITHOUT a division per element (a 64-bit
// division is a ~150-instruction software loop on gfx950: it made the elementwise kernels compute-bound)
struct RowCh {
    int64_t row, drow;
    int ch, dch, c;
    __device__ __forceinline__ RowCh(int64_t first, int64_t stride, int c_) : c(c_) {
        row = first / c_; ch = (int)(first - row * c_);
        drow = stride / c_; dch = (int)(stride - drow * c_);
    }
    __device__ __forceinline__ void next() {
        row += drow; ch += dch;
        if (ch >= c) { ch -= c; ++row; }
    }
};

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
#pragma unroll 4
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
    const int64_t step = (int64_t)gridDim.x * blockDim.x, e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    RowCh w(e0, step, c);
    for (int64_t e = e0; e < total; e += step, w.next()) {
        const int ch = w.ch;
        float v = (x[e] - mean[ch]) * invstd[ch];
        v = v * (gamma ? gamma[ch] : 1.f) + (beta ? beta[ch] : 0.f);
        y[e] = act == 0 ? v : (v > 0.f ? v : v * slope);
    }
}

// gx = gamma invstd (g' - mean(g') - xhat mean(g' xhat));  ggamma = sum g' xhat, gbeta = sum g' (written by a grid-stride walk over the channels)
__global__ void __launch_bounds__(256)
bn_backward_apply_k(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gy, int64_t m, int c,
                    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
                    const double* __restrict__ sums, int act, float slope, float* __restrict__ gx, float* __restrict__ ggamma,
                    float* __restrict__ gbeta) {
    const int64_t total = m * (int64_t)c;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    // (a grid-stride walk over the channels: with 2-3 rows and c > 256 the grid has FEWER threads than channels)
    for (int64_t ch = first; ch < c; ch += step) {
        if (ggamma) ggamma[ch] = (float)sums[c + ch];
        if (gbeta) gbeta[ch] = (float)sums[ch];
    }
    RowCh w(first, step, c);
    for (int64_t e = first; e < total; e += step, w.next()) {
        const int ch = w.ch;
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
    const int64_t total = m * (int64_t)c, step = (int64_t)gridDim.x * blockDim.x, e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    RowCh w(e0, step, c);
    for (int64_t e = e0; e < total; e += step, w.next()) {
        const int64_t r = w.row;
        const int ch = w.ch;
        const int64_t s = idx[r * idx_stride];
        out[e] = (s >= 0 && s < n_src) ? x[s * c + ch] : 0.f;           // (a shadow index reads the zero row)
    }
}

__global__ void __launch_bounds__(256)
scatter_add_rows_k(const float* __restrict__ g, int64_t n_src, int c, const int32_t* __restrict__ idx, int64_t idx_stride, int64_t m,
                   float* __restrict__ gx) {
    const int64_t total = m * (int64_t)c, step = (int64_t)gridDim.x * blockDim.x, e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    RowCh w(e0, step, c);
    for (int64_t e = e0; e < total; e += step, w.next()) {
        const int64_t r = w.row;
        const int ch = w.ch;
        const int64_t s = idx[r * idx_stride];
        if (s >= 0 && s < n_src) atomicAdd(gx + s * c + ch, g[e]);
    }
}

// max_pool's adjoint (kpconv.py:841-858: max over the listed neighbours of the features padded with one zero row): the gradient of
// (q, ch) goes to the FIRST maximal neighbour in list order (torch.max's CPU rule); if that is the shadow row it is dropped
__global__ void __launch_bounds__(256)
max_pool_adjoint_k(const float* __restrict__ feat, int64_t ns, int c, const int32_t* __restrict__ inds, int64_t nq, int64_t H,
                   const float* __restrict__ g, float* __restrict__ gfeat) {
    const int64_t total = nq * (int64_t)c, step = (int64_t)gridDim.x * blockDim.x, e0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    RowCh w(e0, step, c);
    for (int64_t e = e0; e < total; e += step, w.next()) {
        const int64_t q = w.row;
        const int ch = w.ch;
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
//   gx = gs W + g p     accumulated over the column blocks in MFMA registers; the enc half is WRITTEN (every element has one
//                       owner), the f half scatters through idx with atomics (the inherent ones of an index_add)
//   gW[c, :] += gs[:, c]^T x     into the workgroup's PRIVATE [d x d] partial (L2-resident, plain loads / stores); a second
//                       kernel sums the partials of the <= 512 persistent workgroups -- no atomics, deterministic
//   gbias[c] += sum_k gs[k, c]   (zero up to rounding: softmax is shift invariant; kept for fidelity)
// (A first version split the column blocks over workgroups and met in grad_enc / grad_weight through atomics: ~90 M float atomics
//  per launch, 3.1 ms per launch and 22 of the 41 ms of a RandLA-Net training step -- profiles/r05_train_randlanet_hip_kernel_stats_v1.csv.)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int AS_K = 16;                 // neighbours per point (num_neighbors of every in-scope configuration)
constexpr int AS_CB = 64;                // score columns per pass
constexpr int AS_DMAX = 256;             // widest stage (d = dim_output[l]); wider stages stay on the unfused path
constexpr int AS_LDS_FLOATS = 64 * (128 + 1) + 64 * (AS_CB + 1);      // >= 32 * (256 + 1) + 32 * (AS_CB + 1)
constexpr int AS_MAXT = 4;               // 32 x 32 accumulator tiles per wave
// backward: R = 64 rows for d <= 64, 32 rows above; X [R][d | 1] + S [R][65] + GD [R][c1 | 1] (c1 < d)
constexpr int AS_C1MAX = 160;            // widest f half of a stage wider than 128 (GD must fit: 61.8 KB of LDS per workgroup)

struct AttnArgs {
    const float* f; const float* enc; const int32_t* idx; const float* w; const float* wt; const float* bias;
    int64_t batch, n; int c1, c2;
    float* out;                                       // forward result / saved forward result
    const float* gout; float* gf; float* genc; float* gw; float* gbias;
};

// stage the R x d rows of tile `tile` (points tile * TP ...) into X (row pitch ldx); rows past the last point are zeros.  The (row,
// column) of a thread's elements advance incrementally and the batch item of a point comes from 32-bit arithmetic (the host call
// refuses batch * n >= 2^31): no software division per element.
template <int R>
__device__ __forceinline__ void attn_stage_x(const AttnArgs& A, int64_t tile, float* X, int ldx) {
    constexpr int TP = R / AS_K;
    const int d = A.c1 + A.c2;
    const uint32_t npts = (uint32_t)(A.batch * A.n), n = (uint32_t)A.n;
    const uint32_t pt0 = (uint32_t)tile * TP;
    int row = (int)threadIdx.x / d, col = (int)threadIdx.x - row * d;
    const int drow = 256 / d, dcol = 256 - drow * d;
    for (; row < R; ) {
        const uint32_t pt = pt0 + (uint32_t)(row / AS_K);
        float v = 0.f;
        if (pt < npts) {
            const int kk = row % AS_K;
            if (col < A.c1) {
                const uint32_t b = pt / n;
                const int64_t src = A.idx[(int64_t)pt * AS_K + kk];
                if (src >= 0 && src < A.n) v = A.f[((int64_t)b * A.n + src) * A.c1 + col];
            } else {
                v = A.enc[((int64_t)pt * AS_K + kk) * A.c2 + (col - A.c1)];
            }
        }
        X[row * ldx + col] = v;
        row += drow; col += dcol;
        if (col >= d) { col -= d; ++row; }
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
#pragma unroll 8
        for (int kk = 0; kk < d; kk += 2) {                           // (8 rows of W^T in flight per wave: bound by their L2 latency)
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

// DMAX = the widest stage of the class the kernel is instantiated for (16 / 64 / 128 / 256): it sizes the LDS (25 / 25 / 41 / 62 KB ->
// 6 / 6 / 3 / 2 workgroups per CU) and picks the tile height (64 rows for d <= 16, 32 above)
template <int DMAX>
__global__ void __launch_bounds__(256) ML3D_WAVES_PER_SIMD(2)
attn_stage_bwd_k(AttnArgs A, int64_t n_tiles, int groups, float* __restrict__ gw_partial) {
    constexpr int R = DMAX <= 16 ? 64 : 32;
    constexpr int C1MAX = DMAX <= 128 ? DMAX : AS_C1MAX;
    constexpr int TP = R / AS_K;
    constexpr int RT = R / 32;
    constexpr int NX = (RT * ((DMAX + 31) / 32) + 3) / 4;            // gx accumulator tiles per wave (1, 1, 1, 2)
    __shared__ float lds[R * (DMAX + 1) + R * (AS_CB + 1) + R * (C1MAX + 1)];
    const int d = A.c1 + A.c2, ldx = d | 1, ldg = A.c1 | 1;
    float* X = lds;                                                   // [R][ldx]   the staged rows
    float* S = X + R * ldx;                                           // [R][65]    scores, then gs, of the current column block
    float* GD = S + R * (AS_CB + 1);                                  // [R][ldg]   the direct term p g of the f columns
    const int64_t npts = A.batch * A.n;
    const int grp = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hi = lane >> 5, cl = lane & 31;
    const int t = threadIdx.x;
    const int n_jt = (d + 31) / 32;                                   // 32-column tiles across d
    float* gwp = gw_partial + (int64_t)grp * d * d;                   // this workgroup's private [d x d] partial of grad_weight
    float gb[AS_DMAX / AS_CB];                                        // thread t < TP * 64 owns score columns cb0 + t % 64
#pragma unroll
    for (int q = 0; q < AS_DMAX / AS_CB; ++q) gb[q] = 0.f;
    for (int64_t tile = grp; tile < n_tiles; tile += groups) {
        const bool first = tile == grp;
        attn_stage_x<R>(A, tile, X, ldx);
        tr_f32x16 gx[NX];                                        // gx tiles (rt, jt): q = rt * n_jt + jt on wave q % 4, slot q / 4
#pragma unroll
        for (int q = 0; q < NX; ++q) gx[q] = tr_zero16();
        __syncthreads();
#pragma unroll 1
        for (int cbi = 0; cbi * AS_CB < d; ++cbi) {
            const int cb0 = cbi * AS_CB;
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
                float gbs = 0.f;
#pragma unroll
                for (int k = 0; k < AS_K; ++k) {
                    const int row = tp * AS_K + k;
                    const float pg = on ? sv[k] / sum * g : 0.f;
                    const float gs = on ? pg * (X[row * ldx + col] - o) : 0.f;
                    S[row * (AS_CB + 1) + c] = gs;
                    gbs += gs;
                    if (on) {                                          // the direct term d out / d x = p g
                        if (col < A.c1) GD[row * ldg + col] = pg;      // (f columns: joins gx in LDS before the scatter)
                        else A.genc[(pt * AS_K + k) * A.c2 + (col - A.c1)] = pg;     // (enc columns: gx is added to it below)
                    }
                }
#pragma unroll
                for (int q = 0; q < AS_DMAX / AS_CB; ++q) gb[q] += q == cbi ? gbs : 0.f;
            }
            __syncthreads();
            // ---- gW[cb0 + i, j] += sum_rows gs[row, i] x[row, j]: the partial lives in the workgroup's own [d x d] block of
            //      gw_partial (L2-resident; plain loads / stores, no atomics); tiles (it in 0..1, jt) on wave q % 4
#pragma unroll 1
            for (int tq = wave; tq < 2 * n_jt; tq += 4) {
                const int it = tq / n_jt, jt = tq - it * n_jt;
                const int j = jt * 32 + cl;
                const bool jok = j < d;
                const float* sp = S + hi * (AS_CB + 1) + it * 32 + cl;
                const float* xp = X + hi * ldx + (jok ? j : 0);
                tr_f32x16 acc = tr_zero16();
                if (!first && jok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = cb0 + it * 32 + tr_row(r, hi);
                        if (i < d) acc[r] = gwp[(int64_t)i * d + j];
                    }
                }
#pragma unroll 4
                for (int r = 0; r < R; r += 2) {
                    const float av = sp[r * (AS_CB + 1)];
                    const float bv = jok ? xp[r * ldx] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
                }
                if (jok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = cb0 + it * 32 + tr_row(r, hi);
                        if (i < d) gwp[(int64_t)i * d + j] = acc[r];
                    }
                }
            }
            // ---- gx[row, j] += sum_c gs[row, c] W[cb0 + c, j]
#pragma unroll
            for (int q = 0; q < NX; ++q) {
                const int tq = q * 4 + wave;
                if (tq < RT * n_jt) {                                 // (wave-uniform)
                    const int rt = tq / n_jt, jt = tq - rt * n_jt;
                    const int j = jt * 32 + cl;
                    const bool jok = j < d;
                    const float* sp = S + (rt * 32 + cl) * (AS_CB + 1) + hi;
                    tr_f32x16 acc = gx[q];
#pragma unroll 8
                    for (int cc = 0; cc < AS_CB; cc += 2) {               // (8 rows of W in flight per wave: the loop is bound by their L2 latency)
                        const float av = sp[cc];
                        const bool wok = jok && cb0 + cc + hi < d;
                        const float bv = wok ? A.w[(int64_t)(cb0 + cc + hi) * d + j] : 0.f;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
                    }
                    gx[q] = acc;
                }
            }
            __syncthreads();
        }
        // ---- gx out: enc columns add to the direct term stored above (plain read-modify-write: every element has one owner),
        //      f columns + their direct term scatter through idx with atomics
#pragma unroll
        for (int q = 0; q < NX; ++q) {
            const int tq = q * 4 + wave;
            if (tq < RT * n_jt) {
                const int rt = tq / n_jt, jt = tq - rt * n_jt;
                const int j = jt * 32 + cl;
                if (j < d) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rt * 32 + tr_row(r, hi);
                        const int64_t pt = tile * TP + row / AS_K;
                        if (pt < npts) {
                            const int kk = row % AS_K;
                            if (j < A.c1) {
                                const int64_t src = A.idx[pt * AS_K + kk];
                                const int64_t b = (int64_t)((uint32_t)pt / (uint32_t)A.n);
                                if (src >= 0 && src < A.n) atomicAdd(A.gf + (b * A.n + src) * A.c1 + j, gx[q][r] + GD[row * ldg + j]);
                            } else {
                                float* dst = A.genc + (pt * AS_K + kk) * A.c2 + (j - A.c1);
                                *dst = *dst + gx[q][r];
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    if (A.gbias && t < TP * AS_CB) {
#pragma unroll
        for (int q = 0; q < AS_DMAX / AS_CB; ++q)
            if (q * AS_CB + t % AS_CB < d) atomicAdd(A.gbias + q * AS_CB + t % AS_CB, gb[q]);
    }
}

// out[c][e] = sum of partial[g][e] over the groups g of chunk c (32 groups per chunk): one level of the tree that adds up the
// persistent workgroups' private partial sums -- coalesced, fixed order (deterministic)
__global__ void __launch_bounds__(256)
sum_partials_k(const float* __restrict__ partial, int64_t elems, int groups, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= elems) return;
    const int g0 = blockIdx.y * 32, g1 = g0 + 32 < groups ? g0 + 32 : groups;
    float s = 0.f;
    for (int g = g0; g < g1; ++g) s += partial[(int64_t)g * elems + e];
    out[(int64_t)blockIdx.y * elems + e] = s;
}

// workgroups of a per-column reduction over [rows, c]: ~32 elements per thread, at least one row per workgroup (few rows x many channels --
// KPConv's coarse levels -- must not end up on a handful of workgroups: 4 workgroups walked [1000, 1024] in 567 us)
static inline unsigned tr_reduce_blocks(int64_t rows, int c) {
    int64_t b = (rows * (int64_t)c + 8191) / 8192;
    if (b > rows) b = rows;
    if (b > 2048) b = 2048;
    return (unsigned)(b < 1 ? 1 : b);
}


// ---------------------------------------------------------------------------------------------------------------------
// The DEFORMED KPConv aggregation for training (kpconv.py:1011-1066, 1105-1137, KP_influence linear, sum aggregation): the kernel points
// of query q are its own, dkp[q, k, :] = kernel_points[k] + extent * offsets[q, k, :], so the influences depend on TRAINED quantities:
//   wf[q, k, c] = sum_h w[q, k, h] x[inds[q, h], c],   w = max(0, 1 - |s[inds[q, h]] - q - dkp[q, k]| / extent)
// Forward: one wave per query, lane = channel (chunks of 64), lanes 0..14 compute the 15 influences of a neighbour once, every lane
// takes them by broadcast.  Backward, same shape: dx[inds[q, h], c] += sum_k w dwf[q, k, c] (atomics: the scatter of an index_add),
//   d dkp[q, k, :] = sum_h (sum_c dwf[q, k, c] x[inds[q, h], c]) * (nb - dkp[q, k]) / (|nb - dkp[q, k]| extent)      where w > 0
// -- the inner sum over the channels is a wave reduction per kernel point.  Neither pass holds the reference's [Nq, H, Cin] gather.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DK = 15;                   // kernel points (every in-scope configuration)

__device__ __forceinline__ float tr_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct DeformArgs {
    const float* q_pts; const float* s_pts; const int32_t* inds; int64_t nq, ns; int h;
    const float* x; int cin; const float* dkp; float extent;
};

__global__ void __launch_bounds__(256)
kp_deformed_weighted_k(DeformArgs A, float* __restrict__ wf) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= A.nq) return;                                           // (wave-uniform)
    const float qx = A.q_pts[3 * q], qy = A.q_pts[3 * q + 1], qz = A.q_pts[3 * q + 2];
    const int k = lane < DK ? lane : 0;
    const float kx = A.dkp[(q * DK + k) * 3], ky = A.dkp[(q * DK + k) * 3 + 1], kz = A.dkp[(q * DK + k) * 3 + 2];
    const int32_t* row = A.inds + q * A.h;
    for (int c0 = 0; c0 < A.cin; c0 += 64) {
        const int c = c0 + lane;
        const bool live = c < A.cin;
        float acc[DK];
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) acc[kk] = 0.f;
        for (int h = 0; h < A.h; ++h) {
            const int idx = row[h];
            if (idx < 0 || idx >= A.ns) continue;                    // shadow neighbour (wave-uniform: the row is the wave's)
            const float* sp = A.s_pts + 3 * (int64_t)idx;
            const float dx = (sp[0] - qx) - kx, dy = (sp[1] - qy) - ky, dz = (sp[2] - qz) - kz;
            // (the reference's operation sequence, 1 - sqrt(sq) / extent with sq = (dx^2 + dy^2) + dz^2: the same float32 value, so the
            //  clamp's mask is the reference's bit for bit)
            float w = 1.0f - sqrtf(dx * dx + dy * dy + dz * dz) / A.extent;
            w = (lane < DK && w > 0.f) ? w : 0.f;
            const float xv = live ? A.x[(int64_t)idx * A.cin + c] : 0.f;
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) acc[kk] = fmaf(__shfl(w, kk), xv, acc[kk]);
        }
        if (live) {
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) wf[(q * DK + kk) * A.cin + c] = acc[kk];
        }
    }
}

__global__ void __launch_bounds__(256)
kp_deformed_weighted_bwd_k(DeformArgs A, const float* __restrict__ dwf, float* __restrict__ dx, float* __restrict__ gkp) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= A.nq) return;                                           // (wave-uniform)
    const float qx = A.q_pts[3 * q], qy = A.q_pts[3 * q + 1], qz = A.q_pts[3 * q + 2];
    const int k = lane < DK ? lane : 0;
    const float kx = A.dkp[(q * DK + k) * 3], ky = A.dkp[(q * DK + k) * 3 + 1], kz = A.dkp[(q * DK + k) * 3 + 2];
    const int32_t* row = A.inds + q * A.h;
    float gx = 0.f, gy = 0.f, gz = 0.f;                               // lanes 0..14: the gradient of their kernel point
    for (int c0 = 0; c0 < A.cin; c0 += 64) {
        const int c = c0 + lane;
        const bool live = c < A.cin;
        float g[DK];
#pragma unroll
        for (int kk = 0; kk < DK; ++kk) g[kk] = live ? dwf[(q * DK + kk) * A.cin + c] : 0.f;
        for (int h = 0; h < A.h; ++h) {
            const int idx = row[h];
            if (idx < 0 || idx >= A.ns) continue;
            const float* sp = A.s_pts + 3 * (int64_t)idx;
            const float ddx = (sp[0] - qx) - kx, ddy = (sp[1] - qy) - ky, ddz = (sp[2] - qz) - kz;
            const float dist = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
            const float wpre = 1.0f - dist / A.extent;
            const float w = (lane < DK && wpre > 0.f) ? wpre : 0.f;
            const float xv = live ? A.x[(int64_t)idx * A.cin + c] : 0.f;
            float v = 0.f, mine = 0.f;
#pragma unroll
            for (int kk = 0; kk < DK; ++kk) {
                v = fmaf(__shfl(w, kk), g[kk], v);
                const float s = tr_wave_sum(g[kk] * xv);             // d loss / d w[q, kk, h] of this channel chunk
                mine = lane == kk ? s : mine;
            }
            if (live) atomicAdd(dx + (int64_t)idx * A.cin + c, v);
            if (lane < DK && wpre >= 0.f && dist > 0.f) {             // (torch.clamp's gradient mask is inclusive at the bound)
                const float coef = mine / (A.extent * dist);
                gx = fmaf(coef, ddx, gx); gy = fmaf(coef, ddy, gy); gz = fmaf(coef, ddz, gz);
            }
        }
    }
    if (lane < DK) {
        gkp[(q * DK + lane) * 3] = gx; gkp[(q * DK + lane) * 3 + 1] = gy; gkp[(q * DK + lane) * 3 + 2] = gz;
    }
}

static inline unsigned tr_blocks(int64_t total, int per, unsigned cap) {
    int64_t b = (total + per - 1) / per;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}


// ---------------------------------------------------------------------------------------------------------------------
// p2p_fitting_regularizer of the deformable KPConv (kpconv.py:2167-2206 + the min_d2 of kpconv.py:1058-1074), value AND gradient
// in one pass, without the reference's [Nq, H, K] distance tensor: thread (q, k) walks the H neighbours of query q for ITS
// deformed kernel point -- fitting term min_h |s[inds[q, h]] - q - kp[q, k]|^2 / extent^2 (the shadow neighbour sits at 1e6 like the
// reference's padded support), its gradient 2 (kp - nb*) / extent^2 goes to the FIRST minimum (torch.min) -- and the repulsive term
// sum_{j != k} min(|l_j - l_k| - r, 0)^2 over the query's other kernel points l = kp / extent (detached, as in the reference), whose
// K points are exchanged inside a 16-lane group.  Per-workgroup partial sums in double (deterministic; the caller adds them).
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
kp_offset_reg_k(const float* __restrict__ q_pts, const float* __restrict__ s_pts, const int32_t* __restrict__ inds, int64_t nq, int64_t ns,
                int64_t H, int K, const float* __restrict__ dkp, float extent, float repulse, float* __restrict__ min_d2,
                float* __restrict__ g_fit, float* __restrict__ g_rep, double* __restrict__ partial) {
    __shared__ double red[2][256];
    const int t = threadIdx.x, k = t & 15;
    const int64_t q = (int64_t)blockIdx.x * 16 + (t >> 4);
    const bool live = q < nq && k < K;
    float fit = 0.f, rep = 0.f;
    float kx = 0.f, ky = 0.f, kz = 0.f;
    if (live) { const float* p = dkp + (q * K + k) * 3; kx = p[0]; ky = p[1]; kz = p[2]; }
    if (live) {
        const float qx = q_pts[3 * q], qy = q_pts[3 * q + 1], qz = q_pts[3 * q + 2];
        float best = 3.0e38f, bx = 0.f, by = 0.f, bz = 0.f;
        for (int64_t h = 0; h < H; ++h) {
            const int64_t s = inds[q * H + h];
            const bool real = s >= 0 && s < ns;
            const float nx = (real ? s_pts[3 * s] : 1.0e6f) - qx, ny = (real ? s_pts[3 * s + 1] : 1.0e6f) - qy,
                        nz = (real ? s_pts[3 * s + 2] : 1.0e6f) - qz;
            const float dx = nx - kx, dy = ny - ky, dz = nz - kz;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < best) { best = d2; bx = dx; by = dy; bz = dz; }
        }
        if (H == 0) best = 0.f;
        const float ie2 = 1.0f / (extent * extent);
        fit = best * ie2;
        if (min_d2) min_d2[q * K + k] = best;
        if (g_fit) { float* g = g_fit + (q * K + k) * 3; g[0] = -2.f * bx * ie2; g[1] = -2.f * by * ie2; g[2] = -2.f * bz * ie2; }
    }
    // the query's K kernel points, normalised, from the lanes of its 16-lane group
    const float lx = kx / extent, ly = ky / extent, lz = kz / extent;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    const int base = (t & 63) & ~15;
    for (int j = 0; j < K; ++j) {
        const float ox = __shfl(lx, base + j), oy = __shfl(ly, base + j), oz = __shfl(lz, base + j);
        if (j == k || !live) continue;
        const float dx = ox - lx, dy = oy - ly, dz = oz - lz;
        const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
        const float c = fminf(dist - repulse, 0.f);
        rep += c * c;
        if (c < 0.f && dist > 0.f) { const float w = 2.f * c / dist; gx -= w * dx; gy -= w * dy; gz -= w * dz; }   // d dist / d l_k = -(l_j - l_k) / dist
    }
    if (live && g_rep) { float* g = g_rep + (q * K + k) * 3; g[0] = gx / extent; g[1] = gy / extent; g[2] = gz / extent; }
    red[0][t] = (double)fit; red[1][t] = (double)rep;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; }
        __syncthreads();
    }
    if (t == 0) { partial[2 * blockIdx.x] = red[0][0]; partial[2 * blockIdx.x + 1] = red[1][0]; }
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
    int64_t slices = (2048 + ntiles - 1) / ntiles;                      // ~2048 waves in flight
    const int64_t max_slices = (m + 127) / 128;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    int64_t rps = (m + slices - 1) / slices;
    rps = (rps + 1) & ~(int64_t)1;
    slices = (m + rps - 1) / rps;
    const int64_t units = slices * ntiles;
    hipLaunchKernelGGL(gemm_tn_k, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, a, lda, b, ldb, m, k, n, rps, tiles_j, units, c, ldc);
    if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    if (col_sums_a) {
        const unsigned nb = tr_reduce_blocks(m, k);
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
    const unsigned nb = tr_reduce_blocks(rows, channels);
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
    const unsigned nb = tr_reduce_blocks(rows, channels);
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


static int deform_args(DeformArgs& a, const float* q_pts, const float* s_pts, const int32_t* inds, int64_t nq, int64_t ns, int64_t h,
                       const float* x, int cin, const float* dkp, int num_kernel_points, float extent) {
    if (nq < 0 || ns < 0 || h < 0 || h > 0x7fffffff || cin <= 0 || !(extent > 0.f)) return ML3D_E_INVALID;
    if (num_kernel_points != DK) return ML3D_E_UNSUPPORTED;
    if (nq > 0 && (!q_pts || !dkp || (h > 0 && (!inds || !s_pts || !x)))) return ML3D_E_INVALID;
    a.q_pts = q_pts; a.s_pts = s_pts; a.inds = inds; a.nq = nq; a.ns = ns; a.h = (int)h; a.x = x; a.cin = cin; a.dkp = dkp;
    a.extent = extent;
    return 0;
}

extern "C" int64_t ml3d_kpconv_offset_regulariser_blocks(int64_t n_queries) { return n_queries > 0 ? (n_queries + 15) / 16 : 0; }

extern "C" int ml3d_kpconv_offset_regulariser(const float* q_pts, const float* s_pts, const int32_t* neighb_inds, int64_t n_queries,
                                              int64_t n_supports, int64_t max_neighbors, const float* deformed_kernel_points,
                                              int num_kernel_points, float kp_extent, float repulse_extent, float* out_min_d2,
                                              float* out_grad_fitting, float* out_grad_repulsive, double* out_partial_sums, void* stream) {
    if (n_queries < 0 || n_supports < 0 || max_neighbors < 0 || num_kernel_points <= 0 || num_kernel_points > 16 || !(kp_extent > 0.f))
        return num_kernel_points > 16 ? ML3D_E_UNSUPPORTED : ML3D_E_INVALID;
    if (n_queries == 0) return 0;
    if (!q_pts || !deformed_kernel_points || !out_partial_sums || (max_neighbors > 0 && (!neighb_inds || (n_supports > 0 && !s_pts))))
        return ML3D_E_INVALID;
    hipLaunchKernelGGL(kp_offset_reg_k, dim3((unsigned)((n_queries + 15) / 16)), dim3(256), 0, (hipStream_t)stream, q_pts, s_pts, neighb_inds,
                       n_queries, n_supports, max_neighbors, num_kernel_points, deformed_kernel_points, kp_extent, repulse_extent, out_min_d2,
                       out_grad_fitting, out_grad_repulsive, out_partial_sums);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_kpconv_deformed_weighted(const float* q_pts, const float* s_pts, const int32_t* neighb_inds, int64_t n_queries,
                                             int64_t n_supports, int64_t max_neighbors, const float* features, int cin,
                                             const float* deformed_kernel_points, int num_kernel_points, float kp_extent, float* out_wf,
                                             void* stream) {
    DeformArgs a;
    const int rc = deform_args(a, q_pts, s_pts, neighb_inds, n_queries, n_supports, max_neighbors, features, cin, deformed_kernel_points,
                               num_kernel_points, kp_extent);
    if (rc) return rc;
    if (n_queries == 0) return 0;
    if (!out_wf) return ML3D_E_INVALID;
    hipLaunchKernelGGL(kp_deformed_weighted_k, dim3((unsigned)((n_queries + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, out_wf);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_kpconv_deformed_weighted_backward(const float* q_pts, const float* s_pts, const int32_t* neighb_inds, int64_t n_queries,
                                                      int64_t n_supports, int64_t max_neighbors, const float* features, int cin,
                                                      const float* deformed_kernel_points, int num_kernel_points, float kp_extent,
                                                      const float* grad_wf, float* grad_features, float* grad_kernel_points, void* stream) {
    DeformArgs a;
    const int rc = deform_args(a, q_pts, s_pts, neighb_inds, n_queries, n_supports, max_neighbors, features, cin, deformed_kernel_points,
                               num_kernel_points, kp_extent);
    if (rc) return rc;
    if ((n_supports > 0 && !grad_features) || (n_queries > 0 && !grad_kernel_points)) return ML3D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    if (n_supports > 0) zero_async(grad_features, sizeof(float) * (size_t)n_supports * (size_t)cin, st);
    if (n_queries == 0) return 0;
    if (!grad_wf) return ML3D_E_INVALID;
    hipLaunchKernelGGL(kp_deformed_weighted_bwd_k, dim3((unsigned)((n_queries + 3) / 4)), dim3(256), 0, st, a, grad_wf, grad_features,
                       grad_kernel_points);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

static int attn_check(int64_t batch, int64_t n, int k, int c1, int c2) {
    if (batch < 0 || n < 0 || c1 <= 0 || c2 <= 0) return ML3D_E_INVALID;
    const int d = c1 + c2;
    if (batch * n >= ((int64_t)1 << 31) / 16) return ML3D_E_UNSUPPORTED;           // (32-bit point / row arithmetic inside the kernels)
    if (k != AS_K || d > AS_DMAX || (d & 1) || (d > 128 && c1 > AS_C1MAX)) return ML3D_E_UNSUPPORTED;
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

static int attn_bwd_plan(int64_t npts, int d, int64_t* tiles) {
    *tiles = d <= 16 ? (npts + 3) / 4 : (npts + 1) / 2;
    int64_t g = *tiles < 2048 ? *tiles : 2048;
    const int64_t cap = ((int64_t)160 << 20) / ((int64_t)d * d * 4);    // <= 160 MB of private partials (640 workgroups at d = 256)
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

extern "C" size_t ml3d_randla_attention_stage_backward_workspace_bytes(int64_t batch, int64_t n, int c1, int c2) {
    if (batch <= 0 || n <= 0 || c1 <= 0 || c2 <= 0) return 0;
    const int d = c1 + c2;
    int64_t tiles;
    const int64_t groups = attn_bwd_plan(batch * n, d, &tiles);
    return sizeof(float) * (size_t)(groups + (groups + 31) / 32) * (size_t)d * (size_t)d + 256;
}

extern "C" int ml3d_randla_attention_stage_backward(const float* f, const float* enc, const int32_t* neighbor_idx, const float* weight,
                                                    const float* weight_t, const float* bias, const float* out, const float* grad_out,
                                                    int64_t batch, int64_t n, int k, int c1, int c2, float* grad_f, float* grad_enc,
                                                    float* grad_weight, float* grad_bias, void* workspace, size_t workspace_bytes,
                                                    void* stream) {
    const int rc = attn_check(batch, n, k, c1, c2);
    if (rc) return rc;
    if (!grad_weight) return ML3D_E_INVALID;
    const int d = c1 + c2;
    hipStream_t st = (hipStream_t)stream;
    if (grad_bias) zero_async(grad_bias, sizeof(float) * (size_t)d, st);
    const int64_t npts = batch * n;
    if (npts == 0) { zero_async(grad_weight, sizeof(float) * (size_t)d * (size_t)d, st); return 0; }
    if (!f || !enc || !neighbor_idx || !weight || !weight_t || !out || !grad_out || !grad_f || !grad_enc || !workspace) return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_randla_attention_stage_backward_workspace_bytes(batch, n, c1, c2)) return ML3D_E_WORKSPACE;
    zero_async(grad_f, sizeof(float) * (size_t)npts * (size_t)c1, st);
    AttnArgs A = {};
    A.f = f; A.enc = enc; A.idx = neighbor_idx; A.w = weight; A.wt = weight_t; A.bias = bias; A.batch = batch; A.n = n; A.c1 = c1; A.c2 = c2;
    A.out = const_cast<float*>(out); A.gout = grad_out; A.gf = grad_f; A.genc = grad_enc; A.gw = grad_weight; A.gbias = grad_bias;
    float* partial = (float*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int64_t tiles;
    const int groups = attn_bwd_plan(npts, d, &tiles);
    const dim3 grid((unsigned)groups), block(256);
    if (d <= 16) hipLaunchKernelGGL((attn_stage_bwd_k<16>), grid, block, 0, st, A, tiles, groups, partial);
    else if (d <= 64) hipLaunchKernelGGL((attn_stage_bwd_k<64>), grid, block, 0, st, A, tiles, groups, partial);
    else if (d <= 128) hipLaunchKernelGGL((attn_stage_bwd_k<128>), grid, block, 0, st, A, tiles, groups, partial);
    else hipLaunchKernelGGL((attn_stage_bwd_k<256>), grid, block, 0, st, A, tiles, groups, partial);
    // the private partials -> grad_weight: 32 at a time, then the <= 64 chunk sums
    const int64_t elems = (int64_t)d * d;
    const int chunks = (groups + 31) / 32;
    const unsigned eb = (unsigned)((elems + 255) / 256);
    if (chunks == 1) {
        hipLaunchKernelGGL(sum_partials_k, dim3(eb, 1), dim3(256), 0, st, partial, elems, groups, grad_weight);
    } else {
        float* level1 = partial + (int64_t)groups * elems;
        hipLaunchKernelGGL(sum_partials_k, dim3(eb, (unsigned)chunks), dim3(256), 0, st, partial, elems, groups, level1);
        if (chunks <= 32) {
            hipLaunchKernelGGL(sum_partials_k, dim3(eb, 1), dim3(256), 0, st, level1, elems, chunks, grad_weight);
        } else {                                                        // (chunks <= 64: the first-level buffer is free again)
            hipLaunchKernelGGL(sum_partials_k, dim3(eb, (unsigned)((chunks + 31) / 32)), dim3(256), 0, st, level1, elems, chunks, partial);
            hipLaunchKernelGGL(sum_partials_k, dim3(eb, 1), dim3(256), 0, st, partial, elems, (chunks + 31) / 32, grad_weight);
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}
