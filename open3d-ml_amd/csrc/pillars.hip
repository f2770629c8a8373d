// pillars.hip — PointPillars inference forward pieces for gfx950.
//
// Replaces the PyTorch op chains of
//   PointPillarsVoxelization.forward (after voxelize)  ml3d/torch/models/point_pillars.py:359-382
//       ragged_to_dense + feats[idx] gather + out-of-bounds filter
//   PillarFeatureNet.forward + PFNLayer.forward          point_pillars.py:512-555, 417-453
//   PointPillarsScatter.forward                          point_pillars.py:577-616
//   SECOND / SECONDFPN / Anchor3DHead forward            point_pillars.py:666-682, 739-755, 827-841
//
// pillar_pfn: ONE wave per pillar.  The wave reads the pillar's <= 64 points straight from the ragged
// voxelize result (voxel_point_indices), so the dense [M, 32, 4] tensor, the [M, 32, 9] decorated tensor
// and the [M, 32, 64] activation of the reference never exist in HBM: the mean is a wave reduction, the
// 9 decorations go to LDS, lane u owns output unit u of Linear(9 -> 64) + folded BatchNorm + ReLU and
// keeps the max over the points in a register; the 64-float pillar feature is written directly into its
// (b, y, x) pixel of the zeroed NHWC canvas (pillar scatter fused in).  Two-layer PFNs (argoverse /
// nuscenes) write the concatenated [x | max] rows and finish in pfn_next.
// The BEV maps are NHWC so a pixel's channels are one contiguous burst; the 3x3 / strided convs and the
// transposed convs run on the f32 MFMA implicit-GEMM of gemm.hip (BN + ReLU in the epilogue, deconv =
// GEMM + pixel-shuffle store straight into the concatenated 384-channel neck map).  The three 1x1 heads
// are one GEMM; head_to_nchw returns the reference's NCHW tensors.
// Rooflines: pillar_pfn + canvas zeroing = HBM (16 B/point + 256 B/pillar + 4*64*ny*nx canvas bytes);
// convs = f32 matrix peak 157.3 TFLOP/s (68.3 GFLOP per KITTI frame).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "gemm.h"
#include "grid.h"
#include "ml3d_hip.h"

namespace ml3d {

constexpr int PF_MAXP = 64;     // points per pillar handled by one wave
constexpr int PF_MAXC = 16;     // decorated channels (in_channels + 5)

struct PfnArgs {
    const float* points; int64_t point_stride; int in_ch;      // raw [N, stride] rows, first in_ch used (xyz first)
    const int32_t* coords;                                      // [M, 3] (x, y, z) from voxelize
    const int64_t* pidx; const int64_t* prs;                    // ragged point lists
    const int64_t* batch_splits; int batch;                     // voxels of sample b: [bs[b], bs[b+1])
    int64_t n_pillars;
    int P;                                                      // max_num_points (rows of the reference's dense tensor)
    float vx, vy, x_off, y_off;
    int nx, ny;                                                 // canvas / in-bounds limits
    const float* wt; const float* bias; int units;              // folded Linear: wt [in_ch + 5][units]
    int last;                                                   // 1: max -> canvas; 0: [x | max] rows -> xcat
    float* canvas; int canvas_c;                                // NHWC [B, ny, nx, canvas_c]
    float* xcat;                                                // [M, P, 2 * units]
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ int pillar_sample(const int64_t* bs, int batch, int64_t m) {
    int lo = 0, hi = batch;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (bs[mid] <= m) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ void __launch_bounds__(256) pillar_pfn(PfnArgs A) {
    __shared__ float D[4][PF_MAXP][PF_MAXC];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t m = (int64_t)blockIdx.x * 4 + w;
    if (m >= A.n_pillars) return;                 // wave-uniform; no block barrier below
    const int cx = A.coords[3 * m], cy = A.coords[3 * m + 1];
    if (cx >= A.nx || cy >= A.ny) return;         // pillar on the upper range bound (point_pillars.py:373-380)
    const int64_t p0 = A.prs[m];
    const int np = (int)(A.prs[m + 1] - p0);      // <= P
    const int C = A.in_ch, DC = A.in_ch + 5;
    // ---- decorate: lane j = point j ------------------------------------------------------------------------
    float f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = 0.f;
    const bool real = lane < np;
    if (real) {
        const float* p = A.points + A.point_stride * A.pidx[p0 + lane];
#pragma unroll
        for (int c = 0; c < 8; ++c) if (c < C) f[c] = p[c];
    }
    const float inv = 1.0f / (float)np;
    const float mx = wave_sum(f[0]) * inv, my = wave_sum(f[1]) * inv, mz = wave_sum(f[2]) * inv;
    if (lane < A.P && lane < PF_MAXP) {
        float* d = D[w][lane];
        if (real) {
#pragma unroll
            for (int c = 0; c < 8; ++c) if (c < C) d[c] = f[c];
            d[C + 0] = f[0] - mx; d[C + 1] = f[1] - my; d[C + 2] = f[2] - mz;
            d[C + 3] = f[0] - ((float)cx * A.vx + A.x_off);
            d[C + 4] = f[1] - ((float)cy * A.vy + A.y_off);
        } else {
            for (int c = 0; c < DC; ++c) d[c] = 0.f;       // masked rows stay zero (point_pillars.py:546-549)
        }
    }
    wave_sync();
    // ---- Linear + folded BN + ReLU, max over the P rows: lane u = unit u ----------------------------------------
    const int sample = pillar_sample(A.batch_splits, A.batch, m);
    for (int u = lane; u < A.units; u += 64) {
        float wcol[PF_MAXC];
#pragma unroll
        for (int c = 0; c < PF_MAXC; ++c) wcol[c] = c < DC ? A.wt[c * A.units + u] : 0.f;
        const float b = A.bias[u];
        float vmax = -3.0e38f;
        // only the pillar's REAL points are multiplied (a KITTI pillar holds ~2-5 of its 32 rows): a masked row is all zeros
        // (point_pillars.py:546-549), its fma chain returns the bias bit for bit, so it contributes relu(bias) to the maximum
        const int nreal = np < A.P ? np : A.P;
        for (int j = 0; j < nreal; ++j) {
            const float* d = D[w][j];
            float v = b;
#pragma unroll
            for (int c = 0; c < PF_MAXC; ++c) if (c < DC) v = fmaf(d[c], wcol[c], v);
            v = v > 0.f ? v : 0.f;
            vmax = v > vmax ? v : vmax;
            if (!A.last) A.xcat[(m * A.P + j) * (2 * A.units) + u] = v;
        }
        if (nreal < A.P) {
            const float vb = b > 0.f ? b : 0.f;
            vmax = vb > vmax ? vb : vmax;
            if (!A.last) for (int j = nreal; j < A.P; ++j) A.xcat[(m * A.P + j) * (2 * A.units) + u] = vb;
        }
        if (A.last) {
            A.canvas[(((int64_t)sample * A.ny + cy) * A.nx + cx) * A.canvas_c + u] = vmax;
        } else {
            for (int j = 0; j < A.P; ++j) A.xcat[(m * A.P + j) * (2 * A.units) + A.units + u] = vmax;
        }
    }
}

// The KITTI-shaped case (4 input channels in 16-byte rows, one PFN layer of 64 units, <= 60 samples) with the pillar's DEPENDENT
// round trips cut from eight to three.  pillar_pfn spends its time waiting (67 k pillars of 8 sweeps in 100 us at full occupancy, a
// few hundred bytes each: profiles/r06_pillars_kernel_stats.csv): coords -> row splits -> point indices -> four scalar loads per
// point -> LDS -> nine weight loads -> a binary search over batch_splits -> store.  Here the weights are requested first (they do not
// depend on the pillar), ONE 32-bit and ONE 64-bit gather fetch coords, both row splits and the whole batch_splits table (lane =
// entry: the sample is a ballot count), a point is one float4 load, and the decorated rows never go through LDS: the row of point j
// reaches every lane as scalar operands (v_readlane with a uniform j).  Same operations in the same order as pillar_pfn -- the same bits.
__global__ void __launch_bounds__(256) pillar_pfn_v4(PfnArgs A) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t m = (int64_t)blockIdx.x * 4 + w;
    if (m >= A.n_pillars) return;                 // wave-uniform; no block barrier below
    float wcol[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) wcol[c] = A.wt[c * 64 + lane];
    const float b = A.bias[lane];
    int c32 = 0;
    if (lane < 2) c32 = A.coords[3 * m + lane];
    long long v64 = 0x7fffffffffffffffll;
    if (lane == 2 || lane == 3) v64 = A.prs[m + (lane - 2)];
    else if (lane >= 4 && lane < 3 + A.batch) v64 = A.batch_splits[lane - 3];         // entries 1 .. batch - 1
    const int cx = __builtin_amdgcn_readlane(c32, 0), cy = __builtin_amdgcn_readlane(c32, 1);
    if (cx >= A.nx || cy >= A.ny) return;         // pillar on the upper range bound (point_pillars.py:373-380)
    const int lo32 = (int)(unsigned)(v64 & 0xffffffffll), hi32 = (int)(v64 >> 32);
    const int64_t p0 = (int64_t)(unsigned)__builtin_amdgcn_readlane(lo32, 2) | ((int64_t)__builtin_amdgcn_readlane(hi32, 2) << 32);
    const int64_t p1 = (int64_t)(unsigned)__builtin_amdgcn_readlane(lo32, 3) | ((int64_t)__builtin_amdgcn_readlane(hi32, 3) << 32);
    const int sample = __popcll(__ballot(lane >= 4 && v64 <= (long long)m));
    const int np = (int)(p1 - p0);                // <= P
    const bool real = lane < np;
    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
    if (real) f = *reinterpret_cast<const float4*>(A.points + 4 * A.pidx[p0 + lane]);
    const float inv = 1.0f / (float)np;
    const float mx = wave_sum(f.x) * inv, my = wave_sum(f.y) * inv, mz = wave_sum(f.z) * inv;
    float d[9];
    d[0] = f.x; d[1] = f.y; d[2] = f.z; d[3] = f.w;
    d[4] = f.x - mx; d[5] = f.y - my; d[6] = f.z - mz;
    d[7] = f.x - ((float)cx * A.vx + A.x_off);
    d[8] = f.y - ((float)cy * A.vy + A.y_off);
    float vmax = -3.0e38f;
    const int nreal = np < A.P ? np : A.P;
    for (int j = 0; j < nreal; ++j) {
        float v = b;
#pragma unroll
        for (int c = 0; c < 9; ++c) v = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(d[c]), j)), wcol[c], v);
        v = v > 0.f ? v : 0.f;
        vmax = v > vmax ? v : vmax;
    }
    if (nreal < A.P) {                            // masked rows are all zeros: relu(bias), bit for bit (see pillar_pfn)
        const float vb = b > 0.f ? b : 0.f;
        vmax = vb > vmax ? vb : vmax;
    }
    A.canvas[(((int64_t)sample * A.ny + cy) * A.nx + cx) * A.canvas_c + lane] = vmax;
}

// subsequent PFN layers: input rows [M, P, cin] -> Linear + folded BN + ReLU -> max (-> canvas) or [x | max]
struct PfnNextArgs {
    const float* xin; int cin;
    const int32_t* coords; const int64_t* batch_splits; int batch; int64_t n_pillars; int P;
    int nx, ny;
    const float* wt; const float* bias; int units; int last;
    float* canvas; int canvas_c; float* xcat;
};

__global__ void __launch_bounds__(64) pfn_next(PfnNextArgs A) {
    HIP_DYNAMIC_SHARED(float, X)                  // [P][cin]
    const int lane = threadIdx.x;
    const int64_t m = blockIdx.x;
    const int cx = A.coords[3 * m], cy = A.coords[3 * m + 1];
    if (cx >= A.nx || cy >= A.ny) return;
    const float* src = A.xin + m * (int64_t)A.P * A.cin;
    for (int e = lane; e < A.P * A.cin; e += 64) X[e] = src[e];
    __syncthreads();
    const int sample = pillar_sample(A.batch_splits, A.batch, m);
    for (int u = lane; u < A.units; u += 64) {
        const float b = A.bias[u];
        float vmax = -3.0e38f;
        for (int j = 0; j < A.P; ++j) {
            float v = b;
            for (int c = 0; c < A.cin; ++c) v = fmaf(X[j * A.cin + c], A.wt[c * A.units + u], v);
            v = v > 0.f ? v : 0.f;
            vmax = v > vmax ? v : vmax;
            if (!A.last) A.xcat[(m * A.P + j) * (2 * A.units) + u] = v;
        }
        if (A.last) A.canvas[(((int64_t)sample * A.ny + cy) * A.nx + cx) * A.canvas_c + u] = vmax;
        else
            for (int j = 0; j < A.P; ++j) A.xcat[(m * A.P + j) * (2 * A.units) + A.units + u] = vmax;
    }
}

// NHWC [B*H*W, ld] channel slice [c0, c0 + C) -> NCHW [B, C, H, W] through an LDS tile transpose
__global__ void __launch_bounds__(256)
head_to_nchw(const float* __restrict__ in, int64_t ld, int c0, int C, int64_t hw, float* __restrict__ out) {
    __shared__ float T[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int64_t p0 = (int64_t)blockIdx.x * 32;                  // pixel tile (within one sample)
    const int cb = blockIdx.y * 32;
    const int64_t b = blockIdx.z;
    for (int r = ty; r < 32; r += 8) {
        const int64_t p = p0 + r;
        const int c = cb + tx;
        T[r][tx] = (p < hw && c < C) ? in[(b * hw + p) * ld + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = cb + r;
        const int64_t p = p0 + tx;
        if (c < C && p < hw) out[(b * C + c) * hw + p] = T[tx][r];
    }
}

static inline size_t pl_align(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace ml3d

using namespace ml3d;

extern "C" size_t ml3d_pillar_features_workspace_bytes(int64_t n_pillars, int max_num_points, int num_layers,
                                                       const int32_t* units_host) {
    if (n_pillars < 0 || max_num_points <= 0 || num_layers <= 0 || !units_host) return 0;
    size_t widest = 0;
    for (int i = 0; i + 1 < num_layers; ++i) {
        size_t wdt = 2 * (size_t)units_host[i];
        widest = wdt > widest ? wdt : widest;
    }
    // two ping-pong [M, P, 2U] buffers for multi-layer PFNs, nothing for the single-layer case
    return 2 * pl_align(sizeof(float) * (size_t)(n_pillars > 0 ? n_pillars : 1) * max_num_points * widest) + 512;
}

extern "C" int ml3d_pillar_features(const float* points, int64_t point_stride, int in_channels,
                                    const int32_t* voxel_coords, const int64_t* point_indices,
                                    const int64_t* point_row_splits, const int64_t* batch_splits, int64_t batch,
                                    int64_t n_pillars, int max_num_points, float vx, float vy, float x_offset,
                                    float y_offset, int nx, int ny, int num_layers, const int32_t* units_host,
                                    const float* const* weights_host, const float* const* bias_host, float* canvas,
                                    int canvas_channels, void* workspace, size_t workspace_bytes, void* stream) {
    if (batch <= 0 || n_pillars < 0 || in_channels < 3 || point_stride < in_channels || nx <= 0 || ny <= 0 ||
        num_layers <= 0 || !units_host || !weights_host || !bias_host || !canvas || max_num_points <= 0)
        return ML3D_E_INVALID;
    if (in_channels > 8 || max_num_points > PF_MAXP || num_layers > 4) return ML3D_E_UNSUPPORTED;
    if (units_host[num_layers - 1] != canvas_channels) return ML3D_E_INVALID;
    hipStream_t st = (hipStream_t)stream;
    zero_async(canvas, sizeof(float) * (size_t)batch * ny * nx * canvas_channels, st);        // (a fill kernel, not hipMemsetAsync: grid.h)
    if (n_pillars == 0) return 0;
    if (!points || !voxel_coords || !point_indices || !point_row_splits || !batch_splits) return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_pillar_features_workspace_bytes(n_pillars, max_num_points, num_layers, units_host))
        return ML3D_E_WORKSPACE;
    size_t widest = 0;
    for (int i = 0; i + 1 < num_layers; ++i) widest = 2 * (size_t)units_host[i] > widest ? 2 * (size_t)units_host[i] : widest;
    char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float* buf[2];
    buf[0] = (float*)p;
    buf[1] = (float*)(p + pl_align(sizeof(float) * (size_t)n_pillars * max_num_points * widest));
    PfnArgs a;
    a.points = points; a.point_stride = point_stride; a.in_ch = in_channels;
    a.coords = voxel_coords; a.pidx = point_indices; a.prs = point_row_splits;
    a.batch_splits = batch_splits; a.batch = (int)batch; a.n_pillars = n_pillars; a.P = max_num_points;
    a.vx = vx; a.vy = vy; a.x_off = x_offset; a.y_off = y_offset; a.nx = nx; a.ny = ny;
    a.wt = weights_host[0]; a.bias = bias_host[0]; a.units = units_host[0];
    a.last = num_layers == 1 ? 1 : 0;
    a.canvas = canvas; a.canvas_c = canvas_channels; a.xcat = buf[0];
    static const bool v4_off = [] { const char* e = getenv("ML3D_PFN_V4"); return e && atoi(e) == 0; }();
    if (!v4_off && in_channels == 4 && point_stride == 4 && ((uintptr_t)points & 15) == 0 && num_layers == 1 && units_host[0] == 64 &&
        batch <= 60)
        hipLaunchKernelGGL(pillar_pfn_v4, dim3((unsigned)((n_pillars + 3) / 4)), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(pillar_pfn, dim3((unsigned)((n_pillars + 3) / 4)), dim3(256), 0, st, a);
    if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    int cin = 2 * units_host[0];
    for (int l = 1; l < num_layers; ++l) {
        PfnNextArgs n;
        n.xin = buf[(l - 1) & 1]; n.cin = cin;
        n.coords = voxel_coords; n.batch_splits = batch_splits; n.batch = (int)batch; n.n_pillars = n_pillars;
        n.P = max_num_points; n.nx = nx; n.ny = ny;
        n.wt = weights_host[l]; n.bias = bias_host[l]; n.units = units_host[l];
        n.last = l == num_layers - 1 ? 1 : 0;
        n.canvas = canvas; n.canvas_c = canvas_channels; n.xcat = buf[l & 1];
        size_t sm = sizeof(float) * (size_t)max_num_points * cin;
        if (sm > 64 * 1024) return ML3D_E_UNSUPPORTED;
        hipLaunchKernelGGL(pfn_next, dim3((unsigned)n_pillars), dim3(64), sm, st, n);
        if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
        cin = 2 * units_host[l];
    }
    return 0;
}

extern "C" size_t ml3d_conv2d_workspace_bytes(int64_t batch, int out_h, int out_w, int cin, int cout, int kh, int kw) {
    if (batch <= 0 || out_h <= 0 || out_w <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0) return 0;
    return gemm_partial_bytes(batch * out_h * out_w, cout, kh * kw * cin) + 512;
}

extern "C" int ml3d_conv2d_nhwc(const float* in, int64_t batch, int h, int w, int cin, const float* weights,
                                const float* bias, int kh, int kw, int stride, int pad, int act, float slope,
                                int cout, float* out, int64_t out_pixel_stride, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 ||
        !in || !weights || !out || out_pixel_stride < cout)
        return ML3D_E_INVALID;
    if (cin & 3) return ML3D_E_UNSUPPORTED;
    ConvA A;
    A.in = in; A.B = (int)batch; A.H = h; A.W = w; A.C = cin;
    A.OH = (h + 2 * pad - kh) / stride + 1;
    A.OW = (w + 2 * pad - kw) / stride + 1;
    A.KH = kh; A.KW = kw; A.stride = stride; A.pad = pad;
    if (A.OH <= 0 || A.OW <= 0) return ML3D_E_INVALID;
    Epilogue ep = {bias, nullptr, 0, act, slope, 0, 0, 0, 0};
    char* p = workspace ? (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255) : nullptr;
    size_t avail = workspace && workspace_bytes > 256 ? workspace_bytes - 256 : 0;
    return gemm_conv(A, weights, cout, ep, out, out_pixel_stride, p, avail, (hipStream_t)stream);
}

// ---- the same convolution on the bf16 matrix pipe (gemm.h: three-way bf16 splits, float32-equivalent result) ---------------
extern "C" size_t ml3d_gemm_pack_bf16x3_bytes(int k, int n) { return gemm_pack_bf16x3_bytes(k, n); }

extern "C" int ml3d_gemm_pack_bf16x3(const float* weights, int k, int n, void* packed, size_t packed_bytes, void* stream) {
    if (!weights || !packed || k <= 0 || n <= 0) return ML3D_E_INVALID;
    if (k % 32) return ML3D_E_UNSUPPORTED;
    if (packed_bytes < gemm_pack_bf16x3_bytes(k, n)) return ML3D_E_WORKSPACE;
    return gemm_pack_bf16x3(weights, k, n, packed, (hipStream_t)stream);
}

extern "C" int ml3d_conv2d_nhwc_bf16x3(const float* in, int64_t batch, int h, int w, int cin, const void* packed,
                                       const float* bias, int kh, int kw, int stride, int pad, int act, float slope,
                                       int cout, float* out, int64_t out_pixel_stride, void* stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 ||
        !in || !packed || !out || out_pixel_stride < cout)
        return ML3D_E_INVALID;
    ConvA A;
    A.in = in; A.B = (int)batch; A.H = h; A.W = w; A.C = cin;
    A.OH = (h + 2 * pad - kh) / stride + 1;
    A.OW = (w + 2 * pad - kw) / stride + 1;
    A.KH = kh; A.KW = kw; A.stride = stride; A.pad = pad;
    if (A.OH <= 0 || A.OW <= 0) return ML3D_E_INVALID;
    if (!gemm_conv_bf16x3_ok(A)) return ML3D_E_UNSUPPORTED;
    Epilogue ep = {bias, nullptr, 0, act, slope, 0, 0, 0, 0};
    return gemm_conv_bf16x3(A, packed, cout, ep, out, out_pixel_stride, (hipStream_t)stream);
}

extern "C" int ml3d_deconv2d_nhwc(const float* in, int64_t batch, int h, int w, int cin, const float* weights,
                                  const float* bias, int stride, int act, float slope, int cout, float* out,
                                  int64_t out_pixel_stride, void* workspace, size_t workspace_bytes, void* stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || stride <= 0 || !in || !weights || !out ||
        out_pixel_stride < cout)
        return ML3D_E_INVALID;
    RowsA A;
    A.a = in; A.lda = cin; A.k1 = cin;
    A.gather = nullptr; A.gather_stride = 0; A.a_rows = batch * h * w;
    A.a2 = nullptr; A.lda2 = 0; A.k2 = 0;
    A.gather_on_a2 = 0; A.g_rows_per_item = 0; A.g_src_rows_per_item = 0;
    Epilogue ep = {bias, nullptr, 0, act, slope, stride, h, w, cout};
    char* p = workspace ? (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255) : nullptr;
    size_t avail = workspace && workspace_bytes > 256 ? workspace_bytes - 256 : 0;
    return gemm_rows(A, weights, batch * h * w, stride * stride * cout, cin, ep, out, out_pixel_stride, p, avail,
                     (hipStream_t)stream);
}

extern "C" int ml3d_deconv2d_nhwc_bf16x3(const float* in, int64_t batch, int h, int w, int cin, const void* packed,
                                         const float* bias, int stride, int act, float slope, int cout, float* out,
                                         int64_t out_pixel_stride, void* stream) {
    if (batch <= 0 || h <= 0 || w <= 0 || cin <= 0 || cout <= 0 || stride <= 0 || !in || !packed || !out ||
        out_pixel_stride < cout)
        return ML3D_E_INVALID;
    Epilogue ep = {bias, nullptr, 0, act, slope, stride, h, w, cout};
    return gemm_rows_bf16x3(in, cin, cin, nullptr, 0, 0, batch * h * w, packed, stride * stride * cout, ep, out, out_pixel_stride,
                            nullptr, 0, (hipStream_t)stream);
}

extern "C" size_t ml3d_linear_bf16x3_workspace_bytes(int64_t rows, int n, int k) { return gemm_partial_bytes_bf16x3(rows, n, k) + 512; }

extern "C" int ml3d_linear_bf16x3(const float* a, int64_t lda, int k1, const float* a2, int64_t lda2, int k2, int64_t rows,
                                  const void* packed, const float* bias, const float* residual, int64_t ldr, int n, int act,
                                  float slope, float* out, int64_t ldc, void* workspace, size_t workspace_bytes, void* stream) {
    if (rows < 0 || k1 <= 0 || k2 < 0 || n <= 0 || lda < k1 || (k2 > 0 && (!a2 || lda2 < k2)) || ldc < n || !a || !packed || !out ||
        act < 0 || act > 2 || (residual && ldr < n))
        return ML3D_E_INVALID;
    Epilogue ep = {bias, residual, ldr, act, slope, 0, 0, 0, 0};
    char* p = workspace ? (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255) : nullptr;
    const size_t avail = workspace && workspace_bytes > 256 ? workspace_bytes - 256 : 0;
    return gemm_rows_bf16x3(a, lda, k1, a2, lda2, k2, rows, packed, n, ep, out, ldc, p, avail, (hipStream_t)stream);
}

// the same with a GATHERED residual (ABI 12): residual row of output row m = residual_gather[m * stride] (a global row index; rows
// outside [0, residual_rows) add nothing) -- KPFCNN's decoder step split by linearity, (x W_x)[up[:, 0]] + skip W_skip (kpconv.py:283-286)
extern "C" int ml3d_linear_bf16x3_gathered(const float* a, int64_t lda, int k1, const float* a2, int64_t lda2, int k2, int64_t rows,
                                           const void* packed, const float* bias, const float* residual, int64_t ldr,
                                           const int32_t* residual_gather, int64_t residual_gather_stride, int64_t residual_rows, int n,
                                           int act, float slope, float* out, int64_t ldc, void* workspace, size_t workspace_bytes,
                                           void* stream) {
    if (rows < 0 || k1 <= 0 || k2 < 0 || n <= 0 || lda < k1 || (k2 > 0 && (!a2 || lda2 < k2)) || ldc < n || !a || !packed || !out ||
        act < 0 || act > 2 || (residual && ldr < n) || (residual_gather && (!residual || residual_gather_stride < 1 || residual_rows < 0)))
        return ML3D_E_INVALID;
    Epilogue ep = {bias, residual, ldr, act, slope, 0, 0, 0, 0};
    if (residual_gather) {      // global row indices: one "item" spanning every row
        ep.res_gather = residual_gather; ep.rg_rows_per_item = (int64_t)1 << 62; ep.rg_src_rows_per_item = 0;
        ep.rg_stride = residual_gather_stride; ep.rg_limit = residual_rows;
    }
    char* p = workspace ? (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255) : nullptr;
    const size_t avail = workspace && workspace_bytes > 256 ? workspace_bytes - 256 : 0;
    return gemm_rows_bf16x3(a, lda, k1, a2, lda2, k2, rows, packed, n, ep, out, ldc, p, avail, (hipStream_t)stream);
}

extern "C" int ml3d_nhwc_to_nchw(const float* in, int64_t in_pixel_stride, int channel_offset, int channels,
                                 int64_t batch, int64_t hw, float* out, void* stream) {
    if (batch <= 0 || hw <= 0 || channels <= 0 || channel_offset < 0 || !in || !out ||
        in_pixel_stride < channel_offset + channels)
        return ML3D_E_INVALID;
    dim3 grid((unsigned)((hw + 31) / 32), (unsigned)((channels + 31) / 32), (unsigned)batch);
    hipLaunchKernelGGL(head_to_nchw, grid, dim3(256), 0, (hipStream_t)stream, in, in_pixel_stride, channel_offset,
                       channels, hw, out);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}
