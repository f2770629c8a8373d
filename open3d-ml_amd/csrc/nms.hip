// nms.hip — rotated-BEV non-maximum suppression (gfx950).
//
// Replaces open3d.ml.torch.ops.nms(boxes[N,5] (x0,y0,x1,y1,r), scores[N], thr) as called from
// multiclass_nms (ml3d/torch/utils/objdet_helper.py:316-350) for Anchor3DHead.get_bboxes
// (ml3d/torch/models/point_pillars.py:945-1025).  Greedy by descending score (ties: lower index first),
// suppress when IoU > thr, kept indices returned in descending-score order — the oracle's contract, with
// the same float32 operation order for corners / Sutherland-Hodgman clipping / shoelace area (no fma).
//
// (1) order = stable radix sort of (descending score, index);  (2) one 64-thread workgroup per 64x64 block
// of the upper-triangular pair matrix writes a suppression bit mask (the IoU of a pair is computed once);
// (3) one wave walks the candidates in order, OR-ing mask rows of kept boxes into a register-resident
// "removed" bit set (lane w owns words w, w+64, ...).  N is at most nms_pre (100..4096 in the reference's
// configs), so this is latency- not bandwidth-bound; bytes: 20 N read, N^2/8 mask.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "grid.h"
#include "ml3d_hip.h"
#include "sort.h"

namespace ml3d {

struct P2 { float x, y; };

__device__ __forceinline__ float cross2(P2 a, P2 b) { return __fsub_rn(__fmul_rn(a.x, b.y), __fmul_rn(a.y, b.x)); }

__device__ __forceinline__ void box_corners(const float* b, P2* c) {
    const float cx = __fmul_rn(__fadd_rn(b[0], b[2]), 0.5f), cy = __fmul_rn(__fadd_rn(b[1], b[3]), 0.5f);
    const float w = __fsub_rn(b[2], b[0]), h = __fsub_rn(b[3], b[1]);
    const float cs = cosf(b[4]), sn = sinf(b[4]);
    const float hx[4] = {-0.5f * w, 0.5f * w, 0.5f * w, -0.5f * w};
    const float hy[4] = {-0.5f * h, -0.5f * h, 0.5f * h, 0.5f * h};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[i].x = __fsub_rn(__fadd_rn(cx, __fmul_rn(hx[i], cs)), __fmul_rn(hy[i], sn));
        c[i].y = __fadd_rn(__fadd_rn(cy, __fmul_rn(hx[i], sn)), __fmul_rn(hy[i], cs));
    }
}

__device__ float poly_intersection_area(const P2* A, const P2* B) {
    P2 cur[16], nxt[16];
    int nc = 4;
    for (int i = 0; i < 4; ++i) cur[i] = A[i];
    for (int e = 0; e < 4 && nc > 0; ++e) {
        const P2 p0 = B[e], p1 = B[(e + 1) & 3];
        const P2 ed = {__fsub_rn(p1.x, p0.x), __fsub_rn(p1.y, p0.y)};
        int nn = 0;
        for (int i = 0; i < nc; ++i) {
            const P2 s = cur[i], t = cur[(i + 1) % nc];
            const P2 vs = {__fsub_rn(s.x, p0.x), __fsub_rn(s.y, p0.y)}, vt = {__fsub_rn(t.x, p0.x), __fsub_rn(t.y, p0.y)};
            const float ds = cross2(ed, vs), dt = cross2(ed, vt);
            if (ds >= 0.f) nxt[nn++] = s;
            if ((ds >= 0.f) != (dt >= 0.f)) {
                const float u = __fdiv_rn(ds, __fsub_rn(ds, dt));
                P2 ip;
                ip.x = __fadd_rn(s.x, __fmul_rn(u, __fsub_rn(t.x, s.x)));
                ip.y = __fadd_rn(s.y, __fmul_rn(u, __fsub_rn(t.y, s.y)));
                nxt[nn++] = ip;
            }
        }
        nc = nn;
        for (int i = 0; i < nc; ++i) cur[i] = nxt[i];
    }
    if (nc < 3) return 0.f;
    float a = 0.f;
    for (int i = 0; i < nc; ++i) a = __fadd_rn(a, cross2(cur[i], cur[(i + 1) % nc]));
    return __fmul_rn(0.5f, fabsf(a));
}

__device__ __forceinline__ float iou_bev(const float* a, const P2* ca, const float* b) {
    P2 cb[4];
    box_corners(b, cb);
    const float ia = poly_intersection_area(ca, cb);
    const float aa = __fmul_rn(__fsub_rn(a[2], a[0]), __fsub_rn(a[3], a[1]));
    const float ab = __fmul_rn(__fsub_rn(b[2], b[0]), __fsub_rn(b[3], b[1]));
    const float un = __fsub_rn(__fadd_rn(aa, ab), ia);
    return un > 1e-8f ? __fdiv_rn(ia, un) : 0.f;
}

// ---- pairwise box IoU for the detection metric ---------------------------------------------------
// iou_bev(boxes_a [N,5], boxes_b [M,5]) / iou_3d(boxes_a [N,7], boxes_b [M,7]) -> [N,M]
//   ml3d/metrics/mAP.py:85-88: iou_bev(bbox[:, [0, 2, 3, 5, 6]]) = (x, z, w, l, yaw) in the camera frame,
//                               iou_3d(bbox[:, :7])               = (x, y, z, w, h, l, yaw), y = bottom face.
// A centre/size box is turned into the (x0, y0, x1, y1, r) form of the NMS above and clipped by the same routine, so the
// two ops share one definition of the rotated intersection; one thread per (a, b) pair.
__device__ __forceinline__ void center_to_corner_form(float cx, float cy, float dx, float dy, float r, float* o) {
    const float hx = __fmul_rn(0.5f, dx), hy = __fmul_rn(0.5f, dy);
    o[0] = __fsub_rn(cx, hx); o[1] = __fsub_rn(cy, hy); o[2] = __fadd_rn(cx, hx); o[3] = __fadd_rn(cy, hy); o[4] = r;
}

__global__ void __launch_bounds__(256)
iou_pairs(const float* __restrict__ A, const float* __restrict__ B, int64_t n, int64_t m, int mode3d, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * m) return;
    const int64_t i = t / m, j = t - i * m;
    float a[5], b[5];
    if (!mode3d) {
        const float* pa = A + 5 * i; const float* pb = B + 5 * j;
        center_to_corner_form(pa[0], pa[1], pa[2], pa[3], pa[4], a);
        center_to_corner_form(pb[0], pb[1], pb[2], pb[3], pb[4], b);
    } else {
        const float* pa = A + 7 * i; const float* pb = B + 7 * j;
        center_to_corner_form(pa[0], pa[2], pa[3], pa[5], pa[6], a);       // ground plane of the camera frame: (x, z)
        center_to_corner_form(pb[0], pb[2], pb[3], pb[5], pb[6], b);
    }
    P2 ca[4], cb[4];
    box_corners(a, ca);
    box_corners(b, cb);
    const float inter = poly_intersection_area(ca, cb);
    const float aa = __fmul_rn(__fsub_rn(a[2], a[0]), __fsub_rn(a[3], a[1]));
    const float ab = __fmul_rn(__fsub_rn(b[2], b[0]), __fsub_rn(b[3], b[1]));
    float num = inter, den;
    if (!mode3d) {
        den = __fsub_rn(__fadd_rn(aa, ab), inter);
    } else {
        // y axis points down: the box spans [y - h, y]
        const float* pa = A + 7 * i; const float* pb = B + 7 * j;
        const float top = fmaxf(__fsub_rn(pa[1], pa[4]), __fsub_rn(pb[1], pb[4])), bot = fminf(pa[1], pb[1]);
        const float oh = fmaxf(__fsub_rn(bot, top), 0.f);
        num = __fmul_rn(inter, oh);
        den = __fsub_rn(__fadd_rn(__fmul_rn(aa, pa[4]), __fmul_rn(ab, pb[4])), num);
    }
    out[t] = den > 1e-8f ? __fdiv_rn(num, den) : 0.f;
}

// key = ~ordered(score) << 32 | index : ascending key == descending score, ties by ascending index
__global__ void nms_keys(const float* __restrict__ scores, int64_t n, u64* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((u64)(~f2ord(scores[i])) << 32) | (u64)(uint32_t)i;
    vals[i] = (uint32_t)i;
}

// mask[a][wb] bit j: sorted candidate (64*wb + j) is suppressed by sorted candidate a  (only b > a)
__global__ void __launch_bounds__(64)
nms_mask(const float* __restrict__ boxes, const uint32_t* __restrict__ order, int64_t n, float thr, int words,
         u64* __restrict__ mask) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;
    __shared__ float bb[64][5];
    const int t = threadIdx.x;
    const int64_t bj = (int64_t)cb * 64 + t;
    if (bj < n) {
        const float* s = boxes + 5 * (int64_t)order[bj];
        for (int k = 0; k < 5; ++k) bb[t][k] = s[k];
    }
    __syncthreads();
    const int64_t a = (int64_t)rb * 64 + t;
    if (a >= n) return;
    float ba[5];
    const float* s = boxes + 5 * (int64_t)order[a];
    for (int k = 0; k < 5; ++k) ba[k] = s[k];
    P2 ca[4];
    box_corners(ba, ca);
    u64 bits = 0ull;
    const int cols = (int)((n - (int64_t)cb * 64) < 64 ? (n - (int64_t)cb * 64) : 64);
    for (int j = (rb == cb ? t + 1 : 0); j < cols; ++j)
        if (iou_bev(ba, ca, bb[j]) > thr) bits |= 1ull << j;
    mask[a * words + cb] = bits;
}

__global__ void __launch_bounds__(64)
nms_reduce(const u64* __restrict__ mask, const uint32_t* __restrict__ order, int64_t n, int words,
           int64_t* __restrict__ keep, int64_t* __restrict__ count) {
    // lane l owns words l, l + 64, ... of the `removed` set (up to 16 words per lane = 65536 candidates)
    const int lane = threadIdx.x;
    u64 removed[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) removed[k] = 0ull;
    int64_t m = 0;
    for (int64_t a = 0; a < n; ++a) {
        const int w = (int)(a >> 6);
        u64 word = 0ull;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (k == (w >> 6)) word = removed[k];
        word = __shfl(word, w & 63);
        if ((word >> (a & 63)) & 1ull) continue;         // wave-uniform
        if (lane == 0) keep[m] = (int64_t)order[a];
        ++m;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int ww = lane + 64 * k;
            if (ww < words && ww >= w) removed[k] |= mask[a * words + ww];
        }
    }
    if (lane == 0) *count = m;
}

static inline size_t nms_align(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace ml3d

using namespace ml3d;

extern "C" size_t ml3d_nms_workspace_bytes(int64_t n) {
    if (n < 0) return 0;
    const int64_t m = n > 0 ? n : 1;
    const int64_t words = (m + 63) / 64;
    return nms_align(sizeof(u64) * (size_t)m) + nms_align(sizeof(uint32_t) * (size_t)m) +
           nms_align(sizeof(u64) * (size_t)(m * words)) + sort_ws_bytes(m) + 512;
}

extern "C" int ml3d_nms(const float* boxes, const float* scores, int64_t n, float iou_threshold, int64_t* out_keep,
                        int64_t* out_count, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || !out_count) return ML3D_E_INVALID;
    if (n > 65536) return ML3D_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { (void)hipMemsetAsync(out_count, 0, sizeof(int64_t), st); return 0; }
    if (!boxes || !scores || !out_keep) return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_nms_workspace_bytes(n)) return ML3D_E_WORKSPACE;
    const int words = (int)((n + 63) / 64);
    char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    u64* keys = (u64*)p;            p += nms_align(sizeof(u64) * (size_t)n);
    uint32_t* order = (uint32_t*)p; p += nms_align(sizeof(uint32_t) * (size_t)n);
    u64* mask = (u64*)p;            p += nms_align(sizeof(u64) * (size_t)(n * words));
    SortWs sw;
    if (!sort_ws_carve(p, sort_ws_bytes(n), n, &sw)) return ML3D_E_WORKSPACE;
    hipLaunchKernelGGL(nms_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, scores, n, keys, order);
    if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    if (sort_pairs_u64(keys, order, n, 64, sw, st)) return ML3D_E_LAUNCH;
    (void)hipMemsetAsync(mask, 0, sizeof(u64) * (size_t)(n * words), st);
    hipLaunchKernelGGL(nms_mask, dim3((unsigned)words, (unsigned)words), dim3(64), 0, st, boxes, order, n, iou_threshold,
                       words, mask);
    if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    hipLaunchKernelGGL(nms_reduce, dim3(1), dim3(64), 0, st, mask, order, n, words, out_keep, out_count);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_iou_bev(const float* boxes_a, const float* boxes_b, int64_t n, int64_t m, float* out_iou, void* stream) {
    if (n < 0 || m < 0) return ML3D_E_INVALID;
    if (n == 0 || m == 0) return 0;
    if (!boxes_a || !boxes_b || !out_iou) return ML3D_E_INVALID;
    hipLaunchKernelGGL(ml3d::iou_pairs, dim3((unsigned)((n * m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes_a, boxes_b,
                       n, m, 0, out_iou);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_iou_3d(const float* boxes_a, const float* boxes_b, int64_t n, int64_t m, float* out_iou, void* stream) {
    if (n < 0 || m < 0) return ML3D_E_INVALID;
    if (n == 0 || m == 0) return 0;
    if (!boxes_a || !boxes_b || !out_iou) return ML3D_E_INVALID;
    hipLaunchKernelGGL(ml3d::iou_pairs, dim3((unsigned)((n * m + 255) / 256)), dim3(256), 0, (hipStream_t)stream, boxes_a, boxes_b,
                       n, m, 1, out_iou);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}
