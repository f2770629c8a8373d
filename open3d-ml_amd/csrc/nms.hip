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

// Sutherland-Hodgman: the convex quadrilateral A clipped by the four half-planes of B, then the shoelace area.  An edge appends 0, 1 or
// 2 vertices per input vertex, i.e. the output position is data dependent.  History: as local arrays the two vertex lists went to
// SCRATCH (round 3: every vertex access a global-memory round trip); as lane-interleaved LDS lists (round 4-5) the kernel was fast
// but its results were NOT stable under co-running kernels: with the bf16x3 convolutions of another stream on the same CUs (62.5 KB of
// LDS per workgroup, two per CU) nmsb_mask returned different mask bits for lanes 48..63 in ~45 % of the launches, on identical inputs
// (profiles/r05_nms_corun_diagnosis.md; quiet GPU, f32 convolutions, rocBLAS, fills as co-runners: never).  The lists therefore live
// in REGISTERS now: every index is a compile-time constant after unrolling, the data-dependent append is a select per slot that can
// be its target (vertex i of an edge can only land in slots 0 .. 2 i + 1).  Arithmetic and its order are unchanged (the oracle's).
constexpr int POLY_MAX = 16;
struct PolyList { float x[POLY_MAX], y[POLY_MAX]; };

// L[at] = v for a data-dependent `at` < SLOTS
template <int SLOTS>
__device__ __forceinline__ void poly_put(PolyList& L, int at, P2 v) {
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        const bool hit = q == at;
        L.x[q] = hit ? v.x : L.x[q];
        L.y[q] = hit ? v.y : L.y[q];
    }
}

// one input vertex of one clipping edge: s = cur[I], t = its successor; appends to nxt at nn.  Returns false when the list would overflow.
template <int I>
__device__ __forceinline__ bool poly_clip_vertex(const PolyList& cur, int nc, PolyList& nxt, int& nn, P2 p0, P2 ed) {
    // a vertex appends at most two; convex input never passes 8, but inf / NaN corners can alternate the sign test
    // (4 -> 6 -> 9 -> 13 -> 19): such a pair has no meaningful overlap -- area 0 (the oracle's rule too)
    if (nn + 2 > POLY_MAX) return false;
    const P2 s = {cur.x[I], cur.y[I]};
    constexpr int J = I + 1 < POLY_MAX ? I + 1 : 0;
    const bool wrap = I + 1 == nc;
    const P2 t = {wrap ? cur.x[0] : cur.x[J], wrap ? cur.y[0] : cur.y[J]};
    const P2 vs = {__fsub_rn(s.x, p0.x), __fsub_rn(s.y, p0.y)}, vt = {__fsub_rn(t.x, p0.x), __fsub_rn(t.y, p0.y)};
    const float ds = cross2(ed, vs), dt = cross2(ed, vt);
    constexpr int SLOTS = 2 * I + 2 < POLY_MAX ? 2 * I + 2 : POLY_MAX;
    if (ds >= 0.f) { poly_put<SLOTS>(nxt, nn, s); ++nn; }
    if ((ds >= 0.f) != (dt >= 0.f)) {
        const float u = __fdiv_rn(ds, __fsub_rn(ds, dt));
        P2 ip;
        ip.x = __fadd_rn(s.x, __fmul_rn(u, __fsub_rn(t.x, s.x)));
        ip.y = __fadd_rn(s.y, __fmul_rn(u, __fsub_rn(t.y, s.y)));
        poly_put<SLOTS>(nxt, nn, ip);
        ++nn;
    }
    return true;
}

template <int I>
__device__ __forceinline__ bool poly_clip_from(const PolyList& cur, int nc, PolyList& nxt, int& nn, P2 p0, P2 ed) {
    if constexpr (I < POLY_MAX) {
        if (I < nc) {
            if (!poly_clip_vertex<I>(cur, nc, nxt, nn, p0, ed)) return false;
            return poly_clip_from<I + 1>(cur, nc, nxt, nn, p0, ed);
        }
    }
    return true;
}

__device__ __forceinline__ float poly_intersection_area(const P2* A, const P2* B) {
    PolyList cur, nxt;
#pragma unroll
    for (int i = 0; i < POLY_MAX; ++i) { cur.x[i] = i < 4 ? A[i].x : 0.f; cur.y[i] = i < 4 ? A[i].y : 0.f; nxt.x[i] = 0.f; nxt.y[i] = 0.f; }
    int nc = 4;
#pragma unroll 1
    for (int e = 0; e < 4; ++e) {
        if (nc > 0) {
            // (selects, not B[e]: a dynamically indexed local array is promoted to LDS by the compiler)
            const P2 p0 = e == 0 ? B[0] : e == 1 ? B[1] : e == 2 ? B[2] : B[3];
            const P2 p1 = e == 0 ? B[1] : e == 1 ? B[2] : e == 2 ? B[3] : B[0];
            const P2 ed = {__fsub_rn(p1.x, p0.x), __fsub_rn(p1.y, p0.y)};
            int nn = 0;
            if (!poly_clip_from<0>(cur, nc, nxt, nn, p0, ed)) return 0.f;
            nc = nn;
#pragma unroll
            for (int i = 0; i < POLY_MAX; ++i) { cur.x[i] = nxt.x[i]; cur.y[i] = nxt.y[i]; }
        }
    }
    if (nc < 3) return 0.f;
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < POLY_MAX; ++i) {
        if (i < nc) {
            const int j = i + 1 < POLY_MAX ? i + 1 : 0;
            const bool wrap = i + 1 == nc;
            const P2 s = {cur.x[i], cur.y[i]}, t = {wrap ? cur.x[0] : cur.x[j], wrap ? cur.y[0] : cur.y[j]};
            a = __fadd_rn(a, cross2(s, t));
        }
    }
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

__global__ void __launch_bounds__(64)
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
    const float sc = scores[i];
    keys[i] = ((u64)(~f2ord(sc == 0.f ? 0.f : sc)) << 32) | (u64)(uint32_t)i;      // (-0.0 and +0.0: one score, ties by index)
    vals[i] = (uint32_t)i;
}

// mask[a][wb] bit j: sorted candidate (64*wb + j) is suppressed by sorted candidate a  (only b > a).
// One wave per (a, word): lane j tests the pair (a, 64 wb + j) with both boxes in REGISTERS and the ballot is the mask word (the form
// of nmsb_mask).  Round 6: the earlier form -- a thread per row walking 64 boxes staged in LDS, with a per-lane first column on the
// diagonal blocks -- returned a different keep list in ~1 of 200 calls beside a looping bf16x3 convolution on another stream
// (tests/test_gpu_corun.py, profiles/r06_nms_corun.log): LDS reads inside a loop whose trip count differs per lane, the pattern of
// DESIGN.md section 9.10.  This form has no LDS and no loop.
__global__ void __launch_bounds__(64)
nms_mask(const float* __restrict__ boxes, const uint32_t* __restrict__ order, int64_t n, float thr, int words,
         u64* __restrict__ mask) {
    const int64_t a = (int64_t)blockIdx.z * 32768 + blockIdx.y;
    const int cb = blockIdx.x, lane = threadIdx.x;
    if (a >= n || cb < (int)(a >> 6)) return;
    const int64_t j = (int64_t)cb * 64 + lane;
    bool sup = false;
    if (j > a && j < n) {
        float ba[5], bj[5];
        const float* sa = boxes + 5 * (int64_t)order[a];
        const float* sj = boxes + 5 * (int64_t)order[j];
#pragma unroll
        for (int q = 0; q < 5; ++q) { ba[q] = sa[q]; bj[q] = sj[q]; }
        // circumscribed circles (rotation about the box centre): disjoint circles -> disjoint boxes -> IoU exactly 0, which is never
        // > thr when thr >= 0 (NaN / inf boxes and negative thresholds take the full test)
        const float dx = 0.5f * ((ba[0] + ba[2]) - (bj[0] + bj[2])), dy = 0.5f * ((ba[1] + ba[3]) - (bj[1] + bj[3]));
        const float ra = 0.5f * sqrtf((ba[2] - ba[0]) * (ba[2] - ba[0]) + (ba[3] - ba[1]) * (ba[3] - ba[1]));
        const float rj = 0.5f * sqrtf((bj[2] - bj[0]) * (bj[2] - bj[0]) + (bj[3] - bj[1]) * (bj[3] - bj[1]));
        const float reach = (ra + rj) * 1.0001f + 1e-6f;
        if (!(thr >= 0.f) || !(dx * dx + dy * dy > reach * reach)) {
            P2 ca[4];
            box_corners(ba, ca);
            sup = iou_bev(ba, ca, bj) > thr;
        }
    }
    const u64 bits = __ballot(sup);
    if (lane == 0) mask[a * words + cb] = bits;
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

// ---------------------------------------------------------------------------------------------------------------------
// Anchor3DHead.get_bboxes for a WHOLE BATCH without host read-backs (ml3d/torch/models/point_pillars.py:945-1025;
// BBoxCoder.decode, ml3d/torch/utils/objdet_helper.py:286-313; multiclass_nms, :316-350).  The per-sample, per-class loop of
// the reference (B x C calls of nms, each behind a nonzero() that synchronises) becomes four launches:
//   pp_anchor_scores   max over the classes of sigmoid(cls) per anchor            -> the key of the nms_pre top-k (ml3d_topk_rows, below)
//   pp_decode          the selected candidates: anchor + deltas -> boxes, class scores, direction bit, BEV corners form
//   nmsb_order / nmsb_mask / nmsb_reduce   P = B x C independent NMS problems over the k candidates of a sample: a candidate
//                      takes part in class c iff its score_c > score_thr (point_pillars.py:1001); greedy order = descending
//                      score, ties by ascending candidate index -- the order of ml3d_nms on the compacted list
//   pp_collect         kept candidates, class-major like the reference's torch.cat, yaw corrected by the direction classifier
// Head maps are read in the reference's NCHW layout (channel = a * C + c, a * 7 + t, a * 2 + d; anchors ordered (h, w, a)).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f32(float x) { return __fdiv_rn(1.0f, __fadd_rn(1.0f, expf(-x))); }

// A head map is addressed as base[b * sb + channel * sc + pixel * sp]: NCHW (sb = channels * HW, sc = HW, sp = 1, the
// reference's layout) or a channel slice of the fused NHWC head tensor (sb = HW * Ctot, sc = 1, sp = Ctot).
struct HeadMap { const float* base; int64_t sb, sc, sp; };
__device__ __forceinline__ float head_at(const HeadMap& m, int64_t b, int ch, int64_t hw) {
    return m.base[b * m.sb + (int64_t)ch * m.sc + hw * m.sp];
}

__global__ void __launch_bounds__(256)
pp_anchor_scores(HeadMap cls, int64_t B, int A, int C, int64_t HW, float* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * HW) return;
    const int64_t b = t / HW, hw = t - b * HW;
    for (int a = 0; a < A; ++a) {
        float m = -1.0f;
        for (int c = 0; c < C; ++c) m = fmaxf(m, sigmoid_f32(head_at(cls, b, a * C + c, hw)));
        out[(b * HW + hw) * A + a] = m;
    }
}

__global__ void __launch_bounds__(256)
pp_decode(HeadMap cls, HeadMap reg, HeadMap dir, const float* __restrict__ anchors, const int64_t* __restrict__ cand, int64_t B, int64_t k, int A, int C, int64_t HW,
          float* __restrict__ box, float* __restrict__ bev, float* __restrict__ score, int32_t* __restrict__ dirbit) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * k) return;
    const int64_t b = t / k, j = t - b * k;
    const int64_t ai = cand[t];
    const int64_t hw = ai / A;
    const int a = (int)(ai - hw * A);
    const float* an = anchors + 7 * ai;
    float d[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) d[q] = head_at(reg, b, a * 7 + q, hw);
    const float xa = an[0], ya = an[1], wa = an[3], la = an[4], ha = an[5], ra = an[6];
    const float za = __fadd_rn(an[2], __fdiv_rn(ha, 2.0f));
    const float diag = sqrtf(__fadd_rn(__fmul_rn(la, la), __fmul_rn(wa, wa)));
    const float hg = __fmul_rn(expf(d[5]), ha);
    float o[7];
    o[0] = __fadd_rn(__fmul_rn(d[0], diag), xa);
    o[1] = __fadd_rn(__fmul_rn(d[1], diag), ya);
    o[2] = __fsub_rn(__fadd_rn(__fmul_rn(d[2], ha), za), __fdiv_rn(hg, 2.0f));
    o[3] = __fmul_rn(expf(d[3]), wa);
    o[4] = __fmul_rn(expf(d[4]), la);
    o[5] = hg;
    o[6] = __fadd_rn(d[6], ra);
#pragma unroll
    for (int q = 0; q < 7; ++q) box[7 * t + q] = o[q];
    // xywhr_to_xyxyr(box3d_to_bev(.)) (objdet_helper.py:69-100): (x, y, w, l, r) -> corners form
    const float hwid = __fdiv_rn(o[3], 2.0f), hlen = __fdiv_rn(o[4], 2.0f);
    bev[5 * t + 0] = __fsub_rn(o[0], hwid); bev[5 * t + 1] = __fsub_rn(o[1], hlen);
    bev[5 * t + 2] = __fadd_rn(o[0], hwid); bev[5 * t + 3] = __fadd_rn(o[1], hlen); bev[5 * t + 4] = o[6];
    for (int c = 0; c < C; ++c) score[(b * C + c) * k + j] = sigmoid_f32(head_at(cls, b, a * C + c, hw));
    const float d0 = head_at(dir, b, a * 2, hw), d1 = head_at(dir, b, a * 2 + 1, hw);
    dirbit[t] = d1 > d0 ? 1 : 0;                 // torch.max(dim=-1)[1]: the first maximum
}

constexpr int NMSB_MAX = 4096;                   // candidates per problem (nms_pre of the reference's configs: 100 .. 4096)

// one workgroup per problem: rank of every participating candidate among the participating ones
__global__ void __launch_bounds__(256)
nmsb_order(const float* __restrict__ scores, int64_t n, float score_thr, uint32_t* __restrict__ order, int32_t* __restrict__ nvalid) {
    __shared__ u64 keys[NMSB_MAX];
    __shared__ int cnt;
    const int64_t p = blockIdx.x;
    if (threadIdx.x == 0) cnt = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const float sc = scores[p * n + i];
        keys[i] = sc > score_thr ? (((u64)(~f2ord(sc == 0.f ? 0.f : sc)) << 32) | (u64)(uint32_t)i) : ~0ull;
    }
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const u64 me = keys[i];
        if (me == ~0ull) continue;
        int r = 0;
        for (int64_t q = 0; q < n; ++q) r += keys[q] < me ? 1 : 0;
        order[p * n + r] = (uint32_t)i;
        atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) nvalid[p] = cnt;
}

// one wave per (sorted candidate a, 64-candidate word cb >= word of a): lane j tests the pair (a, 64 cb + j) and the ballot IS
// the mask word -- every rotated IoU of a problem in parallel (a first version gave a thread 64 pairs in a row: 0.72 ms per
// 16-sweep step for 48 problems of 100 candidates, latency-bound with the clipping arrays in scratch).  Two boxes whose
// circumscribed circles are disjoint cannot intersect: their IoU is exactly 0 and the clipping is skipped (most pairs).
__global__ void __launch_bounds__(64)
nmsb_mask(const float* __restrict__ bev, const uint32_t* __restrict__ order, const int32_t* __restrict__ nvalid, int64_t n,
          int C, float thr, int words, u64* __restrict__ mask) {
    const int64_t p = blockIdx.z, a = blockIdx.y;
    const int cb = blockIdx.x, lane = threadIdx.x;
    const int64_t nv = nvalid[p];
    if (a >= nv || cb < (int)(a >> 6)) return;
    const float* boxes = bev + (p / C) * n * 5;
    const uint32_t* ord = order + p * n;
    const int64_t j = (int64_t)cb * 64 + lane;
    bool sup = false;
    if (j > a && j < nv) {
        float ba[5], bj[5];
        const float* sa = boxes + 5 * (int64_t)ord[a];
        const float* sj = boxes + 5 * (int64_t)ord[j];
#pragma unroll
        for (int q = 0; q < 5; ++q) { ba[q] = sa[q]; bj[q] = sj[q]; }
        // circumscribed circles (rotation about the box centre): disjoint circles -> disjoint boxes -> IoU 0, never > thr >= 0
        const float dx = 0.5f * ((ba[0] + ba[2]) - (bj[0] + bj[2])), dy = 0.5f * ((ba[1] + ba[3]) - (bj[1] + bj[3]));
        const float ra = 0.5f * sqrtf((ba[2] - ba[0]) * (ba[2] - ba[0]) + (ba[3] - ba[1]) * (ba[3] - ba[1]));
        const float rj = 0.5f * sqrtf((bj[2] - bj[0]) * (bj[2] - bj[0]) + (bj[3] - bj[1]) * (bj[3] - bj[1]));
        const float reach = (ra + rj) * 1.0001f + 1e-6f;
        if (!(dx * dx + dy * dy > reach * reach)) {        // (NaN / inf boxes take the full test, like ml3d_nms)
            P2 ca[4];
            box_corners(ba, ca);
            sup = iou_bev(ba, ca, bj) > thr;
        }
    }
    const u64 bits = __ballot(sup);
    if (lane == 0) mask[(p * n + a) * words + cb] = bits;
}

// One wave walks a problem's candidates in order.  The decision for candidate a depends on the rows OR-ed in before it, the rows
// themselves do not: 64 rows at a time are staged in LDS with coalesced loads, so the walk reads LDS instead of taking one dependent
// global load per kept candidate (the first form: ~0.6 us per kept candidate, 59 us per launch at nms_pre = 100).  (Words below a row's
// own word are never written by nmsb_mask; they are staged along and, as before, never used.)
constexpr int NMSB_STAGE_ROWS = 64;

__global__ void __launch_bounds__(64)
nmsb_reduce(const u64* __restrict__ mask, const uint32_t* __restrict__ order, const int32_t* __restrict__ nvalid, int64_t n,
            int words, int32_t* __restrict__ keep, int32_t* __restrict__ count) {
    __shared__ u64 rows[NMSB_STAGE_ROWS * (NMSB_MAX / 64)];      // 32 KB: 64 rows of up to 64 words
    const int64_t p = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t nv = nvalid[p];
    const u64* mk = mask + p * n * words;
    const uint32_t* ord = order + p * n;
    u64 removed = 0ull;                           // lane l owns word l: 64 words = 4096 candidates (NMSB_MAX)
    int m = 0;
    for (int64_t a0 = 0; a0 < nv; a0 += NMSB_STAGE_ROWS) {
        const int here = (int)(nv - a0 < NMSB_STAGE_ROWS ? nv - a0 : NMSB_STAGE_ROWS);
        for (int e = lane; e < here * words; e += 64) rows[e] = mk[a0 * words + e];
        __syncthreads();
        for (int r = 0; r < here; ++r) {
            const int64_t a = a0 + r;
            const int w = (int)(a >> 6);
            const u64 word = __shfl(removed, w);
            if ((word >> (a & 63)) & 1ull) continue;  // wave-uniform
            if (lane == 0) keep[p * n + m] = (int32_t)ord[a];
            ++m;
            if (lane < words && lane >= w) removed |= rows[r * words + lane];
        }
        __syncthreads();
    }
    if (lane == 0) count[p] = m;
}

// rows [B, C * k, 9] = (box7 with the yaw of the direction classifier, score, label), class-major; total [B]
__global__ void __launch_bounds__(256)
pp_collect(const float* __restrict__ box, const float* __restrict__ score, const int32_t* __restrict__ dirbit,
           const int32_t* __restrict__ keep, const int32_t* __restrict__ count, int64_t B, int64_t k, int C, float dir_offset,
           float* __restrict__ rows, int32_t* __restrict__ total) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * C * k) return;
    const int64_t p = t / k, j = t - p * k;
    const int64_t b = p / C;
    const int c = (int)(p - b * C);
    int off = 0, tot = 0;
    for (int q = 0; q < C; ++q) {
        const int n_q = count[b * C + q];
        if (q < c) off += n_q;
        tot += n_q;
    }
    if (c == 0 && j == 0) total[b] = tot;
    if (j >= count[p]) return;
    const int64_t ci = keep[p * k + j];
    const float* src = box + 7 * (b * k + ci);
    float* dst = rows + 9 * (b * C * k + off + j);
#pragma unroll
    for (int q = 0; q < 6; ++q) dst[q] = src[q];
    const float PI = 3.14159274101257324f;       // float32(np.pi): torch scales an f32 tensor by the python float in f32
    const float val = __fsub_rn(src[6], dir_offset);
    const float rot = __fsub_rn(val, __fmul_rn(floorf(__fadd_rn(__fdiv_rn(val, PI), 1.0f)), PI));
    dst[6] = __fadd_rn(__fadd_rn(rot, dir_offset), __fmul_rn(PI, (float)dirbit[b * k + ci]));
    dst[7] = score[p * k + ci];
    dst[8] = (float)c;
}

// ---------------------------------------------------------------------------------------------------------------------
// ml3d_topk_rows -- `max_scores.topk(nms_pre)` of Anchor3DHead.get_bboxes_single (ml3d/torch/models/point_pillars.py:985-992)
// for every sample of the batch at once: rows x n scores (n = H * W * A = 321 408 anchors at KITTI), the k <= 4096 largest of
// every row.  An MSB radix SELECT on the order-preserving key: three histogram passes over 11 + 11 + 10 key bits (the prologue
// of a pass re-derives the digits chosen so far from the earlier tables, so no "pick" launch sits between two passes), one
// compaction pass, one single-workgroup bitonic sort of the k selected (key, index) pairs.  The rows are a few MB: the cost is
// the six dependent launches, not the traffic (torch.topk's multi-block radix select + its result sort: ~20 launches).
// Order: descending value, ties by ascending index, NaN above +inf (torch.topk's convention).  Ties AT the k-th value are
// resolved by index as well, so the result is a pure function of the input (torch leaves that order unspecified).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TOPK_BINS = 2048;                 // bins of a level's table (level 2 uses the first 1024)
constexpr int TOPK_CHUNK = 4096;                // elements per workgroup: 256 threads x 16 CONSECUTIVE elements

// ascending key = descending value; NaN first; -0.0 and +0.0 are EQUAL values (one key), like every comparison-based top-k
__device__ __forceinline__ unsigned topk_key(float x) { return x != x ? 0u : ~f2ord(x == 0.f ? 0.f : x); }
__device__ __forceinline__ int topk_shift(int level) { return level == 0 ? 21 : level == 1 ? 10 : 0; }
__device__ __forceinline__ unsigned topk_digit(unsigned key, int level) { return (key >> topk_shift(level)) & (level == 2 ? 1023u : 2047u); }

// the digit of `level` the k-th smallest key falls in, given that level's table of the keys that share the digits chosen
// before: *bin = smallest b with count(bins <= b) >= krem; krem becomes the rank INSIDE that bin.  Whole workgroup (256).
__device__ void topk_pick(const int* __restrict__ table, int& krem, unsigned& bin, int* sh /* [8] */) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int loc[TOPK_BINS / 256], s = 0;
#pragma unroll
    for (int i = 0; i < TOPK_BINS / 256; ++i) { loc[i] = table[t * (TOPK_BINS / 256) + i]; s += loc[i]; }
    const int incl = wave_inclusive_scan(s);
    __syncthreads();                                        // (sh is reused from one level to the next)
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    int excl = incl - s;
    for (int w = 0; w < wv; ++w) excl += sh[w];
    __syncthreads();
    if (excl < krem && krem <= excl + s) {                  // exactly one thread: the table holds >= krem keys
        int run = excl;
#pragma unroll
        for (int i = 0; i < TOPK_BINS / 256; ++i) {
            if (krem <= run + loc[i]) { sh[4] = t * (TOPK_BINS / 256) + i; sh[5] = krem - run; break; }
            run += loc[i];
        }
    }
    __syncthreads();
    bin = (unsigned)sh[4];
    krem = sh[5];
}

// tables [rows][3][TOPK_BINS] (zeroed by the host); pass `level` counts the keys that match the digits of the levels before
__global__ void __launch_bounds__(256)
topk_hist(const float* __restrict__ x, int64_t n, int k, int level, int* __restrict__ tables) {
    __shared__ int hist[TOPK_BINS];
    __shared__ int sh[8];
    const int64_t row = blockIdx.y;
    int* tab = tables + row * 3 * TOPK_BINS;
    unsigned prefix = 0;                                    // the key's bits above this level's digit
    int krem = k;
    for (int l = 0; l < level; ++l) {
        unsigned b;
        topk_pick(tab + l * TOPK_BINS, krem, b, sh);
        prefix |= b << topk_shift(l);
    }
    for (int i = threadIdx.x; i < TOPK_BINS; i += 256) hist[i] = 0;
    __syncthreads();
    const float* xr = x + row * n;
    const int64_t base = (int64_t)blockIdx.x * TOPK_CHUNK;
    const unsigned above = level == 0 ? 0u : (0xffffffffu << topk_shift(level - 1));
#pragma unroll 4
    for (int j = 0; j < TOPK_CHUNK / 256; ++j) {
        const int64_t i = base + j * 256 + threadIdx.x;
        if (i < n) {
            const unsigned key = topk_key(xr[i]);
            if ((key & above) == prefix) atomicAdd(&hist[topk_digit(key, level)], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TOPK_BINS; i += 256)
        if (hist[i]) atomicAdd(&tab[level * TOPK_BINS + i], hist[i]);
}

// sel [rows][k] <- (key << 32 | index) of the k selected elements of a row, in no particular order: every key below the
// threshold key T, and the krem lowest-indexed ones among the keys equal to T.  counters [rows][2] (zeroed): slots handed out.
__global__ void __launch_bounds__(256)
topk_take(const float* __restrict__ x, int64_t n, int k, const int* __restrict__ tables, int* __restrict__ counters,
          u64* __restrict__ sel) {
    __shared__ int sh[8];
    const int64_t row = blockIdx.y;
    const int* tab = tables + row * 3 * TOPK_BINS;
    unsigned T = 0;
    int krem = k, n_equal = 0;
    for (int l = 0; l < 3; ++l) {
        unsigned b;
        topk_pick(tab + l * TOPK_BINS, krem, b, sh);
        T |= b << topk_shift(l);
        if (l == 2) n_equal = tab[2 * TOPK_BINS + b];       // keys equal to T in the whole row
    }
    const int n_less = k - krem;                            // keys below T: all of them are selected
    const float* xr = x + row * n;
    u64* out = sel + row * k;
    int* cnt = counters + 2 * row;
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t base = (int64_t)blockIdx.x * TOPK_CHUNK, first = base + (int64_t)t * 16;
    unsigned key[16];
    int eq_mine = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int64_t i = first + j;
        key[j] = i < n ? topk_key(xr[i]) : 0xffffffffu;
        if (i < n && key[j] < T) out[atomicAdd(&cnt[0], 1)] = ((u64)key[j] << 32) | (u64)(uint32_t)i;
        eq_mine += (i < n && key[j] == T) ? 1 : 0;
    }
    if (n_equal == krem) {                                  // the usual case: every key equal to T is taken, any slot will do
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (first + j < n && key[j] == T) out[n_less + atomicAdd(&cnt[1], 1)] = ((u64)T << 32) | (u64)(uint32_t)(first + j);
        return;
    }
    // ties AT the threshold: the krem lowest indices.  A thread owns 16 consecutive elements, so the rank of an equal
    // element = equals of the row before this workgroup's chunk + equals of the threads before + equals before it in the thread.
    const int incl = wave_inclusive_scan(eq_mine);
    if (lane == 63) sh[wv] = incl;
    __syncthreads();
    int rank = incl - eq_mine;
    for (int w = 0; w < wv; ++w) rank += sh[w];
    const int eq_block = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    if (eq_block == 0) return;                              // (uniform)
    int before = 0;                                         // recount of the chunks before this one: only workgroups that hold a
    for (int64_t i = t; i < base; i += 256) before += topk_key(xr[i]) == T ? 1 : 0;      // tie get here, and ties are rare
    before = wave_sum(before);
    if (lane == 0) sh[wv] = before;
    __syncthreads();
    rank += sh[0] + sh[1] + sh[2] + sh[3];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if (first + j < n && key[j] == T) {
            if (rank < krem) out[n_less + rank] = ((u64)T << 32) | (u64)(uint32_t)(first + j);
            ++rank;
        }
    }
}

// one workgroup per row: bitonic sort of the k selected pairs (padded to a power of two with ~0) in LDS, ascending
// (key, index) = descending value, ties by ascending index
__global__ void __launch_bounds__(256)
topk_sort(const u64* __restrict__ sel, int k, int kp2, int64_t* __restrict__ out_index, float* __restrict__ out_value,
          const float* __restrict__ x, int64_t n) {
    __shared__ u64 keys[NMSB_MAX];
    const int64_t row = blockIdx.x;
    for (int i = threadIdx.x; i < kp2; i += 256) keys[i] = i < k ? sel[row * k + i] : ~0ull;
    __syncthreads();
    for (int size = 2; size <= kp2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int p = threadIdx.x; p < kp2 / 2; p += 256) {
                const int lo = ((p & ~(stride - 1)) << 1) | (p & (stride - 1)), hi = lo | stride;
                const bool up = (lo & size) == 0;
                const u64 a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < k; i += 256) {
        const int64_t idx = (int64_t)(uint32_t)keys[i];
        out_index[row * k + i] = idx;
        if (out_value) out_value[row * k + i] = x[row * n + idx];
    }
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
    zero_async(mask, sizeof(u64) * (size_t)(n * words), st);
    hipLaunchKernelGGL(nms_mask, dim3((unsigned)words, (unsigned)(n < 32768 ? n : 32768), (unsigned)((n + 32767) / 32768)), dim3(64), 0, st,
                       boxes, order, n, iou_threshold, words, mask);
    if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
    hipLaunchKernelGGL(nms_reduce, dim3(1), dim3(64), 0, st, mask, order, n, words, out_keep, out_count);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_pp_anchor_scores(const float* cls, const int64_t* cls_strides, int64_t batch, int num_anchors,
                                     int num_classes, int64_t hw, float* out_scores, void* stream) {
    if (batch < 0 || num_anchors <= 0 || num_classes <= 0 || hw < 0) return ML3D_E_INVALID;
    if (batch * hw == 0) return 0;
    if (!cls || !cls_strides || !out_scores) return ML3D_E_INVALID;
    const ml3d::HeadMap mc = {cls, cls_strides[0], cls_strides[1], cls_strides[2]};
    hipLaunchKernelGGL(ml3d::pp_anchor_scores, dim3((unsigned)((batch * hw + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       mc, batch, num_anchors, num_classes, hw, out_scores);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" size_t ml3d_topk_rows_workspace_bytes(int64_t rows, int64_t n, int64_t k) {
    if (rows < 0 || n < 0 || k < 0 || k > n || k > NMSB_MAX) return 0;
    const size_t R = (size_t)(rows > 0 ? rows : 1);
    return nms_align(4 * R * (3 * TOPK_BINS + 2)) /* tables + counters */ + nms_align(8 * R * (size_t)(k > 0 ? k : 1)) /* sel */ + 512;
}

extern "C" int ml3d_topk_rows(const float* values, int64_t rows, int64_t n, int64_t k, int64_t* out_index, float* out_value,
                              void* workspace, size_t workspace_bytes, void* stream) {
    if (rows < 0 || n < 0 || k < 0 || k > n || n > 0x7fffffffll || rows > 65535) return ML3D_E_INVALID;
    if (k > NMSB_MAX) return ML3D_E_UNSUPPORTED;
    if (rows == 0 || k == 0) return 0;
    if (!values || !out_index || !workspace) return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_topk_rows_workspace_bytes(rows, n, k)) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    int* tables = (int*)p;
    int* counters = tables + (size_t)rows * 3 * TOPK_BINS;
    p += nms_align(4 * (size_t)rows * (3 * TOPK_BINS + 2));
    u64* sel = (u64*)p;
    zero_async(tables, 4 * (size_t)rows * (3 * TOPK_BINS + 2), st);        // (a fill kernel, not hipMemsetAsync: grid.h)
    const dim3 grid((unsigned)((n + TOPK_CHUNK - 1) / TOPK_CHUNK), (unsigned)rows);
    for (int level = 0; level < 3; ++level)
        hipLaunchKernelGGL(ml3d::topk_hist, grid, dim3(256), 0, st, values, n, (int)k, level, tables);
    hipLaunchKernelGGL(ml3d::topk_take, grid, dim3(256), 0, st, values, n, (int)k, tables, counters, sel);
    int kp2 = 1;
    while (kp2 < k) kp2 <<= 1;
    hipLaunchKernelGGL(ml3d::topk_sort, dim3((unsigned)rows), dim3(256), 0, st, sel, (int)k, kp2, out_index, out_value, values, n);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" size_t ml3d_pp_boxes_workspace_bytes(int64_t batch, int64_t k, int num_classes) {
    if (batch < 0 || k < 0 || num_classes <= 0 || k > NMSB_MAX) return 0;
    const size_t P = (size_t)(batch > 0 ? batch : 1) * (size_t)num_classes, n = (size_t)(k > 0 ? k : 1);
    const size_t words = (n + 63) / 64;
    return nms_align(4 * P * n) /* order */ + nms_align(4 * P) /* nvalid */ + nms_align(8 * P * n * words) /* mask */ +
           nms_align(4 * P * n) /* keep */ + nms_align(4 * P) /* count */ + nms_align(4 * 7 * (P / num_classes) * n) /* box */ +
           nms_align(4 * 5 * (P / num_classes) * n) /* bev */ + nms_align(4 * P * n) /* score */ +
           nms_align(4 * (P / num_classes) * n) /* dir */ + 512;
}

extern "C" int ml3d_pp_boxes(const float* cls, const float* reg, const float* dir, const int64_t* strides9,
                             const float* anchors, const int64_t* candidates, int64_t batch, int64_t k, int num_anchors, int num_classes, int64_t hw,
                             float score_threshold, float iou_threshold, float dir_offset, float* out_rows,
                             int32_t* out_total, void* workspace, size_t workspace_bytes, void* stream) {
    if (batch < 0 || k < 0 || num_anchors <= 0 || num_classes <= 0 || hw < 0 || !out_total) return ML3D_E_INVALID;
    if (k > NMSB_MAX) return ML3D_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (batch == 0) return 0;
    if (k == 0) { (void)hipMemsetAsync(out_total, 0, sizeof(int32_t) * (size_t)batch, st); return 0; }
    if (!cls || !reg || !dir || !strides9 || !anchors || !candidates || !out_rows) return ML3D_E_INVALID;
    const ml3d::HeadMap mc = {cls, strides9[0], strides9[1], strides9[2]}, mr = {reg, strides9[3], strides9[4], strides9[5]},
                        md = {dir, strides9[6], strides9[7], strides9[8]};
    if (workspace_bytes < ml3d_pp_boxes_workspace_bytes(batch, k, num_classes)) return ML3D_E_WORKSPACE;
    const int64_t P = batch * num_classes;
    const int words = (int)((k + 63) / 64);
    char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    uint32_t* order = (uint32_t*)p;  p += nms_align(4 * (size_t)(P * k));
    int32_t* nvalid = (int32_t*)p;   p += nms_align(4 * (size_t)P);
    u64* mask = (u64*)p;             p += nms_align(8 * (size_t)(P * k * words));
    int32_t* keep = (int32_t*)p;     p += nms_align(4 * (size_t)(P * k));
    int32_t* count = (int32_t*)p;    p += nms_align(4 * (size_t)P);
    float* box = (float*)p;          p += nms_align(4 * 7 * (size_t)(batch * k));
    float* bev = (float*)p;          p += nms_align(4 * 5 * (size_t)(batch * k));
    float* score = (float*)p;        p += nms_align(4 * (size_t)(P * k));
    int32_t* dirbit = (int32_t*)p;
    hipLaunchKernelGGL(ml3d::pp_decode, dim3((unsigned)((batch * k + 255) / 256)), dim3(256), 0, st, mc, mr, md,
                       anchors, candidates, batch, k, num_anchors, num_classes, hw, box, bev, score, dirbit);
    hipLaunchKernelGGL(ml3d::nmsb_order, dim3((unsigned)P), dim3(256), 0, st, score, k, score_threshold, order, nvalid);
    hipLaunchKernelGGL(ml3d::nmsb_mask, dim3((unsigned)words, (unsigned)k, (unsigned)P), dim3(64), 0, st, bev, order, nvalid,
                       k, num_classes, iou_threshold, words, mask);
    hipLaunchKernelGGL(ml3d::nmsb_reduce, dim3((unsigned)P), dim3(64), 0, st, mask, order, nvalid, k, words, keep, count);
    hipLaunchKernelGGL(ml3d::pp_collect, dim3((unsigned)((P * k + 255) / 256)), dim3(256), 0, st, box, score, dirbit, keep, count,
                       batch, k, num_classes, dir_offset, out_rows, out_total);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_iou_bev(const float* boxes_a, const float* boxes_b, int64_t n, int64_t m, float* out_iou, void* stream) {
    if (n < 0 || m < 0) return ML3D_E_INVALID;
    if (n == 0 || m == 0) return 0;
    if (!boxes_a || !boxes_b || !out_iou) return ML3D_E_INVALID;
    hipLaunchKernelGGL(ml3d::iou_pairs, dim3((unsigned)((n * m + 63) / 64)), dim3(64), 0, (hipStream_t)stream, boxes_a, boxes_b,
                       n, m, 0, out_iou);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}

extern "C" int ml3d_iou_3d(const float* boxes_a, const float* boxes_b, int64_t n, int64_t m, float* out_iou, void* stream) {
    if (n < 0 || m < 0) return ML3D_E_INVALID;
    if (n == 0 || m == 0) return 0;
    if (!boxes_a || !boxes_b || !out_iou) return ML3D_E_INVALID;
    hipLaunchKernelGGL(ml3d::iou_pairs, dim3((unsigned)((n * m + 63) / 64)), dim3(64), 0, (hipStream_t)stream, boxes_a, boxes_b,
                       n, m, 1, out_iou);
    return hipGetLastError() == hipSuccess ? 0 : ML3D_E_LAUNCH;
}
