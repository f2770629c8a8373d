// micro-benchmark for DESIGN.md §9.10: does a small workgroup that keeps per-lane, dynamically indexed vertex lists in lane-interleaved
// LDS (the rotated-IoU clipping of rounds 4-5: ds_write_b64 / ds_read_b64 / ds_read2st64_b64 at sh[i * 64 + lane]) return different
// results when workgroups with a large LDS footprint share its CUs?  Kernel A evaluates the SAME rotated-box intersection twice per
// lane -- vertex lists in LDS (the old code, verbatim) and in registers (the shipped code) -- and counts the lanes whose two areas differ
// bit for bit.  Co-runners on a second stream: B<LDS bytes, MFMA, LDS traffic> in four flavours.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o lds_corun lds_corun.hip && ./lds_corun
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

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
constexpr int POLY_MAX = 16;
constexpr int POLY_LDS = 2 * POLY_MAX * 64;

// ---- rounds 4-5: the lists in lane-interleaved LDS (verbatim) --------------------------------------------------------------------
__device__ float area_lds(const P2* A, const P2* B, P2* sh) {
    const int lane = threadIdx.x & 63;
    P2* cur = sh + lane;
    P2* nxt = sh + POLY_MAX * 64 + lane;
    int nc = 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i * 64] = A[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (nc > 0) {
            const P2 p0 = B[e], p1 = B[(e + 1) & 3];
            const P2 ed = {__fsub_rn(p1.x, p0.x), __fsub_rn(p1.y, p0.y)};
            int nn = 0;
            for (int i = 0; i < nc; ++i) {
                if (nn + 2 > POLY_MAX) return 0.f;
                const P2 s = cur[i * 64], t = cur[(i + 1 == nc ? 0 : i + 1) * 64];
                const P2 vs = {__fsub_rn(s.x, p0.x), __fsub_rn(s.y, p0.y)}, vt = {__fsub_rn(t.x, p0.x), __fsub_rn(t.y, p0.y)};
                const float ds = cross2(ed, vs), dt = cross2(ed, vt);
                if (ds >= 0.f) nxt[64 * nn++] = s;
                if ((ds >= 0.f) != (dt >= 0.f)) {
                    const float u = __fdiv_rn(ds, __fsub_rn(ds, dt));
                    P2 ip;
                    ip.x = __fadd_rn(s.x, __fmul_rn(u, __fsub_rn(t.x, s.x)));
                    ip.y = __fadd_rn(s.y, __fmul_rn(u, __fsub_rn(t.y, s.y)));
                    nxt[64 * nn++] = ip;
                }
            }
            nc = nn;
            P2* sw = cur; cur = nxt; nxt = sw;
        }
    }
    if (nc < 3) return 0.f;
    float a = 0.f;
    for (int i = 0; i < nc; ++i) a = __fadd_rn(a, cross2(cur[i * 64], cur[(i + 1 == nc ? 0 : i + 1) * 64]));
    return __fmul_rn(0.5f, fabsf(a));
}

// ---- the same b64 lists, every access a SEPARATE volatile 64-bit load / store (no ds_read2 / ds_write2 / 2st64 merging) -------------
__device__ __forceinline__ P2 ldv(const P2* p) {
    const unsigned long long u = *reinterpret_cast<const volatile unsigned long long*>(p);
    P2 r; r.x = __uint_as_float((unsigned)u); r.y = __uint_as_float((unsigned)(u >> 32));
    return r;
}
__device__ __forceinline__ void stv(P2* p, P2 v) {
    *reinterpret_cast<volatile unsigned long long*>(p) = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
}
__device__ float area_lds64v(const P2* A, const P2* B, P2* sh) {
    const int lane = threadIdx.x & 63;
    P2* cur = sh + lane;
    P2* nxt = sh + POLY_MAX * 64 + lane;
    int nc = 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) stv(cur + i * 64, A[i]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (nc > 0) {
            const P2 p0 = B[e], p1 = B[(e + 1) & 3];
            const P2 ed = {__fsub_rn(p1.x, p0.x), __fsub_rn(p1.y, p0.y)};
            int nn = 0;
            for (int i = 0; i < nc; ++i) {
                if (nn + 2 > POLY_MAX) return 0.f;
                const P2 s = ldv(cur + i * 64), t = ldv(cur + (i + 1 == nc ? 0 : i + 1) * 64);
                const P2 vs = {__fsub_rn(s.x, p0.x), __fsub_rn(s.y, p0.y)}, vt = {__fsub_rn(t.x, p0.x), __fsub_rn(t.y, p0.y)};
                const float ds = cross2(ed, vs), dt = cross2(ed, vt);
                if (ds >= 0.f) stv(nxt + 64 * nn++, s);
                if ((ds >= 0.f) != (dt >= 0.f)) {
                    const float u = __fdiv_rn(ds, __fsub_rn(ds, dt));
                    P2 ip;
                    ip.x = __fadd_rn(s.x, __fmul_rn(u, __fsub_rn(t.x, s.x)));
                    ip.y = __fadd_rn(s.y, __fmul_rn(u, __fsub_rn(t.y, s.y)));
                    stv(nxt + 64 * nn++, ip);
                }
            }
            nc = nn;
            P2* sw = cur; cur = nxt; nxt = sw;
        }
    }
    if (nc < 3) return 0.f;
    float a = 0.f;
    for (int i = 0; i < nc; ++i) a = __fadd_rn(a, cross2(ldv(cur + i * 64), ldv(cur + (i + 1 == nc ? 0 : i + 1) * 64)));
    return __fmul_rn(0.5f, fabsf(a));
}

// ---- the same lists with x and y in separate float planes: every LDS access is a ds_read_b32 / ds_write_b32 ------------------------
__device__ float area_lds32(const P2* A, const P2* B, float* sh) {
    const int lane = threadIdx.x & 63;
    float* cur = sh + lane;                                          // x at [i * 64], y at [POLY_MAX * 64 + i * 64]
    float* nxt = sh + 2 * POLY_MAX * 64 + lane;
    constexpr int Y = POLY_MAX * 64;
    int nc = 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) { cur[i * 64] = A[i].x; cur[Y + i * 64] = A[i].y; }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (nc > 0) {
            const P2 p0 = B[e], p1 = B[(e + 1) & 3];
            const P2 ed = {__fsub_rn(p1.x, p0.x), __fsub_rn(p1.y, p0.y)};
            int nn = 0;
            for (int i = 0; i < nc; ++i) {
                if (nn + 2 > POLY_MAX) return 0.f;
                const int i1 = (i + 1 == nc ? 0 : i + 1);
                const P2 s = {cur[i * 64], cur[Y + i * 64]}, t = {cur[i1 * 64], cur[Y + i1 * 64]};
                const P2 vs = {__fsub_rn(s.x, p0.x), __fsub_rn(s.y, p0.y)}, vt = {__fsub_rn(t.x, p0.x), __fsub_rn(t.y, p0.y)};
                const float ds = cross2(ed, vs), dt = cross2(ed, vt);
                if (ds >= 0.f) { nxt[64 * nn] = s.x; nxt[Y + 64 * nn] = s.y; ++nn; }
                if ((ds >= 0.f) != (dt >= 0.f)) {
                    const float u = __fdiv_rn(ds, __fsub_rn(ds, dt));
                    nxt[64 * nn] = __fadd_rn(s.x, __fmul_rn(u, __fsub_rn(t.x, s.x)));
                    nxt[Y + 64 * nn] = __fadd_rn(s.y, __fmul_rn(u, __fsub_rn(t.y, s.y)));
                    ++nn;
                }
            }
            nc = nn;
            float* sw = cur; cur = nxt; nxt = sw;
        }
    }
    if (nc < 3) return 0.f;
    float a = 0.f;
    for (int i = 0; i < nc; ++i) {
        const int i1 = (i + 1 == nc ? 0 : i + 1);
        const P2 s = {cur[i * 64], cur[Y + i * 64]}, t = {cur[i1 * 64], cur[Y + i1 * 64]};
        a = __fadd_rn(a, cross2(s, t));
    }
    return __fmul_rn(0.5f, fabsf(a));
}

// ---- shipped: the lists in registers (nms.hip) -----------------------------------------------------------------------------------
struct PolyList { float x[POLY_MAX], y[POLY_MAX]; };
template <int SLOTS>
__device__ __forceinline__ void poly_put(PolyList& L, int at, P2 v) {
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) { const bool hit = q == at; L.x[q] = hit ? v.x : L.x[q]; L.y[q] = hit ? v.y : L.y[q]; }
}
template <int I>
__device__ __forceinline__ bool clip_vertex(const PolyList& cur, int nc, PolyList& nxt, int& nn, P2 p0, P2 ed) {
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
__device__ __forceinline__ bool clip_from(const PolyList& cur, int nc, PolyList& nxt, int& nn, P2 p0, P2 ed) {
    if constexpr (I < POLY_MAX) {
        if (I < nc) {
            if (!clip_vertex<I>(cur, nc, nxt, nn, p0, ed)) return false;
            return clip_from<I + 1>(cur, nc, nxt, nn, p0, ed);
        }
    }
    return true;
}
__device__ __forceinline__ float area_reg(const P2* A, const P2* B) {
    PolyList cur, nxt;
#pragma unroll
    for (int i = 0; i < POLY_MAX; ++i) { cur.x[i] = i < 4 ? A[i].x : 0.f; cur.y[i] = i < 4 ? A[i].y : 0.f; nxt.x[i] = 0.f; nxt.y[i] = 0.f; }
    int nc = 4;
#pragma unroll 1
    for (int e = 0; e < 4; ++e) {
        if (nc > 0) {
            const P2 p0 = e == 0 ? B[0] : e == 1 ? B[1] : e == 2 ? B[2] : B[3];
            const P2 p1 = e == 0 ? B[1] : e == 1 ? B[2] : e == 2 ? B[3] : B[0];
            const P2 ed = {__fsub_rn(p1.x, p0.x), __fsub_rn(p1.y, p0.y)};
            int nn = 0;
            if (!clip_from<0>(cur, nc, nxt, nn, p0, ed)) return 0.f;
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

// kernel A: the launch shape of nmsb_mask -- one wave per (box a, 64-box word); lane j clips box a against box 64 cb + j
template <int B32>          // 0: b64 lists (the old code), 1: b32 planes, 2: b64 lists through separate volatile accesses
__global__ void __launch_bounds__(64)
clip_pairs(const float* __restrict__ boxes, int n, unsigned long long* __restrict__ mismatch_lanes, unsigned int* __restrict__ counts) {
    __shared__ P2 poly[POLY_LDS];
    const int a = blockIdx.y, cb = blockIdx.x, lane = threadIdx.x;
    const int j = cb * 64 + lane;
    bool bad = false;
    if (j < n && j != a) {
        float ba[5], bj[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) { ba[q] = boxes[5 * a + q]; bj[q] = boxes[5 * j + q]; }
        P2 ca[4], cbx[4];
        box_corners(ba, ca);
        box_corners(bj, cbx);
        const float l = B32 == 1 ? area_lds32(ca, cbx, reinterpret_cast<float*>(poly)) : B32 == 2 ? area_lds64v(ca, cbx, poly) : area_lds(ca, cbx, poly);
        const float r = area_reg(ca, cbx);
        bad = __float_as_uint(l) != __float_as_uint(r);
    }
    const unsigned long long m = __ballot(bad);
    if (lane == 0) {
        atomicAdd(&counts[0], 1u);                                   // waves
        if (m) { atomicAdd(&counts[1], 1u); atomicOr(mismatch_lanes, m); atomicAdd(&counts[2], (unsigned)__popcll(m)); }
    }
}

// kernel A': no geometry -- every lane writes eight 64-bit patterns to its lane-interleaved slots and reads them back (volatile: real
// ds_write_b64 / ds_read_b64), with the full wave (DIVERGENT = false) or under a pseudo-random per-lane execution mask
template <bool DIVERGENT>
__global__ void __launch_bounds__(64)
rw64(int iters, unsigned long long* __restrict__ mismatch_lanes, unsigned int* __restrict__ counts) {
    __shared__ unsigned long long sh[32 * 64];
    volatile unsigned long long* v = sh;
    const unsigned lane = threadIdx.x;
    bool bad = false;
    for (int it = 0; it < iters; ++it) {
        const bool active = !DIVERGENT || (((lane * 2654435761u + (unsigned)it * 40503u + blockIdx.x) >> 9) & 1u);
        if (active) {
            const int base = (it & 3) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[(base + i) * 64 + lane] = ((unsigned long long)(lane * 131u + i * 7u + it) << 32) | (0x9e3779b9u ^ (lane + 64u * i) ^ (unsigned)it);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                bad |= v[(base + i) * 64 + lane] != (((unsigned long long)(lane * 131u + i * 7u + it) << 32) | (0x9e3779b9u ^ (lane + 64u * i) ^ (unsigned)it));
        }
    }
    const unsigned long long m = __ballot(bad);
    if (lane == 0) {
        atomicAdd(&counts[0], 1u);
        if (m) { atomicAdd(&counts[1], 1u); atomicOr(mismatch_lanes, m); atomicAdd(&counts[2], (unsigned)__popcll(m)); }
    }
}

// kernel A'': the TWO-ADDRESS 64-bit forms the compiler picked for the vertex lists (cur[i * 64] and cur[(i + 1) * 64] in one instruction):
// ds_write2st64_b64 / ds_read2st64_b64 by inline asm, write then read back, full wave or under a per-lane execution mask.  (The array is
// the kernel's only LDS object: its LDS byte address is its index * 8.)
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
template <bool DIVERGENT>
__global__ void __launch_bounds__(64)
rw2st64(int iters, unsigned long long* __restrict__ mismatch_lanes, unsigned int* __restrict__ counts) {
    __shared__ unsigned long long sh[32 * 64];
    const unsigned lane = threadIdx.x;
    if (lane == 0 && iters < 0) sh[0] = 1;                            // (keeps the allocation)
    bool bad = false;
    for (int it = 0; it < iters; ++it) {
        const bool active = !DIVERGENT || (((lane * 2654435761u + (unsigned)it * 40503u + blockIdx.x) >> 9) & 1u);
        if (active) {
            const unsigned base = (unsigned)(it & 3) * 8u;
#pragma unroll
            for (unsigned i = 0; i < 8; i += 2) {
                const unsigned long long a = ((unsigned long long)(lane * 131u + i * 7u + it) << 32) | (0x9e3779b9u ^ (lane + 64u * i) ^ (unsigned)it);
                const unsigned long long b = ((unsigned long long)(lane * 137u + i * 5u + it) << 32) | (0x7f4a7c15u ^ (lane + 64u * i) ^ (unsigned)it);
                const unsigned addr = ((base + i) * 64u + lane) * 8u;
                asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:0 offset1:1" :: "v"(addr), "v"(a), "v"(b) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (unsigned i = 0; i < 8; i += 2) {
                const unsigned long long a = ((unsigned long long)(lane * 131u + i * 7u + it) << 32) | (0x9e3779b9u ^ (lane + 64u * i) ^ (unsigned)it);
                const unsigned long long b = ((unsigned long long)(lane * 137u + i * 5u + it) << 32) | (0x7f4a7c15u ^ (lane + 64u * i) ^ (unsigned)it);
                const unsigned addr = ((base + i) * 64u + lane) * 8u;
                u64x2 r;
                asm volatile("ds_read2st64_b64 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
                bad |= r[0] != a || r[1] != b;
            }
        }
    }
    const unsigned long long m = __ballot(bad);
    if (lane == 0) {
        atomicAdd(&counts[0], 1u);
        if (m) { atomicAdd(&counts[1], 1u); atomicOr(mismatch_lanes, m); atomicAdd(&counts[2], (unsigned)__popcll(m)); }
    }
}

// kernel A3: ds_write2_b32 / ds_read2_b32 (two ADJACENT dwords: how the compiler moves the 4-byte-aligned 8-byte vertices), inline asm
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <bool DIVERGENT>
__global__ void __launch_bounds__(64)
rw2b32(int iters, unsigned long long* __restrict__ mismatch_lanes, unsigned int* __restrict__ counts) {
    __shared__ unsigned long long sh[32 * 64];
    const unsigned lane = threadIdx.x;
    if (lane == 0 && iters < 0) sh[0] = 1;
    bool bad = false;
    for (int it = 0; it < iters; ++it) {
        const bool active = !DIVERGENT || (((lane * 2654435761u + (unsigned)it * 40503u + blockIdx.x) >> 9) & 1u);
        if (active) {
            const unsigned base = (unsigned)(it & 3) * 8u;
#pragma unroll
            for (unsigned i = 0; i < 8; ++i) {
                const unsigned a = 0x9e3779b9u ^ (lane * 131u + i * 7u + it), b = 0x7f4a7c15u ^ (lane + 64u * i) ^ (unsigned)it;
                const unsigned addr = ((base + i) * 64u + lane) * 8u;
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:1" :: "v"(addr), "v"(a), "v"(b) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (unsigned i = 0; i < 8; ++i) {
                const unsigned a = 0x9e3779b9u ^ (lane * 131u + i * 7u + it), b = 0x7f4a7c15u ^ (lane + 64u * i) ^ (unsigned)it;
                const unsigned addr = ((base + i) * 64u + lane) * 8u;
                u32x2 r;
                asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
                bad |= r[0] != a || r[1] != b;
            }
        }
    }
    const unsigned long long m = __ballot(bad);
    if (lane == 0) {
        atomicAdd(&counts[0], 1u);
        if (m) { atomicAdd(&counts[1], 1u); atomicOr(mismatch_lanes, m); atomicAdd(&counts[2], (unsigned)__popcll(m)); }
    }
}

// kernel A4 (written after the round's GPU budget was spent: NOT RUN): are the destination registers of INACTIVE lanes preserved by a
// 64-bit LDS read?  Every lane keeps a live 64-bit value in a register; under a per-lane mask the active lanes overwrite it by
// ds_read_b64 with a read-write destination; afterwards every lane checks: active -> the LDS pattern, inactive -> its own old value.
template <int WIDTH>        // 64: ds_read_b64, 32: ds_read_b32 (control)
__global__ void __launch_bounds__(64)
inactive_lanes(int iters, unsigned long long* __restrict__ mismatch_lanes, unsigned int* __restrict__ counts) {
    __shared__ unsigned long long sh[32 * 64];
    const unsigned lane = threadIdx.x;
    for (int i = 0; i < 32; ++i) sh[i * 64 + lane] = ((unsigned long long)(lane * 131u + i) << 32) | (0x9e3779b9u ^ (lane + 64u * i));
    __syncthreads();
    bool bad = false;
    for (int it = 0; it < iters; ++it) {
        const bool active = (((lane * 2654435761u + (unsigned)it * 40503u + blockIdx.x) >> 9) & 1u) != 0u;
        const unsigned i = (unsigned)it & 31u;
        const unsigned long long keep = 0xabcdef0123456789ull ^ ((unsigned long long)lane << 20) ^ (unsigned long long)it;
        unsigned long long v = keep;
        const unsigned addr = (i * 64u + lane) * 8u;
        if (active) {
            if (WIDTH == 64) {
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(v) : "v"(addr) : "memory");
            } else {
                unsigned lo = (unsigned)v;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(lo) : "v"(addr) : "memory");
                v = (v & 0xffffffff00000000ull) | lo;
            }
        }
        const unsigned long long pat = ((unsigned long long)(lane * 131u + i) << 32) | (0x9e3779b9u ^ (lane + 64u * i));
        const unsigned long long want = active ? (WIDTH == 64 ? pat : ((keep & 0xffffffff00000000ull) | (unsigned)pat)) : keep;
        bad |= v != want;
    }
    const unsigned long long m = __ballot(bad);
    if (lane == 0) {
        atomicAdd(&counts[0], 1u);
        if (m) { atomicAdd(&counts[1], 1u); atomicOr(mismatch_lanes, m); atomicAdd(&counts[2], (unsigned)__popcll(m)); }
    }
}

// co-runner B: 256 threads, LDS_BYTES of static LDS, optional bf16 MFMA, optional LDS traffic (b128 reads / writes)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// MFMA: 0 none, 1 v_mfma_f32_32x32x16_bf16, 2 v_mfma_f32_32x32x2_f32;  TRAFFIC: 0 none, 1 two b128 reads + one b128 write + barrier,
// 2 the two b128 reads only (no write, no barrier)
template <int LDS_BYTES, int MFMA, int TRAFFIC>
__global__ void __launch_bounds__(256)
corun(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    u32x4* L = reinterpret_cast<u32x4*>(lds);
    constexpr int N16 = LDS_BYTES / 16;
    const int t = threadIdx.x;
    for (int i = t; i < N16; i += 256) L[i] = u32x4{0x3f803f80u + (unsigned)i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u32x4 a = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    for (int it = 0; it < iters; ++it) {
        if (TRAFFIC) {
            const int i0 = (t * 7 + it * 13) % N16, i1 = (t * 3 + it * 29) % N16;
            a = L[i0];
            b = L[i1];
            if (TRAFFIC == 1) L[(t + it * 256) % N16] = u32x4{a[0] ^ b[1], a[1], b[2], a[3]};
        }
        if (MFMA == 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        } else if (MFMA == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a[u] & 0x3fffffffu), __uint_as_float(b[u] & 0x3fffffffu), acc, 0, 0, 0);
        } else {
            acc[0] += __uint_as_float(a[0] & 0x3fffffffu);
        }
        if (TRAFFIC == 1) __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + t] = s;
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int LDS_BYTES, int MFMA, int TRAFFIC>
static void launch_corun(hipStream_t st, float* out, int iters) {
    hipLaunchKernelGGL((corun<LDS_BYTES, MFMA, TRAFFIC>), dim3(512), dim3(256), 0, st, out, iters);
}

int main() {
    const int n = 100;                                               // boxes per problem (nms_pre of pointpillars_kitti.yml)
    std::vector<float> hb(5 * n);
    srand(7);
    for (int i = 0; i < n; ++i) {                                     // overlapping car-sized boxes in a 20 m patch, any yaw
        const float cx = 20.f * rand() / RAND_MAX, cy = 20.f * rand() / RAND_MAX, w = 1.4f + 0.6f * rand() / RAND_MAX, l = 3.4f + 1.2f * rand() / RAND_MAX;
        hb[5 * i] = cx - w / 2; hb[5 * i + 1] = cy - l / 2; hb[5 * i + 2] = cx + w / 2; hb[5 * i + 3] = cy + l / 2; hb[5 * i + 4] = 6.28f * rand() / RAND_MAX;
    }
    float *boxes, *out;
    unsigned long long* lanes;
    unsigned int* counts;
    CHECK(hipMalloc(&boxes, sizeof(float) * 5 * n));
    CHECK(hipMalloc(&out, sizeof(float) * 512 * 256));
    CHECK(hipMalloc(&lanes, 8));
    CHECK(hipMalloc(&counts, 16));
    CHECK(hipMemcpy(boxes, hb.data(), sizeof(float) * 5 * n, hipMemcpyHostToDevice));
    hipStream_t sa, sb;
    CHECK(hipStreamCreate(&sa));
    CHECK(hipStreamCreate(&sb));
    const char* names[27] = {"quiet GPU", "62.5 KB LDS + bf16 MFMA + b128 reads / write / barrier", "62.5 KB LDS + the LDS traffic, no MFMA",
                            "62.5 KB LDS allocated + bf16 MFMA, no LDS traffic", "8 KB LDS + bf16 MFMA + b128 reads / write / barrier",
                            "8 KB LDS + f32 MFMA (32x32x2) + b128 reads / write / barrier", "8 KB LDS + bf16 MFMA + b128 READS only",
                            "8 KB LDS + bf16 MFMA + traffic; kernel A with b32 LDS accesses", "quiet GPU; kernel A with b32 LDS accesses",
                            "quiet GPU; A' = plain b64 write / read-back, full wave", "8 KB LDS + bf16 MFMA + traffic; A' full wave",
                            "quiet GPU; A' under a per-lane execution mask", "8 KB LDS + bf16 MFMA + traffic; A' under a per-lane execution mask",
                             "quiet GPU; A'' = ds_write2st64_b64 / ds_read2st64_b64, full wave", "8 KB LDS + bf16 MFMA + traffic; A'' full wave",
                             "quiet GPU; A'' under a per-lane execution mask", "8 KB LDS + bf16 MFMA + traffic; A'' under a per-lane execution mask",
                             "quiet GPU; kernel A, b64 lists through unmerged volatile accesses", "8 KB LDS + bf16 MFMA + traffic; kernel A, unmerged volatile b64",
                             "quiet GPU; A3 = ds_write2_b32 / ds_read2_b32, full wave", "8 KB LDS + bf16 MFMA + traffic; A3 full wave",
                             "quiet GPU; A3 under a per-lane execution mask", "8 KB LDS + bf16 MFMA + traffic; A3 under a per-lane execution mask",
                             "quiet GPU; A4 = inactive lanes keep their registers across ds_read_b64", "8 KB LDS + bf16 MFMA + traffic; A4 (ds_read_b64)",
                             "quiet GPU; A4 with ds_read_b32 (control)", "8 KB LDS + bf16 MFMA + traffic; A4 with ds_read_b32 (control)"};
    for (int mode = 0; mode < 27; ++mode) {
        CHECK(hipMemset(lanes, 0, 8));
        CHECK(hipMemset(counts, 0, 16));
        CHECK(hipDeviceSynchronize());
        for (int rep = 0; rep < 40; ++rep) {
            const int iters = 3000;
            if (mode == 1) launch_corun<64000, 1, 1>(sb, out, iters);
            if (mode == 2) launch_corun<64000, 0, 1>(sb, out, iters);
            if (mode == 3) launch_corun<64000, 1, 0>(sb, out, iters);
            if (mode == 4 || mode == 7 || mode == 10 || mode == 12 || mode == 14 || mode == 16 || mode == 18 || mode == 20 || mode == 22 || mode == 24 || mode == 26) launch_corun<8192, 1, 1>(sb, out, iters);
            if (mode == 5) launch_corun<8192, 2, 1>(sb, out, iters);
            if (mode == 6) launch_corun<8192, 1, 2>(sb, out, iters);
            for (int k = 0; k < 24; ++k) {                               // 24 problems of 100 boxes per decode, like 8 sweeps x 3 classes
                if (mode == 23 || mode == 24) hipLaunchKernelGGL((inactive_lanes<64>), dim3(2 * n), dim3(64), 0, sa, 256, lanes, counts);
                else if (mode >= 25) hipLaunchKernelGGL((inactive_lanes<32>), dim3(2 * n), dim3(64), 0, sa, 256, lanes, counts);
                else if (mode == 19 || mode == 20) hipLaunchKernelGGL((rw2b32<false>), dim3(2 * n), dim3(64), 0, sa, 64, lanes, counts);
                else if (mode == 21 || mode == 22) hipLaunchKernelGGL((rw2b32<true>), dim3(2 * n), dim3(64), 0, sa, 64, lanes, counts);
                else if (mode >= 17) hipLaunchKernelGGL((clip_pairs<2>), dim3(2, n), dim3(64), 0, sa, boxes, n, lanes, counts);
                else if (mode == 13 || mode == 14) hipLaunchKernelGGL((rw2st64<false>), dim3(2 * n), dim3(64), 0, sa, 64, lanes, counts);
                else if (mode >= 15 && mode <= 16) hipLaunchKernelGGL((rw2st64<true>), dim3(2 * n), dim3(64), 0, sa, 64, lanes, counts);
                else if (mode == 9 || mode == 10) hipLaunchKernelGGL((rw64<false>), dim3(2 * n), dim3(64), 0, sa, 64, lanes, counts);
                else if (mode >= 11) hipLaunchKernelGGL((rw64<true>), dim3(2 * n), dim3(64), 0, sa, 64, lanes, counts);
                else if (mode >= 7) hipLaunchKernelGGL((clip_pairs<1>), dim3(2, n), dim3(64), 0, sa, boxes, n, lanes, counts);
                else hipLaunchKernelGGL((clip_pairs<0>), dim3(2, n), dim3(64), 0, sa, boxes, n, lanes, counts);
            }
        }
        CHECK(hipDeviceSynchronize());
        unsigned long long hl;
        unsigned int hc[4];
        CHECK(hipMemcpy(&hl, lanes, 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hc, counts, 16, hipMemcpyDeviceToHost));
        printf("co-runner %-66s: %u waves, %u with a wrong lane (%u lanes), lane mask %016llx\n",
               names[mode], hc[0], hc[1], hc[2], hl);
        fflush(stdout);
    }
    return 0;
}
