// hipemu — a host-side stand-in for <hip/hip_runtime.h>, TEST INFRASTRUCTURE ONLY.
//
// The authoring container has no GPU.  To debug the indexing / protocol logic
// of the HIP kernels under open3d-ml_amd/csrc before spending GPU minutes, the
// tests compile the SAME .hip sources as host C++ (clang++ -x c++) with this
// directory first on the include path.  Each workgroup is run as a set of
// cooperative fibers (ucontext) on one OS thread; __syncthreads() and the
// wave-64 collectives (__shfl*, __ballot, MFMA f32) are rendezvous points.
// A collective reached by only part of a live wave is reported as a deadlock —
// the product kernels keep collectives wave-uniform.
//
// This is never shipped: the product library is built only by hipcc for gfx950
// and there is no CPU fallback in open3d-ml_amd/.
#pragma once
#include <ucontext.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local
#define __constant__ static
#ifndef __restrict__
#define __restrict__ __restrict
#endif
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_smem());

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct hipemu_uint3 { unsigned x, y, z; };

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
typedef void* hipEvent_t;
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return 0; }
// the emulated device: 4 CUs, 2 resident workgroups each (keeps persistent kernels multi-tile per wave in tests)
static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return 0; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return 0; }

namespace hipemu {

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 256 * 1024;

enum State { RUN = 0, AT_BARRIER = 1, AT_WAVE = 2, DONE = 3 };

struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    int state = RUN;
};

struct WaveScratch {
    // two generations of per-lane 64-byte slots (enough for a + b + flag)
    alignas(16) unsigned char slot[2][WAVE][80];
    unsigned char valid[2][WAVE];
    unsigned pub[2][WAVE];   // ordinal of the collective the slot was published for
    unsigned gen[WAVE];
};

struct Block {
    std::vector<Fiber> fibers;
    std::vector<WaveScratch> waves;
    ucontext_t sched;
    int cur = -1;
    int nthreads = 0;
    dim3 bdim, gdim, bidx;
    std::vector<unsigned char> dyn;
    const std::function<void()>* body = nullptr;
};

extern thread_local Block* g_blk;
extern thread_local hipemu_uint3 g_tid, g_bid;
extern thread_local dim3 g_bdim, g_gdim;

inline void* dyn_smem() { return g_blk->dyn.data(); }

inline void yield_to_sched() {
    Block* b = g_blk;
    swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}

inline int flat_tid() { return (int)(g_tid.x + g_bdim.x * (g_tid.y + g_bdim.y * g_tid.z)); }

inline void block_barrier() {
    Block* b = g_blk;
    b->fibers[b->cur].state = AT_BARRIER;
    yield_to_sched();
}

// Wave rendezvous: publish `n` bytes, wait for all live lanes, then peers' slots are readable
// until this lane's next-but-one collective.
struct WaveView {
    WaveScratch* w;
    int g;     // generation parity used
    int lane;
    unsigned ord;   // ordinal of this collective (per lane count of collectives so far)
    const void* peer(int l) const { return w->slot[g][l]; }
    // a peer's slot counts only if it was published for THIS collective: a lane that has exited (or, after a
    // divergent exit, simply does not take part any more) leaves its last slot readable for the lanes that
    // are still returning from that same rendezvous, and is invisible to every later collective
    bool peer_valid(int l) const { return w->valid[g][l] != 0 && w->pub[g][l] == ord; }
};

inline WaveView wave_exchange(const void* data, size_t n) {
    Block* b = g_blk;
    int t = b->cur;
    int wv = t / WAVE, lane = t % WAVE;
    WaveScratch& W = b->waves[wv];
    int g = (int)(W.gen[lane] & 1u);
    const unsigned ord = ++W.gen[lane];
    if (n > 80) { fprintf(stderr, "hipemu: exchange too large\n"); abort(); }
    memcpy(W.slot[g][lane], data, n);
    W.valid[g][lane] = 1;
    W.pub[g][lane] = ord;
    b->fibers[t].state = AT_WAVE;
    yield_to_sched();
    return WaveView{&W, g, lane, ord};
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void trace(const char* name);

}  // namespace hipemu

#define threadIdx (hipemu::g_tid)
#define blockIdx (hipemu::g_bid)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    (hipemu::trace(#kernel), hipemu::launch(dim3(grid), dim3(block), (size_t)(shmem), [=]() { kernel(__VA_ARGS__); }))

static inline void __syncthreads() { hipemu::block_barrier(); }
static inline void __builtin_amdgcn_s_barrier() { hipemu::block_barrier(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}

// ---- wave collectives -------------------------------------------------------
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    auto vw = hipemu::wave_exchange(&v, sizeof(T));
    int base = (vw.lane / width) * width;
    int l = base + ((src % width) + width) % width;
    T r = v;
    if (vw.peer_valid(l)) memcpy(&r, vw.peer(l), sizeof(T));
    return r;
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    auto vw = hipemu::wave_exchange(&v, sizeof(T));
    int l = vw.lane ^ mask;
    T r = v;
    if ((l / width) == (vw.lane / width) && l < 64 && vw.peer_valid(l)) memcpy(&r, vw.peer(l), sizeof(T));
    return r;
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    auto vw = hipemu::wave_exchange(&v, sizeof(T));
    int l = vw.lane + (int)delta;
    T r = v;
    if ((l / width) == (vw.lane / width) && l < 64 && vw.peer_valid(l)) memcpy(&r, vw.peer(l), sizeof(T));
    return r;
}
template <typename T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
    auto vw = hipemu::wave_exchange(&v, sizeof(T));
    int l = vw.lane - (int)delta;
    T r = v;
    if (l >= 0 && (l / width) == (vw.lane / width) && vw.peer_valid(l)) memcpy(&r, vw.peer(l), sizeof(T));
    return r;
}
static inline unsigned long long __ballot(int pred) {
    int p = pred ? 1 : 0;
    auto vw = hipemu::wave_exchange(&p, sizeof(int));
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (vw.peer_valid(l)) { int q; memcpy(&q, vw.peer(l), sizeof(int)); if (q) m |= (1ull << l); }
    return m;
}
#define ML3D_HIPEMU 1   // lets a kernel swap an inline-asm instruction for its portable equivalent
static inline void __builtin_amdgcn_fence(int, const char*) {}
static inline void __builtin_amdgcn_fence(int, const char*, const char*) {}
static inline void __builtin_amdgcn_iglp_opt(int) {}
// (clang provides __builtin_nontemporal_load/store on the host as well)
static inline void __builtin_amdgcn_sched_group_barrier(int, int, int) {}
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline void __builtin_amdgcn_wave_barrier() { int z = 0; (void)hipemu::wave_exchange(&z, sizeof(z)); }
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
static inline int __all(int pred) {
    int p = pred ? 1 : 0;
    auto vw = hipemu::wave_exchange(&p, sizeof(int));
    for (int l = 0; l < 64; ++l)
        if (vw.peer_valid(l)) { int q; memcpy(&q, vw.peer(l), sizeof(int)); if (!q) return 0; }
    return 1;
}
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
static inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
static inline unsigned __lane_id() { return (unsigned)(hipemu::flat_tid() % 64); }

template <typename T>
static inline T hipemu_readfirstlane(T v) {
    auto vw = hipemu::wave_exchange(&v, sizeof(T));
    for (int l = 0; l < 64; ++l)
        if (vw.peer_valid(l)) { T r; memcpy(&r, vw.peer(l), sizeof(T)); return r; }
    return v;
}
#define __builtin_amdgcn_readfirstlane(x) hipemu_readfirstlane(x)
static inline int __builtin_amdgcn_readlane(int v, int lane) {
    auto vw = hipemu::wave_exchange(&v, sizeof(int));
    int r = v;
    if (vw.peer_valid(lane)) memcpy(&r, vw.peer(lane), sizeof(int));
    return r;
}

// ---- f32-input MFMA (layouts per /opt/skills/guides/cdna_hip_programming.md §3) ---------------
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));

static inline hipemu_f32x16 hipemu_mfma_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    float ab[2] = {a, b};
    auto vw = hipemu::wave_exchange(ab, sizeof(ab));
    int lane = vw.lane;
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int col = lane & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float pa[2], pb[2];
            memcpy(pa, vw.peer(row + 32 * k), sizeof(pa));  // A[i=row][k] lives in lane row+32k
            memcpy(pb, vw.peer(col + 32 * k), sizeof(pb));  // B[k][j=col] lives in lane col+32k
            acc = fmaf(pa[0], pb[1], acc);
        }
        d[r] = acc;
    }
    return d;
}
static inline hipemu_f32x4 hipemu_mfma_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    float ab[2] = {a, b};
    auto vw = hipemu::wave_exchange(ab, sizeof(ab));
    int lane = vw.lane;
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (lane >> 4) * 4 + r;
        int col = lane & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float pa[2], pb[2];
            memcpy(pa, vw.peer(row + 16 * k), sizeof(pa));
            memcpy(pb, vw.peer(col + 16 * k), sizeof(pb));
            acc = fmaf(pa[0], pb[1], acc);
        }
        d[r] = acc;
    }
    return d;
}
// v_mfma_f32_32x32x16_bf16: lane l supplies A[i = l % 32][k = 8 (l / 32) + e] and B[k = 8 (l / 32) + e][j = l % 32], e = 0..7, as eight
// bf16 packed in four dwords.  Every product of two bf16 is exact in float; the 16 products and C are summed here in double and
// rounded to float ONCE (the hardware's internal adder is not documented bit for bit -- measured on an MI355X the kernel's error
// against a float64 product equals the f32 MFMA kernel's, profiles/r05_bf16x3_conv.log -- so tests of kernels built on this
// instruction carry a tolerance, not an equality)
typedef uint32_t hipemu_u32x4 __attribute__((ext_vector_type(4)));
static inline hipemu_f32x16 hipemu_mfma_32x32x16_bf16(hipemu_u32x4 a, hipemu_u32x4 b, hipemu_f32x16 c) {
    uint32_t ab[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    auto vw = hipemu::wave_exchange(ab, sizeof(ab));
    const int lane = vw.lane;
    auto bf = [](const uint32_t* w, int e) { uint32_t u = (e & 1) ? (w[e >> 1] & 0xffff0000u) : (w[e >> 1] << 16); float f; memcpy(&f, &u, 4); return f; };
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        double acc = (double)c[r];
        for (int half = 0; half < 2; ++half) {
            uint32_t pa[8], pb[8];
            memcpy(pa, vw.peer(row + 32 * half), sizeof(pa));
            memcpy(pb, vw.peer(col + 32 * half), sizeof(pb));
            for (int e = 0; e < 8; ++e) acc += (double)bf(pa, e) * (double)bf(pb + 4, e);
        }
        d[r] = (float)acc;
    }
    return d;
}
// v_mfma_f32_16x16x32_bf16: lane l supplies A[i = l % 16][k = 8 (l / 16) + e] and B[k = 8 (l / 16) + e][j = l % 16], e = 0..7; the result
// D[i = 4 (l / 16) + r][j = l % 16], r = 0..3 (summed in double, rounded once: see the 32 x 32 x 16 form above)
static inline hipemu_f32x4 hipemu_mfma_16x16x32_bf16(hipemu_u32x4 a, hipemu_u32x4 b, hipemu_f32x4 c) {
    uint32_t ab[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    auto vw = hipemu::wave_exchange(ab, sizeof(ab));
    const int lane = vw.lane;
    auto bf = [](const uint32_t* w, int e) { uint32_t u = (e & 1) ? (w[e >> 1] & 0xffff0000u) : (w[e >> 1] << 16); float f; memcpy(&f, &u, 4); return f; };
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r, col = lane & 15;
        double acc = (double)c[r];
        for (int q = 0; q < 4; ++q) {
            uint32_t pa[8], pb[8];
            memcpy(pa, vw.peer(row + 16 * q), sizeof(pa));
            memcpy(pb, vw.peer(col + 16 * q), sizeof(pb));
            for (int e = 0; e < 8; ++e) acc += (double)bf(pa, e) * (double)bf(pb + 4, e);
        }
        d[r] = (float)acc;
    }
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu_mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu_mfma_16x16x4f32

// ---- atomics (blocks of one launch may run on several OS threads) -----------------------------
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline float atomicAdd(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    float f;
    do { memcpy(&f, &old, 4); f += v; memcpy(&nw, &f, 4); }
    while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&f, &old, 4);
    return f;
}
static inline double atomicAdd(double* p, double v) {
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    unsigned long long old = __atomic_load_n(q, __ATOMIC_RELAXED), want;
    double cur;
    do { memcpy(&cur, &old, 8); cur += v; memcpy(&want, &cur, 8); } while (!__atomic_compare_exchange_n(q, &old, want, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    memcpy(&cur, &old, 8);
    return cur;
}
static inline int atomicMin(int* p, int v) { return __atomic_fetch_min(p, v, __ATOMIC_RELAXED); }
static inline int atomicMax(int* p, int v) { return __atomic_fetch_max(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMin(unsigned* p, unsigned v) { return __atomic_fetch_min(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicMax(unsigned* p, unsigned v) { return __atomic_fetch_max(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) { return __atomic_fetch_min(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { return __atomic_fetch_max(p, v, __ATOMIC_RELAXED); }
static inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline int atomicCAS(int* p, int cmp, int v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
static inline unsigned atomicCAS(unsigned* p, unsigned cmp, unsigned v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
static inline unsigned long long atomicCAS(unsigned long long* p, unsigned long long cmp, unsigned long long v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
static inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

// ---- misc device math used by the kernels ------------------------------------------------------
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline double __longlong_as_double(long long i) { double d; memcpy(&d, &i, 8); return d; }
static inline long long __double_as_longlong(double d) { long long i; memcpy(&i, &d, 8); return i; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
#define __expf(a) expf(a)
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {
    return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c));
}
static inline float __frcp_rn(float a) { return 1.0f / a; }
template <typename T> static inline T __ldg(const T* p) { return *p; }
using std::max;
using std::min;
