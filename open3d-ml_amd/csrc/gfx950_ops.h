// gfx950_ops.h — single gfx950 instructions the compiler will not pick on its own.
// (The host emulator of the test-suite shadows this header with portable equivalents: tests/hipemu/include.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// register budget of a kernel: exactly n waves per SIMD (512 / n VGPRs per lane)
#define ML3D_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))

namespace ml3d {

// Compare-exchange of two packed (d2, index) keys held as doubles (see knn.hip): v_min_f64 + v_max_f64.
// fmin()/fmax() would first canonicalise each operand (v_max_f64 x, x, x: signalling-NaN quieting the IEEE mode asks
// for) -- three instructions per slot instead of two.  The keys are never NaN, so the raw instructions are exact.
__device__ __forceinline__ void key_minmax(double a, double b, double& lo, double& hi) {
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
}
// One slot of the sorted insertion of a key x into a register-resident ascending list: slot <- min(slot, x), x <- max(slot, x),
// the slot updated IN PLACE.  (key_minmax's two fresh outputs per compare-exchange made the compiler rotate the whole list
// back into its loop-carried registers after every insertion: 16 v_mov_b64 + 15 s_nop on top of the 31 useful instructions --
// round-5 listing of knn_query_multi<16, true>.)  The keys are never NaN, so the raw instructions are exact.
__device__ __forceinline__ void key_insert_step(double& slot, double& x) {
    double t;
    asm("v_max_f64 %0, %1, %2" : "=v"(t) : "v"(slot), "v"(x));
    asm("v_min_f64 %0, %0, %1" : "+v"(slot) : "v"(x));
    x = t;
}
// acc <- min(acc, key) for the lanes where `take` holds, as ONE predicated v_min_f64 (exec masking: the volatile asm keeps the
// compiler from turning the condition into a 64-bit select -- two v_cndmask_b32 per candidate in a VALU-issue-bound loop)
__device__ __forceinline__ void key_min_if(bool take, double& acc, double key) {
    if (take) asm volatile("v_min_f64 %0, %0, %1" : "+v"(acc) : "v"(key));
}
__device__ __forceinline__ double key_min(double a, double b) {
    double lo;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    return lo;
}

// ---- bf16 matrix path (gemm.hip: f32-equivalent products as six bf16 MFMAs over a three-way split of both operands) ---------
// two floats -> two bf16 (round to nearest even), packed low | high << 16: one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t bf16_pack2(float a, float b) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// x of the lanes whose bit is clear in `lanes` (= __ballot(ok), a scalar pair) becomes 0, IN PLACE: a conditional select the compiler
// cannot turn into a copy of the register on the path that skips it
__device__ __forceinline__ void keep_if(uint32_t& x, bool /* ok: the host emulator's form */, unsigned long long lanes) {
    asm("v_cndmask_b32 %0, 0, %0, %1" : "+v"(x) : "s"(lanes));
}
// a - b as ONE v_sub_f32 (opaque to the SLP vectoriser, which would pair two of them into a v_pk_add_f32)
__device__ __forceinline__ float sub_f32(float a, float b) {
    float d;
    asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// v_mfma_f32_32x32x16_bf16: lane l supplies A[l % 32][8 (l / 32) + e] and B[8 (l / 32) + e][l % 32], e = 0 .. 7 (eight bf16 = four dwords
// each); the 32 x 32 float result has the layout of every 32 x 32 MFMA (row (r & 3) + 8 (r >> 2) + 4 (l / 32), column l % 32)
typedef float ml3d_f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t ml3d_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ml3d_f32x16 mfma_bf16_32x32x16(ml3d_u32x4 a, ml3d_u32x4 b, ml3d_f32x16 c) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// v_mfma_f32_16x16x32_bf16: lane l supplies A[l % 16][8 (l / 16) + e] and B[8 (l / 16) + e][l % 16], e = 0 .. 7; the 16 x 16 float result has
// the layout of every 16 x 16 MFMA (row 4 (l / 16) + r, column l % 16).  Same matrix rate as the 32 x 32 x 16 form (16 cycles per SIMD).
typedef float ml3d_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ml3d_f32x4 mfma_bf16_16x16x32(ml3d_u32x4 a, ml3d_u32x4 b, ml3d_f32x4 c) {
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// v_permlane32_swap_b32: lanes 32-63 of x trade places with lanes 0-31 of y (one instruction; no LDS, no index register)
__device__ __forceinline__ void lane32_swap(uint32_t& x, uint32_t& y) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// "does any ACTIVE lane of this wave see pred" in DIVERGENT control flow (lanes whose loops have different trip counts): a ballot
// over the current exec mask.  The host emulator runs every lane as its own fiber and cannot rendezvous lanes that sit at
// different iterations, so its stand-in answers with the lane's own predicate -- callers must stay correct when the answer is
// `true` more often than that (k-NN pending queue: an early flush is always exact).
__device__ __forceinline__ bool wave_any_active(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0ull; }

// One slot per ACTIVE lane of an append-only list, in DIVERGENT control flow: one atomicAdd per wave (the lowest active lane adds the
// number of active lanes, the others take base + their rank among them), slots of a wave contiguous and in lane order.  (The host
// emulator cannot rendezvous lanes in divergent flow: there every lane adds 1 -- slot values differ, set of slots identical.)
__device__ __forceinline__ unsigned wave_append(unsigned* counter) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
    const unsigned lane = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));   // rank among the active lanes
    unsigned base = 0u;
    if (lane == 0u) base = atomicAdd(counter, (unsigned)__builtin_popcountll(m));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
    return base + lane;
}

// ---- inter-workgroup hand-off words of the fused scans (sort.hip / voxel.hip) -----------------------------------------------------
// A hand-off word carries its value AND a ready tag in ONE dword, so a relaxed device-scope load / store of that dword is all the
// protocol needs: no fence (no L2 write-back: every XCD has its own L2, a release fence at device scope costs a buffer_wbl2), no
// second location whose order against the first would matter.  Device scope makes the access bypass the non-coherent levels.
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void spin_pause() { __builtin_amdgcn_s_sleep(2); }

}  // namespace ml3d
