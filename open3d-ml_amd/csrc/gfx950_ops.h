// gfx950_ops.h — single gfx950 instructions the compiler will not pick on its own.
// (The host emulator of the test-suite shadows this header with portable equivalents: tests/hipemu/include.)
#pragma once
#include <hip/hip_runtime.h>

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

// "does any ACTIVE lane of this wave see pred" in DIVERGENT control flow (lanes whose loops have different trip counts): a ballot
// over the current exec mask.  The host emulator runs every lane as its own fiber and cannot rendezvous lanes that sit at
// different iterations, so its stand-in answers with the lane's own predicate -- callers must stay correct when the answer is
// `true` more often than that (k-NN pending queue: an early flush is always exact).
__device__ __forceinline__ bool wave_any_active(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0ull; }

}  // namespace ml3d
