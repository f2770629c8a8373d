// micro-benchmark 2: VALU work INTERLEAVED with MFMAs in program order: loop { 8 x [ v_mfma_f32_32x32x2_f32 ; KV x v_fma_f32 ] }
// CHAINS accumulators used round-robin (1 = every MFMA depends on the previous one).  Full-chip launch, kernel time ->
// SIMD cycles per [MFMA + KV VALU] group.  Co-execution would keep it at 64 cycles while KV x ~2.3 <= 64.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KV, int CHAINS>
__global__ void __launch_bounds__(256) spin(float* out, int iters) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u % CHAINS], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < KV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KV, int CHAINS>
static void run(int blocks_per_cu) {
    int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    (void)hipMalloc(&out, sizeof(float) * cus * 16 * 1024);
    const int iters = 4000;
    hipLaunchKernelGGL((spin<KV, CHAINS>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, 50);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((spin<KV, CHAINS>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("chains=%d VALU per MFMA=%2d waves/SIMD=%d : kernel %.3f ms = %6.1f SIMD cycles per [MFMA+VALU] group per wave @2.4GHz\n", CHAINS,
           KV, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / iters / 8 / blocks_per_cu);
    (void)hipFree(out);
}

int main() {
    run<0, 1>(1); run<4, 1>(1); run<8, 1>(1); run<16, 1>(1); run<24, 1>(1); run<32, 1>(1);
    run<8, 2>(1); run<16, 2>(1); run<24, 2>(1);
    run<8, 1>(2); run<16, 1>(2); run<8, 2>(2); run<16, 2>(2);
    return 0;
}
