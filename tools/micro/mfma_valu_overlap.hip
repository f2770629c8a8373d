// micro-benchmark: do VALU instructions overlap v_mfma_f32_32x32x2_f32 on a gfx950 SIMD?
// Each wave loops { 8 dependent MFMAs ; NV independent v_fma_f32 }.  MFMA-only time per iteration = 8 x 64 cycles;
// VALU-only = NV x 4 cycles (wave64 on a 16-lane SIMD).  Reported: cycles per iteration at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NM>
__global__ void __launch_bounds__(256) spin(float* out, int iters, long long* clk) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < NM; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NV; ++u) v[u & 7] = __builtin_fmaf(v[u & 7], b, a);
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int NV, int NM>
static void run(int blocks_per_cu) {
    int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out; long long* clk;
    (void)hipMalloc(&out, sizeof(float) * cus * 16 * 1024);
    (void)hipMalloc(&clk, 16);
    const int iters = 4000;
    hipLaunchKernelGGL((spin<NV, NM>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, 50, clk);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((spin<NV, NM>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, iters, clk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h = 0; (void)hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    // SIMD cycles per (iteration of every resident wave) assuming 2.4 GHz
    printf("MFMA/iter=%d VALU/iter=%3d waves/SIMD=%d : %7.1f cycles/iter/wave (block 0 clock)  kernel %.3f ms = %7.1f SIMD cycles per iter-round @2.4GHz  (mfma %d x waves, valu %d x waves)\n", NM, NV,
           blocks_per_cu, (double)h / iters, ms, ms * 1e-3 * 2.4e9 / iters, NM * 64, NV * 4);
    (void)hipFree(out); (void)hipFree(clk);
}

int main() {
    run<0, 8>(1); run<32, 8>(1); run<64, 8>(1); run<128, 8>(1); run<128, 0>(1);
    run<0, 8>(2); run<64, 8>(2); run<128, 8>(2);
    run<64, 8>(3); run<128, 8>(3);
    return 0;
}
