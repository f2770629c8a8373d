// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate (dependent chain vs 2 independent chains) at 1..4 waves/SIMD,
// and the shader clock under that load.   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(256) spin(float* out, int iters, long long* clk) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
    long long t0 = clock64();
    long long w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = clock64();
    long long w1 = wall_clock64();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int CHAINS>
static void run(int blocks_per_cu, int threads) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    float* out; long long* clk;
    hipMalloc(&out, sizeof(float) * cus * 16 * 1024);
    hipMalloc(&clk, 16);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = cus * blocks_per_cu;
    hipLaunchKernelGGL(spin<CHAINS>, dim3(grid), dim3(threads), 0, 0, out, 100, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<CHAINS>, dim3(grid), dim3(threads), 0, 0, out, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)grid * (threads / 64) * iters * 8.0 * CHAINS * 4096.0;
    printf("chains=%d waves/SIMD=%.1f  %.3f ms  %.1f TFLOP/s  shader clk/wall(100MHz) = %.0f MHz\n", CHAINS,
           blocks_per_cu * (threads / 64) / 4.0, ms, flops / ms * 1e-9, (double)h[0] / (double)h[1] * 100.0);
    hipFree(out); hipFree(clk);
}

int main() {
    run<1>(1, 256); run<1>(2, 256); run<1>(4, 256);
    run<2>(1, 256); run<2>(2, 256);
    run<4>(1, 256);
    return 0;
}
