// micro-benchmark: issue rate of the VALU instructions the k-NN selection is built from, per SIMD, at 1 and 2 waves/SIMD:
//   v_min_f64 / v_max_f64 (64-bit key compare-exchange), v_min_u32, v_pk_mul_f32, v_cmp_lt_u64 + v_cndmask (the integer
//   alternative), ds_read_b128 broadcast.   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x

template <int MODE>
__global__ void __launch_bounds__(256) spin(double* out, int iters, long long* clk) {
    __shared__ float4 lds[256];
    lds[threadIdx.x] = make_float4(threadIdx.x, 1.f, 2.f, 3.f);
    __syncthreads();
    double a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = blockIdx.x * 1e-6 + 2.0 * i; }
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = threadIdx.x * 7 + i;
    float4 acc = make_float4(0, 0, 0, 0);
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {        // 16 x (v_min_f64 + v_max_f64) on 8 independent pairs, twice
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    double lo, hi;
                    asm volatile("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a[i]), "v"(b[i]));
                    asm volatile("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a[i]), "v"(b[i]));
                    a[i] = lo; b[i] = hi;
                }
        } else if (MODE == 1) { // 32 x v_min_u32
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_min_u32 %0, %1, %2" : "=v"(u[i]) : "v"(u[i]), "v"(u[(i + 1) & 7]));
        } else if (MODE == 2) { // 32 x v_pk_mul_f32
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(b[i]));
        } else if (MODE == 3) { // 8 x (v_cmp_lt_u64 + 4 v_cndmask): the integer compare-exchange of a 64-bit key, 4x
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    unsigned long long x = __double_as_longlong(a[i]), y = __double_as_longlong(b[i]);
                    bool lt = x < y;
                    a[i] = __longlong_as_double(lt ? x : y);
                    b[i] = __longlong_as_double(lt ? y : x);
                    asm volatile("" : "+v"(a[i]), "+v"(b[i]));
                }
        } else if (MODE == 4) { // 32 x ds_read_b128, wave-uniform address (broadcast)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                float4 c = lds[(it + i) & 255];
                acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
            }
        } else if (MODE == 5) { // 32 x ds_read_b128, per-lane random addresses
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                float4 c = lds[(threadIdx.x * 37 + it * 11 + i * 53) & 255];
                acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
            }
        }
    }
    long long t1 = clock64();
    double s = acc.x + acc.y + acc.z + acc.w;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i] + u[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>
static void run(const char* what, int per_iter, int blocks_per_cu) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    double* out; long long* clk;
    hipMalloc(&out, sizeof(double) * cus * 16 * 256);
    hipMalloc(&clk, 16);
    const int iters = 20000;
    int grid = cus * blocks_per_cu;
    hipLaunchKernelGGL(spin<MODE>, dim3(grid), dim3(256), 0, 0, out, 100, clk);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(spin<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, clk);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
    // each SIMD hosts blocks_per_cu waves; cycles per instruction per SIMD = cycles / (instructions issued on that SIMD)
    printf("%-44s waves/SIMD=%d  %.2f shader cycles per wave-instruction (per SIMD)\n", what, blocks_per_cu,
           (double)h / ((double)iters * per_iter * blocks_per_cu));
    hipFree(out); hipFree(clk);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run<0>("v_min_f64 + v_max_f64 (independent pairs)", 32, w);
        run<1>("v_min_u32", 32, w);
        run<2>("v_pk_mul_f32", 32, w);
        run<3>("u64 compare-exchange (cmp + 4 cndmask) x32", 32, w);
        run<4>("ds_read_b128 broadcast (+4 v_add)", 32, w);
        run<5>("ds_read_b128 random lanes (+4 v_add)", 32, w);
    }
    return 0;
}
