// sort.hip — stable LSD radix sort (see sort.h).
#include "sort.h"

#include "grid.h"

namespace ml3d {

static inline size_t sort_align(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int64_t sort_blocks(int64_t n) { return (n + 1023) / 1024; }

size_t sort_ws_bytes(int64_t n) {
    int64_t nb = sort_blocks(n > 0 ? n : 1);
    int64_t hist = 1 + 256 * nb;
    size_t b = 0;
    b += sort_align(sizeof(u64) * (size_t)(n > 0 ? n : 1));
    b += sort_align(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    b += sort_align(sizeof(int) * (size_t)hist);
    b += sort_align(sizeof(int) * (size_t)((hist + 1023) / 1024 + 1));
    return b + 256;
}

bool sort_ws_carve(void* ws, size_t bytes, int64_t n, SortWs* out) {
    if (bytes < sort_ws_bytes(n)) return false;
    int64_t nb = sort_blocks(n > 0 ? n : 1);
    int64_t hist = 1 + 256 * nb;
    char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    out->keys_alt = (u64*)p;       p += sort_align(sizeof(u64) * (size_t)(n > 0 ? n : 1));
    out->vals_alt = (uint32_t*)p;  p += sort_align(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    out->hist = (int*)p;           p += sort_align(sizeof(int) * (size_t)hist);
    out->block_sums = (int*)p;
    out->n = n;
    return true;
}

// hist[1 + d * nb + blk] = number of keys of block blk whose digit is d  (hist[0] = 0)
__global__ void __launch_bounds__(256)
rs_hist(const u64* __restrict__ keys, int64_t n, int shift, int* hist, int nb) {
    __shared__ int h[256];
    const int t = threadIdx.x;
    h[t] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * 1024;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int64_t i = base + r * 256 + t;
        if (i < n) atomicAdd(&h[(int)((keys[i] >> shift) & 255ull)], 1);
    }
    __syncthreads();
    hist[1 + (int64_t)t * nb + blockIdx.x] = h[t];
    if (blockIdx.x == 0 && t == 0) hist[0] = 0;
}

// After the inclusive scan of hist[1..], hist[d * nb + blk] is the first output slot of (d, blk).
__global__ void __launch_bounds__(256)
rs_scatter(const u64* __restrict__ kin, const uint32_t* __restrict__ vin, u64* __restrict__ kout,
           uint32_t* __restrict__ vout, int64_t n, int shift, const int* __restrict__ hist, int nb) {
    __shared__ int base[256];
    __shared__ int cnt[4][256];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    base[t] = hist[(int64_t)t * nb + blockIdx.x];
    const int64_t first = (int64_t)blockIdx.x * 1024;
    for (int r = 0; r < 4; ++r) {
        const int64_t i = first + r * 256 + t;
        const bool valid = i < n;
        const u64 key = valid ? kin[i] : 0ull;
        const uint32_t val = valid ? vin[i] : 0u;
        const int d = (int)((key >> shift) & 255ull);
        cnt[0][t] = 0; cnt[1][t] = 0; cnt[2][t] = 0; cnt[3][t] = 0;
        // lanes of this wave holding the same digit (wave-level multisplit)
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int bit = (d >> b) & 1;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const int rank = __popcll(peers & ((1ull << lane) - 1ull));
        __syncthreads();
        if (valid && rank == 0) cnt[w][d] = __popcll(peers);
        __syncthreads();
        if (valid) {
            int off = base[d] + rank;
            for (int w2 = 0; w2 < w; ++w2) off += cnt[w2][d];
            kout[off] = key;
            vout[off] = val;
        }
        __syncthreads();
        base[t] += cnt[0][t] + cnt[1][t] + cnt[2][t] + cnt[3][t];
        __syncthreads();
    }
}

int sort_pairs_u64(u64* keys, uint32_t* vals, int64_t n, int key_bits, const SortWs& ws, hipStream_t stream) {
    if (n <= 1) return 0;
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 64) key_bits = 64;
    int passes = (key_bits + 7) / 8;
    if (passes & 1) ++passes;   // even: the result lands back in (keys, vals)
    if (passes > 8) passes = 8;
    const int nb = (int)sort_blocks(n);
    const int64_t hist_n = 256 * (int64_t)nb;
    u64* kin = keys; uint32_t* vin = vals;
    u64* kout = ws.keys_alt; uint32_t* vout = ws.vals_alt;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        hipLaunchKernelGGL(rs_hist, dim3(nb), dim3(256), 0, stream, kin, n, shift, ws.hist, nb);
        if (hipGetLastError() != hipSuccess) return -3;
        if (scan_inclusive_i32(ws.hist + 1, hist_n, ws.block_sums, stream)) return -3;
        hipLaunchKernelGGL(rs_scatter, dim3(nb), dim3(256), 0, stream, kin, vin, kout, vout, n, shift, ws.hist, nb);
        if (hipGetLastError() != hipSuccess) return -3;
        u64* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    return 0;
}

}  // namespace ml3d
