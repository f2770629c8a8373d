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

// hist[1 + d * nb + blk] = number of keys of block blk whose digit is d  (hist[0] = 0);
// LOCAL (short sorts, see rs_scatter): hist[blk * 256 + d], no scan follows
template <bool LOCAL>
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
    if constexpr (LOCAL) {
        hist[blockIdx.x * 256 + t] = h[t];
    } else {
        hist[1 + (int64_t)t * nb + blockIdx.x] = h[t];
        if (blockIdx.x == 0 && t == 0) hist[0] = 0;
    }
}

// After the inclusive scan of hist[1..], hist[d * nb + blk] is the first output slot of (d, blk).
// LOCAL (nb <= RS_LOCAL_BLOCKS): every block derives its 256 first slots from the RAW block histograms itself -- thread d sums
// digit d over all blocks (and over the blocks before its own), the 256 totals are scanned in LDS -- so a pass is two launches
// instead of three.  A sort of the 77 000 points of a Semantic3D sub-cloud (the sampler's patch query, once per patch) is
// launch-latency bound: 8 passes x (4.7 + 14.8 + 6.7 us) with the scan, the scan the largest part.
constexpr int RS_LOCAL_BLOCKS = 128;

template <bool LOCAL>
__global__ void __launch_bounds__(256)
rs_scatter(const u64* __restrict__ kin, const uint32_t* __restrict__ vin, u64* __restrict__ kout,
           uint32_t* __restrict__ vout, int64_t n, int shift, const int* __restrict__ hist, int nb) {
    __shared__ int base[256];
    __shared__ __attribute__((aligned(16))) int cnt[4][256];
    __shared__ __attribute__((aligned(16))) int cnt2[LOCAL ? 4 : 1][256];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if constexpr (LOCAL) {
        // wave w sums blocks w, w + 4, ... for ALL digits (lane = four digits, one 16-byte load per block: the loads of a wave
        // are independent, <= 32 of them), the four partial rows meet in LDS
        int4 tot = make_int4(0, 0, 0, 0), before = tot;
#pragma unroll 8
        for (int b = w; b < nb; b += 4) {
            const int4 v = reinterpret_cast<const int4*>(hist + b * 256)[lane];
            tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
            if (b < (int)blockIdx.x) { before.x += v.x; before.y += v.y; before.z += v.z; before.w += v.w; }
        }
        reinterpret_cast<int4*>(cnt[w])[lane] = tot;
        reinterpret_cast<int4*>(cnt2[w])[lane] = before;
        __syncthreads();
        const int total = cnt[0][t] + cnt[1][t] + cnt[2][t] + cnt[3][t];
        const int bef = cnt2[0][t] + cnt2[1][t] + cnt2[2][t] + cnt2[3][t];
        const int incl = wave_inclusive_scan(total);
        __syncthreads();
        if (lane == 63) cnt[0][w] = incl;
        __syncthreads();
        int carry = 0;
        for (int w2 = 0; w2 < w; ++w2) carry += cnt[0][w2];
        __syncthreads();
        base[t] = carry + incl - total + bef;
    } else {
        base[t] = hist[(int64_t)t * nb + blockIdx.x];
    }
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

bool sort_result_in_alt(int64_t n, int key_bits) {
    if (n <= 1) return false;
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 64) key_bits = 64;
    return (((key_bits + 7) / 8) & 1) != 0;
}

int sort_pairs_u64(u64* keys, uint32_t* vals, int64_t n, int key_bits, const SortWs& ws, hipStream_t stream, bool allow_odd) {
    if (n <= 1) return 0;
    if (key_bits < 1) key_bits = 1;
    if (key_bits > 64) key_bits = 64;
    int passes = (key_bits + 7) / 8;
    if ((passes & 1) && !allow_odd) ++passes;   // even: the result lands back in (keys, vals)
    if (passes > 8) passes = 8;
    const int nb = (int)sort_blocks(n);
    const int64_t hist_n = 256 * (int64_t)nb;
    u64* kin = keys; uint32_t* vin = vals;
    u64* kout = ws.keys_alt; uint32_t* vout = ws.vals_alt;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        if (nb <= RS_LOCAL_BLOCKS) {
            hipLaunchKernelGGL(rs_hist<true>, dim3(nb), dim3(256), 0, stream, kin, n, shift, ws.hist, nb);
            hipLaunchKernelGGL(rs_scatter<true>, dim3(nb), dim3(256), 0, stream, kin, vin, kout, vout, n, shift, ws.hist, nb);
        } else {
            hipLaunchKernelGGL(rs_hist<false>, dim3(nb), dim3(256), 0, stream, kin, n, shift, ws.hist, nb);
            if (hipGetLastError() != hipSuccess) return -3;
            if (scan_inclusive_i32(ws.hist + 1, hist_n, ws.block_sums, stream)) return -3;
            hipLaunchKernelGGL(rs_scatter<false>, dim3(nb), dim3(256), 0, stream, kin, vin, kout, vout, n, shift, ws.hist, nb);
        }
        if (hipGetLastError() != hipSuccess) return -3;
        u64* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    return 0;
}

// ---- fused passes (sort.h) ----------------------------------------------------------------------------------------------------------
static inline size_t fs_al(size_t x) { return (x + 255) & ~(size_t)255; }

size_t fused_ws_bytes(int64_t n, int64_t batch) {
    const size_t tiles = (size_t)fused_tiles(n > 0 ? n : 1);
    size_t b = 0;
    b += fs_al(4 * 4 * 256);                       // ghist
    b += fs_al(4 * 8);                             // ticket
    b += fs_al(4 * 6 * FS_MAX_GROUPS);             // gcnt
    b += fs_al(4 * tiles * 256);                   // agg
    b += fs_al(4 * FS_MAX_GROUPS * 256);           // gtot
    b += fs_al(4 * tiles * 4);                     // tA
    b += fs_al(4 * FS_MAX_GROUPS * 4);             // gA
    b += fs_al(4 * tiles * 4);                     // tC
    b += fs_al(4 * FS_MAX_GROUPS * 4);             // gC
    b += fs_al(4 * (size_t)(batch + 2));           // fvf
    return b + 256;
}

bool fused_ws_carve(void* ws, size_t bytes, int64_t n, int64_t batch, FusedWs* o) {
    if (bytes < fused_ws_bytes(n, batch)) return false;
    const size_t tiles = (size_t)fused_tiles(n > 0 ? n : 1);
    char* p = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
    o->base = p;
    o->ghist = (int*)p;          p += fs_al(4 * 4 * 256);
    o->ticket = (uint32_t*)p;    p += fs_al(4 * 8);
    o->gcnt = (uint32_t*)p;      p += fs_al(4 * 6 * FS_MAX_GROUPS);
    o->agg = (uint32_t*)p;       p += fs_al(4 * tiles * 256);
    o->gtot = (uint32_t*)p;      p += fs_al(4 * FS_MAX_GROUPS * 256);
    o->tA = (uint32_t*)p;        p += fs_al(4 * tiles * 4);
    o->gA = (uint32_t*)p;        p += fs_al(4 * FS_MAX_GROUPS * 4);
    o->tC = (uint32_t*)p;        p += fs_al(4 * tiles * 4);
    o->gC = (uint32_t*)p;        p += fs_al(4 * FS_MAX_GROUPS * 4);
    o->fvf = (uint32_t*)p;       p += fs_al(4 * (size_t)(batch + 2));
    o->bytes = (size_t)(p - o->base);
    o->tiles = (int)tiles;
    return true;
}

// One pass = one launch.  Tile = 2048 keys in input order; wave w owns the contiguous quarter [512 w, 512 w + 512) and walks it in
// eight 64-key rounds (coalesced), so (wave, round, lane) IS the input order and the ranks below keep the sort stable:
//   rank of a key among its tile's keys of the same digit = counts of the waves before its own + what its wave counted in earlier
//   rounds + its place among the lanes of this round holding the digit (wave-level multisplit by ballot, as rs_scatter).
// Thread t then speaks for digit t: tile count -> hand-off table -> first slot of (digit, tile) = digits below (global histogram,
// scanned here) + the same digit in the tiles in front (flat two-level prefix, sort.h).
template <bool FIRST>
__global__ void __launch_bounds__(256)
fs_pass(const uint32_t* __restrict__ kin, const uint32_t* __restrict__ vin, uint32_t* __restrict__ kout,
        uint32_t* __restrict__ vout, int64_t n, int shift, uint32_t tag, const int* __restrict__ ghist, uint32_t* ticket,
        uint32_t* gcnt, uint32_t* agg, uint32_t* gtot) {
    __shared__ int cnt[4][256];
    __shared__ int base[256];
    __shared__ int wsum[4];
    __shared__ int s_tile, s_last;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int tiles = (int)gridDim.x;
    if (t == 0) s_tile = (int)atomicAdd(ticket, 1u);
    cnt[0][t] = 0; cnt[1][t] = 0; cnt[2][t] = 0; cnt[3][t] = 0;
    __syncthreads();
    const int tile = s_tile;
    const int64_t first = (int64_t)tile * FS_TILE + (int64_t)w * 512;
    uint32_t key[8];
    int rk[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int64_t i = first + r * 64 + lane;
        key[r] = i < n ? kin[i] : 0xffffffffu;
    }
    const u64 below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const bool valid = first + r * 64 + lane < n;
        const int d = (int)((key[r] >> shift) & 255u);
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int bit = (d >> b) & 1;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const int prior = cnt[w][d];                         // (a wave's LDS accesses execute in order: every lane has read ...
        const int rank = __popcll(peers & below);
        wave_lds_sync();
        if (valid && rank == 0) cnt[w][d] = prior + __popcll(peers);   // ... before the digit's first lane writes)
        wave_lds_sync();
        rk[r] = prior + rank;
    }
    __syncthreads();
    const int c0 = cnt[0][t], c1 = cnt[1][t], c2 = cnt[2][t], c3 = cnt[3][t];
    const uint32_t count = (uint32_t)(c0 + c1 + c2 + c3);
    cnt[0][t] = 0; cnt[1][t] = c0; cnt[2][t] = c0 + c1; cnt[3][t] = c0 + c1 + c2;   // first slot of wave w inside (digit, tile)
    // keys with a smaller digit anywhere in the input
    const int gh = ghist[t];
    const int incl = wave_inclusive_scan(gh);
    if (lane == 63) wsum[w] = incl;
    // hand the count on, find out whether this tile closes its group
    const int g = tile / FS_GROUP, gfirst = g * FS_GROUP;
    const int members = min(FS_GROUP, tiles - gfirst);
    st_agent(&agg[(size_t)tile * 256 + t], fs_word(tag, count));
    if (t == 0) s_last = atomicAdd(&gcnt[g], 1u) == (uint32_t)(members - 1) ? 1 : 0;
    __syncthreads();
    int gbase = incl - gh;
    for (int w2 = 0; w2 < w; ++w2) gbase += wsum[w2];
    if (s_last) {
        uint32_t sum;
        for (;;) {
            sum = 0u;
            bool ok = true;
#pragma unroll 8
            for (int j = 0; j < members; ++j) {
                const uint32_t v = ld_agent(&agg[(size_t)(gfirst + j) * 256 + t]);
                ok = ok && fs_ready(v, tag);
                sum += v & FS_VAL_MASK;
            }
            if (ok) break;
            spin_pause();
        }
        st_agent(&gtot[(size_t)g * 256 + t], fs_word(tag, sum));
    }
    uint32_t excl;
    for (;;) {
        excl = 0u;
        bool ok = true;
#pragma unroll 8
        for (int j = gfirst; j < tile; ++j) {
            const uint32_t v = ld_agent(&agg[(size_t)j * 256 + t]);
            ok = ok && fs_ready(v, tag);
            excl += v & FS_VAL_MASK;
        }
#pragma unroll 8
        for (int gg = 0; gg < g; ++gg) {
            const uint32_t v = ld_agent(&gtot[(size_t)gg * 256 + t]);
            ok = ok && fs_ready(v, tag);
            excl += v & FS_VAL_MASK;
        }
        if (ok) break;
        spin_pause();
    }
    base[t] = gbase + (int)excl;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int64_t i = first + r * 64 + lane;
        if (i < n) {
            const int d = (int)((key[r] >> shift) & 255u);
            const int pos = base[d] + cnt[w][d] + rk[r];
            kout[pos] = key[r];
            vout[pos] = FIRST ? (uint32_t)i : vin[i];
        }
    }
}

int sort_u32_fused(uint32_t* a_keys, uint32_t* a_vals, uint32_t* b_keys, uint32_t* b_vals, int64_t n, int key_bits,
                   const FusedWs& ws, hipStream_t stream) {
    if (!fused_sort_fits(n, key_bits)) return -1;
    const int passes = fused_passes(key_bits);
    const int tiles = fused_tiles(n);
    uint32_t *kin = a_keys, *vin = a_vals, *kout = b_keys, *vout = b_vals;
    for (int p = 0; p < passes; ++p) {
        if (p == 0)
            hipLaunchKernelGGL(fs_pass<true>, dim3(tiles), dim3(256), 0, stream, kin, vin, kout, vout, n, 8 * p, (uint32_t)(p + 1),
                               ws.ghist + 256 * p, ws.ticket + p, ws.gcnt + FS_MAX_GROUPS * p, ws.agg, ws.gtot);
        else
            hipLaunchKernelGGL(fs_pass<false>, dim3(tiles), dim3(256), 0, stream, kin, vin, kout, vout, n, 8 * p, (uint32_t)(p + 1),
                               ws.ghist + 256 * p, ws.ticket + p, ws.gcnt + FS_MAX_GROUPS * p, ws.agg, ws.gtot);
        if (hipGetLastError() != hipSuccess) return -3;
        uint32_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    return passes;
}

}  // namespace ml3d
