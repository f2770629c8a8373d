// sort.h — stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass (gfx950).
//
// Used by voxelize / grid-subsample to group points by voxel key while keeping the
// ORIGINAL point order inside a voxel (the canonical order of the oracle, and what makes
// the float barycentre sums reproducible bit for bit).  Per pass: block histogram ->
// scan of the [256][blocks] table -> stable scatter (wave-level multisplit by ballot).
// HBM traffic per pass: 2 x 12 B/pair read + 12 B/pair written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ml3d {

typedef unsigned long long u64;

struct SortWs {
    u64* keys_alt;
    uint32_t* vals_alt;
    int* hist;        // 1 + 256 * blocks
    int* block_sums;  // scan scratch
    int64_t n;
};

size_t sort_ws_bytes(int64_t n);
bool sort_ws_carve(void* ws, size_t bytes, int64_t n, SortWs* out);
// Sorts ascending by the low `key_bits` bits of the key, stable.  The result is left in
// (keys, vals); an odd number of passes is rounded up so no copy-back is needed.
// allow_odd: an odd number of passes is NOT rounded up; the result then lies in (ws.keys_alt, ws.vals_alt) -- sort_result_in_alt(n,
// key_bits) tells (a pure function of its arguments, so a later call that re-carves the same workspace finds the result again).
int sort_pairs_u64(u64* keys, uint32_t* vals, int64_t n, int key_bits, const SortWs& ws, hipStream_t stream, bool allow_odd = false);
bool sort_result_in_alt(int64_t n, int key_bits);

}  // namespace ml3d
