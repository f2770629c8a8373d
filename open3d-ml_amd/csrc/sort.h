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

#include <gfx950_ops.h>

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

// ---- fused sort of 32-bit keys for SHORT inputs (round 6; voxelize) ---------------------------------------------------------------
// The pass above is three to five launches (block histograms -> scan of the [256][blocks] table -> scatter), and a voxelize call
// strings 28 of them together: at 1-2 M points every launch is a few microseconds of work behind a dependent-dispatch latency of
// the same size.  Here a pass is ONE launch: a tile (2048 keys) counts its digits, hands its 256 counts to
// the tiles behind it through a table in global memory, takes its own first slots from the tiles in front of it, and scatters.
// The hand-off is a flat two-level prefix rather than a chained look-back: at these sizes EVERY tile of the launch is resident at
// once, so a chain (tile j waits for j - 1) costs one memory round trip per link; here tile j sums the counts of the <= 31 tiles
// before it in its GROUP of 32 and the totals of the <= 63 groups before its own (a group's total is published by whichever of its
// tiles arrives last) -- three dependent round trips whatever the tile count.  No tile ever waits for a LATER tile, tiles take
// their numbers from a ticket in dispatch order, so the launch makes progress with any number of resident workgroups (two such
// launches on two streams cannot block each other the way two grid barriers could).  Words in the table carry value and ready tag
// in one dword (ld_agent / st_agent, gfx950_ops.h): no fences.  Limits: n <= FS_MAX_TILES * FS_TILE (4.2 M), keys of <= 32 bits.
constexpr int FS_TILE = 2048;
constexpr int FS_GROUP = 32;
constexpr int FS_MAX_GROUPS = 64;
constexpr int FS_MAX_TILES = FS_GROUP * FS_MAX_GROUPS;
constexpr uint32_t FS_VAL_MASK = (1u << 29) - 1u;      // value bits of a hand-off word; the tag sits above

struct FusedWs {             // ONE region, zeroed once per call (fused_ws_zero)
    int* ghist;              // [4][256]   digit histograms of the whole input, all passes (filled by the caller's key kernel)
    uint32_t* ticket;        // [8]        tile tickets: passes 0-3, [4] the caller's grouping kernel
    uint32_t* gcnt;          // [6][64]    arrivals per group: passes 0-3, [4] / [5] the grouping kernel's two stages
    uint32_t* agg;           // [tiles][256]  digit counts of a tile (re-used by every pass: the tag is the pass number + 1)
    uint32_t* gtot;          // [64][256]     digit counts of a group
    uint32_t* tA;            // [tiles][4]    grouping kernel, stage 1 (voxel.hip)
    uint32_t* gA;            // [64][4]
    uint32_t* tC;            // [tiles][4]    grouping kernel, stage 2
    uint32_t* gC;            // [64][4]
    uint32_t* fvf;           // [batch + 2]   first voxel ordinal of a batch item, tagged
    char* base;
    size_t bytes;
    int tiles;
};
__device__ __forceinline__ uint32_t fs_word(uint32_t tag, uint32_t v) { return (tag << 29) | v; }
__device__ __forceinline__ bool fs_ready(uint32_t w, uint32_t tag) { return (w >> 29) == tag; }
// value of a hand-off word once its tag stands (a single word: the callers that need many words batch their loads instead)
__device__ __forceinline__ uint32_t fs_wait(const uint32_t* p, uint32_t tag) {
    uint32_t w = ld_agent(p);
    while (!fs_ready(w, tag)) { spin_pause(); w = ld_agent(p); }
    return w & FS_VAL_MASK;
}
inline int fused_tiles(int64_t n) { return (int)((n + FS_TILE - 1) / FS_TILE); }
inline bool fused_sort_fits(int64_t n, int key_bits) { return n > 0 && key_bits <= 32 && fused_tiles(n) <= FS_MAX_TILES; }
size_t fused_ws_bytes(int64_t n, int64_t batch);
bool fused_ws_carve(void* ws, size_t bytes, int64_t n, int64_t batch, FusedWs* out);
// Sorts (a_keys, position) pairs ascending by the low key_bits bits, stable; the first pass takes the POSITION of a key as its value
// (no value array is read).  ghist must hold the digit histograms (8 bits per pass, pass p at ghist + 256 p).  Returns the
// number of passes: odd -> the result lies in (b_keys, b_vals), even -> in (a_keys, a_vals).  < 0: launch failure.
int sort_u32_fused(uint32_t* a_keys, uint32_t* a_vals, uint32_t* b_keys, uint32_t* b_vals, int64_t n, int key_bits,
                   const FusedWs& ws, hipStream_t stream);
inline int fused_passes(int key_bits) { return key_bits <= 8 ? 1 : (key_bits + 7) / 8; }


}  // namespace ml3d
