// kpbatch.hip — the WHOLE batch build of KPConv segmentation inference in one library call (gfx950 host orchestration).
//
// Replaces the per-layer loop of KPConvBatch.segmentation_inputs
// (ml3d/torch/dataloaders/concat_batcher.py:186-305): per layer the conv neighbours (radius r), the pooled points (grid
// dl = 2 r / conv_radius on a randomly oriented grid, kpconv.py:2037-2164), the pool neighbours (r) and the upsample neighbours
// (2 r), r doubling per layer -- the same kernels the Python loop of ml3d/torch/models/kpconv.py (KPConvBatch) strings
// together through ml3d_radius_dense_gather / _expand, ml3d_subsample_count / _fill and ml3d_rotate_points, so every matrix
// is bit-identical to that loop's.
//
// Why one call: round 4's profile put 6.0 ms of kernels into an 8.35 ms step -- the build is a DEPENDENT chain of ~300 launches
// cut by 9 blocking size read-backs, and between two read-backs the stream ran dry while the interpreter enqueued the next
// ~30 launches (10-20 us each through ctypes + torch allocations).  Here the chain is enqueued from C++ (no interpreter, no
// allocator: the caller hands in one workspace and one output arena), the row splits of the pooled levels are built ON the
// device (no per-level upload), and the read-backs are merged to ONE per layer:
//   sync l  reads {conv(l) longest row, subsample(l) count + per-item lengths, pool(l-1) / upsample(l-1) longest rows}
// i.e. the two searches of layer l - 1 that do not feed the chain are read one sync LATER, after the next layer's conv search and
// subsampling count have been enqueued behind them: L syncs per batch instead of 2 (L - 1) + 1.
//
// The library still allocates nothing and keeps no state: `workspace` (scratch, ml3d_kpconv_batch_workspace_bytes), `arena`
// (the results: pooled points and the dense int32 matrices, bump-allocated as their sizes become known; too small ->
// ML3D_E_WORKSPACE with the bytes needed so far in out->arena_used), `host_scratch` (pinned host memory for the read-backs).
// A row longer than `cap` (the stash width of the one-traversal search) returns ML3D_KPBATCH_FALLBACK: the caller runs its
// two-phase per-layer path for that batch (deformable layers, whose rows hold hundreds of neighbours, never come here).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "ml3d_hip.h"

namespace ml3d {

// splits[0] = 0, splits[i + 1] = sum_{j <= i} len[j]: one workgroup, thread t owns a contiguous chunk
__global__ void __launch_bounds__(256)
kpb_lens_to_splits(const int64_t* __restrict__ len, int batch, int64_t* __restrict__ splits) {
    __shared__ int64_t part[256];
    const int t = threadIdx.x;
    const int per = (batch + 255) / 256;
    const int b0 = t * per, b1 = min(batch, b0 + per);
    int64_t s = 0;
    for (int b = b0; b < b1; ++b) s += len[b];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        int64_t run = 0;
        for (int i = 0; i < 256; ++i) { const int64_t v = part[i]; part[i] = run; run += v; }
        splits[0] = 0;
    }
    __syncthreads();
    int64_t run = part[t];
    for (int b = b0; b < b1; ++b) { run += len[b]; splits[b + 1] = run; }
}

static inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace ml3d

using namespace ml3d;

// per-parity scratch set: layer l uses set l & 1, so the searches of layer l that are read one sync later (pool / upsample)
// keep their stashes while layer l + 1's conv search and subsampling already run in the other set
struct KpbSet {
    char* conv_ws;  size_t conv_wsb;     // grid over the layer's points + stash [n, cap]  (conv search, then the pool search)
    char* up_ws;    size_t up_wsb;       // grid over the pooled points + stash [n, cap]   (upsample search)
    char* sub_ws;   size_t sub_wsb;      // subsampling keys / sort
    float* rot_in;                       // the layer's points in the pooling grid's orientation [n, 3]
    float* rot_out;                      // the pooled points in that orientation [n, 3]
};

static size_t kpb_set_bytes(int64_t n0, int64_t batch, int cap) {
    return al256(ml3d_radius_dense_workspace_bytes(n0, n0, batch, cap)) * 2 + al256(ml3d_subsample_workspace_bytes(n0, batch)) +
           2 * al256(sizeof(float) * 3 * (size_t)n0);
}

extern "C" size_t ml3d_kpconv_batch_workspace_bytes(int64_t n_points, int64_t batch, int num_layers, int cap) {
    if (n_points < 0 || batch <= 0 || num_layers <= 0 || num_layers > ML3D_KPBATCH_MAX_LAYERS || cap <= 0) return 0;
    if (ml3d_radius_dense_workspace_bytes(n_points, n_points, batch, cap) == 0) return 0;
    // two scratch sets + per layer: row splits int64[batch + 1], the device-side size record int64[8 + batch]
    return 2 * kpb_set_bytes(n_points, batch, cap) +
           (size_t)num_layers * (al256(8 * (size_t)(batch + 1)) + al256(8 * (size_t)(8 + batch))) + 1024;
}

extern "C" size_t ml3d_kpconv_batch_host_scratch_bytes(int64_t batch, int num_layers) {
    if (batch <= 0 || num_layers <= 0) return 0;
    return 8 * (size_t)(8 + batch) + 8 * (size_t)(batch + 1);          // one size record + the staging of layer 0's row splits
}

namespace {
struct Bump {
    char* base; size_t cap; size_t used;
    void* take(size_t bytes) {       // nullptr when the arena is too small (`used` still advances: the caller learns the need)
        const size_t at = al256(used);
        used = at + bytes;
        return used <= cap ? base + at : nullptr;
    }
};
}  // namespace

extern "C" int ml3d_kpconv_batch_build(const float* points, const int64_t* lengths_host, int64_t batch, int64_t n_points,
                                       const ML3DKpBatchDesc* desc, const float* const* rotations, void* arena,
                                       size_t arena_bytes, ML3DKpBatchOut* out, int32_t* out_lengths_host, void* workspace,
                                       size_t workspace_bytes, void* host_scratch, size_t host_scratch_bytes, void* stream) {
    if (!desc || !out || !lengths_host || !out_lengths_host || batch <= 0 || batch > 65535 || n_points < 0) return ML3D_E_INVALID;
    const int L = desc->num_layers, cap = desc->cap;
    if (L <= 0 || L > ML3D_KPBATCH_MAX_LAYERS || cap <= 0) return ML3D_E_INVALID;
    for (int l = 0; l < L; ++l)
        if (!(desc->radius[l] > 0.f) || (l + 1 < L && !(desc->dl[l] > 0.f))) return ML3D_E_INVALID;
    if (n_points > 0 && !points) return ML3D_E_INVALID;
    if (workspace_bytes < ml3d_kpconv_batch_workspace_bytes(n_points, batch, L, cap) || !workspace) return ML3D_E_WORKSPACE;
    if (host_scratch_bytes < ml3d_kpconv_batch_host_scratch_bytes(batch, L) || !host_scratch) return ML3D_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    memset(out, 0, sizeof(*out));
    out->num_layers = L;

    // ---- carve the workspace ---------------------------------------------------------------------------------------------------
    char* p = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    KpbSet S[2];
    const size_t rwb = al256(ml3d_radius_dense_workspace_bytes(n_points, n_points, batch, cap));
    const size_t swb = al256(ml3d_subsample_workspace_bytes(n_points, batch));
    const size_t pb = al256(sizeof(float) * 3 * (size_t)n_points);
    for (int k = 0; k < 2; ++k) {
        S[k].conv_ws = p; S[k].conv_wsb = rwb; p += rwb;
        S[k].up_ws = p;   S[k].up_wsb = rwb;   p += rwb;
        S[k].sub_ws = p;  S[k].sub_wsb = swb;  p += swb;
        S[k].rot_in = (float*)p;  p += pb;
        S[k].rot_out = (float*)p; p += pb;
    }
    int64_t* splits[ML3D_KPBATCH_MAX_LAYERS];
    int64_t* rec[ML3D_KPBATCH_MAX_LAYERS];     // device size record of sync l: [conv 2][sub 2][pool(l-1) 2][up(l-1) 2][lengths(l+1) batch]
    for (int l = 0; l < L; ++l) { splits[l] = (int64_t*)p; p += al256(8 * (size_t)(batch + 1)); }
    for (int l = 0; l < L; ++l) { rec[l] = (int64_t*)p;    p += al256(8 * (size_t)(8 + batch)); }
    const size_t rec_n = (size_t)(8 + batch);
    int64_t* hrec = (int64_t*)host_scratch;
    int64_t* hsplits = hrec + rec_n;            // staging of layer 0's row splits (read by the copy engine: never reused in this call)

    Bump A = {(char*)arena, arena ? arena_bytes : 0, 0};

    // ---- layer 0: the caller's points; its row splits from the host lengths (the one upload of the build) -----------------------
    {
        hsplits[0] = 0;
        int64_t sum = 0;
        for (int64_t b = 0; b < batch; ++b) {
            if (lengths_host[b] < 0) return ML3D_E_INVALID;
            sum += lengths_host[b];
            hsplits[b + 1] = sum;
            out_lengths_host[b] = (int32_t)lengths_host[b];
        }
        if (sum != n_points) return ML3D_E_INVALID;
        if (hipMemcpyAsync(splits[0], hsplits, 8 * (size_t)(batch + 1), hipMemcpyHostToDevice, st) != hipSuccess) return ML3D_E_LAUNCH;
    }
    const float* pts[ML3D_KPBATCH_MAX_LAYERS + 1];
    int64_t n[ML3D_KPBATCH_MAX_LAYERS + 1];
    pts[0] = points; n[0] = n_points;
    int rc;

    // phase A of layer l (scratch set l & 1): conv search -> stash + rec[l][0..1]; subsampling count -> rec[l][2..3], lengths of
    // level l + 1 -> rec[l][8..].  (rec[l][4..7] belong to the pool / upsample searches of layer l - 1, enqueued before this.)
    auto mark = [&](int l, int i) {       // measurement hook: layer 0 only, only when the caller handed events in
        if (l == 0 && desc->trace_events[i]) (void)hipEventRecord((hipEvent_t)desc->trace_events[i], st);
    };
    // the pooling of a layer runs one workgroup per item in LDS (ml3d_subsample_items_*: two launches instead of ~30) when every
    // item of the level has at most ml3d_subsample_items_max_points() points; larger items keep the sort-based two-phase op
    bool items_sub[ML3D_KPBATCH_MAX_LAYERS] = {};
    auto max_item = [&](int l) -> int64_t {
        int64_t m = 0;
        for (int64_t b = 0; b < batch; ++b) m = out_lengths_host[(size_t)l * (size_t)batch + b] > m ? out_lengths_host[(size_t)l * (size_t)batch + b] : m;
        return m;
    };
    auto phase_a = [&](int l) -> int {
        KpbSet& W = S[l & 1];
        if (desc->has_conv[l]) {
            mark(l, 0);
            rc = ml3d_radius_dense_gather(pts[l], splits[l], pts[l], splits[l], batch, n[l], n[l], desc->radius[l], cap, 0, rec[l] + 0,
                                          W.conv_ws, W.conv_wsb, st);
            if (rc) return rc;
            mark(l, 1);
        }
        if (l + 1 < L) {
            mark(l, 4);
            const float* src = pts[l];
            if (rotations && rotations[l]) {
                rc = ml3d_rotate_points(pts[l], splits[l], batch, n[l], rotations[l], 0, W.rot_in, st);
                if (rc) return rc;
                src = W.rot_in;
            }
            items_sub[l] = max_item(l) <= ml3d_subsample_items_max_points();
            if (items_sub[l])
                rc = ml3d_subsample_items_count(src, splits[l], batch, n[l], desc->dl[l], max_item(l), rec[l] + 8, rec[l] + 2, st);
            else
                rc = ml3d_subsample_count(src, splits[l], batch, n[l], desc->dl[l], rec[l] + 8, rec[l] + 2, W.sub_ws, W.sub_wsb, st);
            if (rc) return rc;
            mark(l, 5);
        }
        return 0;
    };

    (void)hipMemsetAsync(rec[0], 0, 8 * rec_n, st);
    rc = phase_a(0);
    if (rc) return rc;
    for (int l = 0; l < L; ++l) {
        KpbSet& W = S[l & 1];
        ML3DKpLayerOut& O = out->layer[l];
        // ---- the layer's one sync: conv(l), subsample(l) [+ lengths of level l + 1], pool(l-1), upsample(l-1) -----------------
        if (hipMemcpyAsync(hrec, rec[l], 8 * rec_n, hipMemcpyDeviceToHost, st) != hipSuccess) return ML3D_E_LAUNCH;
        if (hipStreamSynchronize(st) != hipSuccess) return ML3D_E_LAUNCH;
        ++out->host_syncs;
        O.n_points = n[l];
        O.points_offset = l == 0 ? -1 : (int64_t)((const char*)pts[l] - (const char*)arena);
        O.conv_offset = O.pool_offset = O.up_offset = -1;
        if ((l > 0 && (hrec[4] || hrec[6])) || (desc->has_conv[l] && hrec[0])) return ML3D_KPBATCH_FALLBACK;   // a row longer than `cap`
        if (l > 0) {
            // the previous layer's pool / upsample searches: their longest rows are known now -> the dense matrices
            KpbSet& V = S[(l - 1) & 1];
            ML3DKpLayerOut& Q = out->layer[l - 1];
            Q.pool_cols = hrec[5]; Q.up_cols = hrec[7];
            if (n[l] > 0 && Q.pool_cols > 0) {
                int32_t* m = (int32_t*)A.take(4 * (size_t)n[l] * (size_t)Q.pool_cols);
                if (!m) { out->arena_used = (int64_t)A.used; return ML3D_E_WORKSPACE; }
                rc = ml3d_radius_dense_expand(n[l - 1], n[l], batch, cap, Q.pool_cols, (int32_t)n[l - 1], m, V.conv_ws, V.conv_wsb, st);
                if (rc) return rc;
                Q.pool_offset = (int64_t)((char*)m - (char*)arena);
            }
            if (n[l - 1] > 0 && Q.up_cols > 0) {
                int32_t* m = (int32_t*)A.take(4 * (size_t)n[l - 1] * (size_t)Q.up_cols);
                if (!m) { out->arena_used = (int64_t)A.used; return ML3D_E_WORKSPACE; }
                rc = ml3d_radius_dense_expand(n[l], n[l - 1], batch, cap, Q.up_cols, (int32_t)n[l], m, V.up_ws, V.up_wsb, st);
                if (rc) return rc;
                Q.up_offset = (int64_t)((char*)m - (char*)arena);
            }
        }
        if (desc->has_conv[l]) {
            O.conv_cols = hrec[1];
            if (n[l] > 0 && O.conv_cols > 0) {
                int32_t* m = (int32_t*)A.take(4 * (size_t)n[l] * (size_t)O.conv_cols);
                if (!m) { out->arena_used = (int64_t)A.used; return ML3D_E_WORKSPACE; }
                mark(l, 2);
                rc = ml3d_radius_dense_expand(n[l], n[l], batch, cap, O.conv_cols, (int32_t)n[l], m, W.conv_ws, W.conv_wsb, st);
                if (rc) return rc;
                mark(l, 3);
                O.conv_offset = (int64_t)((char*)m - (char*)arena);
            }
        }
        if (l + 1 == L) break;
        // ---- pooled level l + 1 ------------------------------------------------------------------------------------------------
        if (hrec[3] == 2) return ML3D_KPBATCH_FALLBACK;    // an item's grid has more cells than the per-item kernel's bitmap: per-layer path
        if (hrec[3]) return ML3D_E_UNSUPPORTED;            // an item spans >= 2^40 voxels at this dl
        const int64_t m_next = hrec[2];
        n[l + 1] = m_next;
        for (int64_t b = 0; b < batch; ++b) out_lengths_host[(size_t)(l + 1) * (size_t)batch + b] = (int32_t)hrec[8 + b];
        float* pool_p = (float*)A.take(sizeof(float) * 3 * (size_t)m_next);
        if (!pool_p) { out->arena_used = (int64_t)A.used; return ML3D_E_WORKSPACE; }
        hipLaunchKernelGGL(kpb_lens_to_splits, dim3(1), dim3(256), 0, st, rec[l] + 8, (int)batch, splits[l + 1]);
        if (hipGetLastError() != hipSuccess) return ML3D_E_LAUNCH;
        const bool rot = rotations && rotations[l];
        mark(l, 6);
        if (items_sub[l])
            rc = ml3d_subsample_items_fill(rot ? W.rot_in : pts[l], splits[l], batch, n[l], desc->dl[l], max_item(l), rec[l] + 8,
                                           rot ? W.rot_out : pool_p, st);
        else
            rc = ml3d_subsample_fill(rot ? W.rot_in : pts[l], nullptr, 0, nullptr, batch, n[l], rot ? W.rot_out : pool_p, nullptr, nullptr,
                                     W.sub_ws, W.sub_wsb, st);
        if (rc) return rc;
        if (rot) {
            rc = ml3d_rotate_points(W.rot_out, splits[l + 1], batch, m_next, rotations[l], 1, pool_p, st);
            if (rc) return rc;
        }
        mark(l, 7);
        pts[l + 1] = pool_p;
        // pool search: the pooled points against this layer's points (the conv search's grid when there was one: same supports,
        // same radius, and its stash has just been expanded); upsample search: this layer's points against the pooled ones, 2 r.
        // Their sizes go into the NEXT layer's record and are read with its sync.
        (void)hipMemsetAsync(rec[l + 1], 0, 8 * rec_n, st);
        rc = ml3d_radius_dense_gather(pts[l], splits[l], pool_p, splits[l + 1], batch, n[l], m_next, desc->radius[l], cap,
                                      desc->has_conv[l] ? 1 : 0, rec[l + 1] + 4, W.conv_ws, W.conv_wsb, st);
        if (rc) return rc;
        rc = ml3d_radius_dense_gather(pool_p, splits[l + 1], pts[l], splits[l], batch, m_next, n[l], 2.f * desc->radius[l], cap, 0,
                                      rec[l + 1] + 6, W.up_ws, W.up_wsb, st);
        if (rc) return rc;
        rc = phase_a(l + 1);                               // the next layer's conv search + subsampling count, other scratch set
        if (rc) return rc;
    }
    out->arena_used = (int64_t)A.used;
    return 0;
}
