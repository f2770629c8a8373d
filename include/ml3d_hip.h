/*
 * ml3d_hip.h — C ABI of libml3d_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the point-cloud inference hot path of isl-org/Open3D-ML.
 * The reference has no native code of its own: every primitive below is what
 * its Python imports from the un-vendored `open3d` wheel,
 * so each entry point cites the reference CALL SITE whose native op it
 * replaces (paths relative to the reference checkout).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM) unless the name ends in `_host`;
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *    nothing synchronises — results are ready when the stream is;
 *  - the library never allocates: scratch comes from the caller
 *    (`*_workspace_bytes` tells how much), ragged results are two-phase
 *    (count -> caller allocates -> fill);
 *  - return value 0 = ok, <0 = ML3D_E_* (the Python wrappers raise RuntimeError);
 *  - no global state: re-entrant, one stream per call; no environment reads;
 *  - `nothing synchronises` has ONE exception, ml3d_kpconv_batch_build, whose
 *    point is to do a batch build's size read-backs inside the call;
 *  - HIP graph capture (hipStreamBeginCapture on `stream`): the RandLA path
 *    (ml3d_randla_knn_pyramid*, ml3d_randla_forward*, ml3d_nearest_to_center_dev,
 *    ml3d_patch_crop / _recenter, ml3d_vote_update) contains kernel nodes only
 *    and replays correctly (tests/test_gpu_api.py).  Other entry points clear
 *    buffers with hipMemsetAsync, and on ROCm 7.2 a captured graph's memset
 *    node was observed NOT to clear its target on later replays:
 *    do not capture them until that is fixed upstream.
 *  - index results are bit-exact w.r.t. the canonical orders documented in
 *    oracle/ml3d_oracle.c; float results are within 1e-4 abs of the reference
 *    PyTorch-CPU forward.
 */
#ifndef ML3D_HIP_H
#define ML3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ML3D_OK 0
#define ML3D_E_INVALID (-1)   /* bad argument (shape, k, null pointer) */
#define ML3D_E_WORKSPACE (-2) /* workspace too small                   */
#define ML3D_E_LAUNCH (-3)    /* hipGetLastError() != hipSuccess        */
#define ML3D_E_UNSUPPORTED (-4)

/* Bumped whenever a signature or a struct of this header changes (2: ml3d_radius_fill takes a spill buffer).  The Python binding   */
/* refuses a library whose version differs from the header it was written against: a stale .so would misread its arguments.        */
#define ML3D_ABI_VERSION 12
int ml3d_abi_version(void);

/* ------------------------------------------------------------------------- */
/* exact k-NN on a counting-sorted uniform grid                               */
/* replaces o3c.nns.NearestNeighborSearch(...).knn_search(q, k)               */
/*   ml3d/datasets/utils/dataprocessing.py:99-103  (callers                   */
/*   ml3d/torch/models/randlanet.py:220,224) and the batched torch op         */
/*   open3d.ml.torch.ops.knn_search (ml3d/torch/models/point_transformer.py:  */
/*   724-729).                                                                */
/* points  [n_points, 3] f32, batch items delimited by points_row_splits      */
/* queries [n_queries,3] f32, delimited by queries_row_splits (int64[batch+1])*/
/* out_index [n_queries, k] i32, ascending (d2, index); rows of an item with   */
/*   fewer than k points are padded with -1 / +inf.                            */
/* out_dist2 may be NULL.  index_local != 0 -> indices relative to the item's  */
/*   first point, else global row numbers into `points`.                       */
/* queries == points && same splits is recognised as the self-query case.      */
/* ------------------------------------------------------------------------- */
size_t ml3d_knn_workspace_bytes(int64_t n_points, int64_t n_queries, int64_t batch);

int ml3d_knn_search(const float* points, const int64_t* points_row_splits,
                    const float* queries, const int64_t* queries_row_splits,
                    int64_t batch, int64_t n_points, int64_t n_queries, int k,
                    int index_local, int32_t* out_index, float* out_dist2,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* batched fixed-radius search (two-phase ragged result)                       */
/* replaces open3d.ml.torch.layers.FixedRadiusSearch()(supports, queries, r,   */
/*   s_splits, q_splits) — ml3d/torch/models/kpconv.py:2021-2026               */
/*   (batch_neighbors; 3 calls per layer from KPConvBatch.segmentation_inputs, */
/*   ml3d/torch/dataloaders/concat_batcher.py:186-305).                        */
/* neighbour iff d2 <= radius*radius (f32, no fma); a row lists its neighbours */
/* ascending (d2, index).  Indices are GLOBAL rows of `points` unless          */
/* index_local != 0.                                                           */
/*  count: builds the grid in `workspace`, writes neighbors_row_splits         */
/*         int64[n_queries + 1] and out_stats int64[2] = {total, longest row}. */
/*         The scan of the counts is int32: 2^31 or more neighbours in one     */
/*         call set out_stats[1] = 2^63 (an impossible row length) -- callers   */
/*         must treat that as ML3D_E_UNSUPPORTED and split the batch.           */
/*  fill : must get the SAME workspace (address and contents) back.  Rows      */
/*         longer than the LDS buffer sort in `spill` (>= 8 * total + 8 bytes); */
/*         spill == NULL puts that scratch at the tail of a workspace of       */
/*         ml3d_radius_workspace_bytes(.., total) bytes instead.               */
/*         dense_cols == 0 -> ragged out_index[total];                         */
/*         dense_cols  > 0 -> the ragged_to_dense of kpconv.py:2030-2032 fused */
/*         in: out_index[n_queries, dense_cols], rows truncated / padded with  */
/*         pad_value.  out_dist2 may be NULL.                                  */
/* ------------------------------------------------------------------------- */
size_t ml3d_radius_workspace_bytes(int64_t n_points, int64_t n_queries, int64_t batch,
                                   int64_t total_neighbors);

int ml3d_radius_count(const float* points, const int64_t* points_row_splits,
                      const float* queries, const int64_t* queries_row_splits,
                      int64_t batch, int64_t n_points, int64_t n_queries, float radius,
                      int64_t* out_row_splits, int64_t* out_stats,
                      void* workspace, size_t workspace_bytes, void* stream);

int ml3d_radius_fill(const float* points, const int64_t* points_row_splits,
                     const float* queries, const int64_t* queries_row_splits,
                     int64_t batch, int64_t n_points, int64_t n_queries, float radius,
                     const int64_t* row_splits, int64_t total_neighbors, int index_local,
                     int64_t dense_cols, int32_t pad_value, int32_t* out_index, float* out_dist2,
                     void* workspace, size_t workspace_bytes, void* spill, size_t spill_bytes,
                     void* stream);

/* ------------------------------------------------------------------------- */
/* ragged_to_dense — replaces open3d.ml.torch.ops.ragged_to_dense              */
/*   (ml3d/torch/models/kpconv.py:2030, ml3d/torch/models/point_pillars.py:364)*/
/* values [K, elem_bytes] (elem_bytes multiple of 4), row_splits int64[rows+1],*/
/* out [rows, out_cols, elem_bytes]; row r = values[rs[r] : rs[r] + out_cols]  */
/* padded with default_value (device, elem_bytes).                             */
/* ------------------------------------------------------------------------- */
/* batch_neighbors (ml3d/torch/models/kpconv.py:2002-2034) in ONE traversal:   */
/* the dense [n_queries, longest] int32 matrix of the KPConv batcher, padded   */
/* with the shadow index.  ml3d_radius_dense_gather searches once and parks    */
/* every row -- canonical (d2, index) order, GLOBAL indices -- in a stash of   */
/* `cap` (<= 256) entries per query inside the workspace; out_stats [2] int64  */
/* = {overflow flag, longest row}.  The caller reads them (its one host sync   */
/* per search, the reference's .item() at kpconv.py:2028), allocates           */
/* [n_queries, dense_cols <= cap] and calls ml3d_radius_dense_expand (a copy + */
/* pad stream).  overflow != 0 (a row longer than cap): use ml3d_radius_count  */
/* + ml3d_radius_fill for that search.  reuse_grid != 0: `workspace` is the    */
/* one an earlier gather over the SAME points / row splits / radius used and   */
/* whose expand has been enqueued -- its grid is searched again with other     */
/* queries (the conv and the pool search of a KPConv layer).                   */
/* ------------------------------------------------------------------------- */
size_t ml3d_radius_dense_workspace_bytes(int64_t n_points, int64_t n_queries, int64_t batch, int cap);

int ml3d_radius_dense_gather(const float* points, const int64_t* points_row_splits,
                             const float* queries, const int64_t* queries_row_splits,
                             int64_t batch, int64_t n_points, int64_t n_queries, float radius,
                             int cap, int reuse_grid, int64_t* out_stats, void* workspace,
                             size_t workspace_bytes, void* stream);

int ml3d_radius_dense_expand(int64_t n_points, int64_t n_queries, int64_t batch, int cap,
                             int64_t dense_cols, int32_t pad_value, int32_t* out_index,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
int ml3d_ragged_to_dense(const void* values, const int64_t* row_splits, int64_t rows,
                         int64_t out_cols, int64_t elem_bytes, const void* default_value,
                         void* out, void* stream);

/* ------------------------------------------------------------------------- */
/* voxelize (two-phase) — replaces open3d.ml.torch.ops.voxelize                */
/*   ml3d/torch/models/point_pillars.py:354-357.                               */
/* points: rows of `point_stride` floats whose first 3 are xyz (the reference  */
/* passes the view points[:, :3] of an [N,4] tensor — no copy needed here);    */
/* keep iff min <= p <= max; coord = (int)((p - min) / voxel_size);            */
/* voxels ascending linear id x + X*(y + Y*z) per batch item, points inside a  */
/* voxel in original order, first max_points_per_voxel kept, first max_voxels  */
/* voxels per item kept.  voxel_size / range_* are HOST float[3] (the          */
/* reference keeps them as CPU tensors, point_pillars.py:317-320).             */
/*  count: out_batch_splits int64[batch+1], out_stats int64[2] =               */
/*         {n_voxels M, n_point_indices K}.                                    */
/*  fill : voxel_coords int32[M,3] (x,y,z), point_indices int64[K] (global     */
/*         rows), point_row_splits int64[M+1].  Same workspace as count.       */
/* ------------------------------------------------------------------------- */
size_t ml3d_voxelize_workspace_bytes(int64_t n_points, int64_t batch);

int ml3d_voxelize_count(const float* points, int64_t point_stride, const int64_t* row_splits,
                        int64_t batch, int64_t n_points, const float* voxel_size_host,
                        const float* range_min_host, const float* range_max_host,
                        int64_t max_points_per_voxel, int64_t max_voxels,
                        int64_t* out_batch_splits, int64_t* out_stats,
                        void* workspace, size_t workspace_bytes, void* stream);

int ml3d_voxelize_fill(int64_t batch, int64_t n_points, const float* voxel_size_host,
                       const float* range_min_host, const float* range_max_host,
                       int64_t max_points_per_voxel, int64_t max_voxels,
                       const int64_t* batch_splits, int32_t* out_voxel_coords,
                       int64_t* out_point_indices, int64_t* out_point_row_splits,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* grid subsample (two-phase) — replaces open3d.ml.contrib.subsample /         */
/*   subsample_batch: ml3d/datasets/utils/dataprocessing.py:32-49,             */
/*   ml3d/torch/models/kpconv.py:2098-2155.                                    */
/* Per batch item: origin = floor(min/dl)*dl, voxel = floor((p-origin)/dl),    */
/* output = barycentre (float32 sums in original point order / count),         */
/* feature mean, majority label (ties: smallest); voxels ascending linear key. */
/*  count: out_lengths int64[batch], out_stats int64[2] = {total M, error}     */
/*         (error != 0: an item spans >= 2^40 voxels — unsupported).           */
/*  fill : out_points [M,3], out_features [M,feature_dim] / out_labels [M]     */
/*         when the inputs are non-NULL.  Same workspace as count.             */
/* ------------------------------------------------------------------------- */
size_t ml3d_subsample_workspace_bytes(int64_t n_points, int64_t batch);

int ml3d_subsample_count(const float* points, const int64_t* row_splits, int64_t batch,
                         int64_t n_points, float sample_dl, int64_t* out_lengths,
                         int64_t* out_stats, void* workspace, size_t workspace_bytes,
                         void* stream);

int ml3d_subsample_fill(const float* points, const float* features, int64_t feature_dim,
                        const int32_t* labels, int64_t batch, int64_t n_points,
                        float* out_points, float* out_features, int32_t* out_labels,
                        void* workspace, size_t workspace_bytes, void* stream);

/* The same op with ONE WORKGROUP PER BATCH ITEM and no workspace (ABI 12): what the KPConv     */
/* batch build calls ~4 times per batch on ~100 spheres (concat_batcher.py:243-247 ->         */
/* kpconv.py:2037-2164).  The item's bounding box, an occupancy bitmap of its grid, the voxel  */
/* ordinals (popcount prefix: ascending key order without a sort), the grouping and the        */
/* float32 sums in original point order all happen in LDS; count and fill are one launch each  */
/* (the fill repeats the grouping).  Points only (no features / labels).  Limits per item:     */
/* ml3d_subsample_items_max_points() points -- pass the largest item length in                 */
/* max_item_points (HOST value, the same to count and fill: it picks one of three size classes */
/* -- 1024 / 4096 / 12 288 points with 16 384 / 65 536 / 262 144 grid cells, 13 / 49 / 145 KB  */
/* of LDS; ML3D_E_UNSUPPORTED when it is larger: use ml3d_subsample_count); the grid size is   */
/* checked on the device: out_stats[1] == 2 after the count means "an item's grid has more     */
/* cells than the class takes: repeat the call with ml3d_subsample_count".  Results identical  */
/* to ml3d_subsample_count / _fill bit for bit.                                                 */
/*  count: out_lengths int64[batch], out_stats int64[2] = {total M, 0 or 2}                     */
/*  fill : lengths = the count's out_lengths (device), out_points [M,3]                         */
int64_t ml3d_subsample_items_max_points(void);

int ml3d_subsample_items_count(const float* points, const int64_t* row_splits, int64_t batch,
                               int64_t n_points, float sample_dl, int64_t max_item_points,
                               int64_t* out_lengths, int64_t* out_stats, void* stream);

int ml3d_subsample_items_fill(const float* points, const int64_t* row_splits, int64_t batch,
                              int64_t n_points, float sample_dl, int64_t max_item_points,
                              const int64_t* lengths, float* out_points, void* stream);

/* Per-item row-vector rotation p' = p . R[b] (transpose != 0: p . R[b]^T) around the grid   */
/* subsample of batch_grid_subsampling (random_grid_orient, kpconv.py:2059-2110); float32     */
/* products and sums rounded one by one like the numpy expression it replaces.                */
/* rotations [batch, 3, 3] (device), out [n_points, 3] (may not alias points).                */
int ml3d_rotate_points(const float* points, const int64_t* row_splits, int64_t batch,
                       int64_t n_points, const float* rotations, int transpose, float* out,
                       void* stream);

/* ------------------------------------------------------------------------- */
/* The WHOLE batch build of KPConv segmentation inference in one call          */
/*   replaces the per-layer loop of KPConvBatch.segmentation_inputs            */
/*   (ml3d/torch/dataloaders/concat_batcher.py:186-305): per layer l the conv  */
/*   neighbours (radius[l]), the pooled points (grid dl[l], optionally on a    */
/*   rotated grid: kpconv.py:2059-2110), the pool neighbours (radius[l]) and   */
/*   the upsample neighbours (2 radius[l]) as dense int32 matrices padded with */
/*   the shadow index, exactly what ml3d_radius_dense_gather / _expand,        */
/*   ml3d_subsample_count / _fill and ml3d_rotate_points produce layer by      */
/*   layer -- enqueued from C++ with ONE blocking size read-back per layer.    */
/* Layers 0 .. num_layers - 2 pool; the last one only has its conv search.     */
/*  points [n_points,3] (device), lengths_host int64[batch] (HOST),            */
/*  rotations: num_layers - 1 device pointers to float32 [batch,3,3] (or NULL  */
/*   / NULL entries: axis-aligned grids).                                      */
/*  arena (device, caller-owned): pooled points and matrices are placed in it  */
/*   as their sizes become known; `out` (HOST) gets byte offsets into it       */
/*   (-1: empty / layer 0's points are the caller's), rows and columns.        */
/*   out_lengths_host int32[num_layers * batch] (HOST): per-item point counts. */
/*  host_scratch: PINNED host memory (ml3d_kpconv_batch_host_scratch_bytes).   */
/* Returns 0, ML3D_E_WORKSPACE (arena too small: out->arena_used = bytes        */
/* needed so far -- retry with a larger arena), ML3D_KPBATCH_FALLBACK (a row   */
/* longer than desc->cap: use the two-phase per-layer searches for this batch) */
/* or another negative error.  Results identical to the per-layer calls.       */
/* ------------------------------------------------------------------------- */
#define ML3D_KPBATCH_MAX_LAYERS 8
#define ML3D_KPBATCH_FALLBACK 1
typedef struct {
    int32_t num_layers;
    int32_t cap;                                   /* stash width of the one-traversal search (<= 256; 128 in the Python path) */
    int32_t has_conv[ML3D_KPBATCH_MAX_LAYERS];     /* the layer has convolution blocks (concat_batcher.py:219-234)            */
    float radius[ML3D_KPBATCH_MAX_LAYERS];         /* conv / pool radius of layer l (the upsample search uses 2 radius[l])     */
    float dl[ML3D_KPBATCH_MAX_LAYERS];             /* pooling grid of layer l (unused for the last layer)                      */
    /* optional hipEvent_t handles (NULL: off) recorded on `stream` around four pieces of LAYER 0, for measurement:            */
    /* [0,1] conv search (grid build + gather)  [2,3] its expand  [4,5] subsampling count (+ rotation)  [6,7] subsampling fill */
    void* trace_events[8];
} ML3DKpBatchDesc;
typedef struct {
    int64_t n_points, points_offset;               /* level l: rows, byte offset of float32 [n,3] in the arena (-1: layer 0)   */
    int64_t conv_offset, conv_cols;                /* int32 [n_points, conv_cols]                                              */
    int64_t pool_offset, pool_cols;                /* int32 [n_points of level l+1, pool_cols]                                 */
    int64_t up_offset, up_cols;                    /* int32 [n_points, up_cols]                                                */
} ML3DKpLayerOut;
typedef struct {
    int32_t num_layers, host_syncs;
    int64_t arena_used;
    ML3DKpLayerOut layer[ML3D_KPBATCH_MAX_LAYERS];
} ML3DKpBatchOut;

size_t ml3d_kpconv_batch_workspace_bytes(int64_t n_points, int64_t batch, int num_layers, int cap);
size_t ml3d_kpconv_batch_host_scratch_bytes(int64_t batch, int num_layers);
int ml3d_kpconv_batch_build(const float* points, const int64_t* lengths_host, int64_t batch,
                            int64_t n_points, const ML3DKpBatchDesc* desc,
                            const float* const* rotations, void* arena, size_t arena_bytes,
                            ML3DKpBatchOut* out, int32_t* out_lengths_host, void* workspace,
                            size_t workspace_bytes, void* host_scratch, size_t host_scratch_bytes,
                            void* stream);

/* ------------------------------------------------------------------------- */
/* KPConv (rigid) inference blocks — BatchNorm folded by the caller            */
/* ------------------------------------------------------------------------- */
/* ml3d_kpconv_rigid replaces KPConv.forward, non-deformable branch            */
/*   (ml3d/torch/models/kpconv.py:1048-1068,1105-1118,1139-1159) followed by   */
/*   BatchNormBlock + LeakyReLU of SimpleBlock / ResnetBottleneckBlock         */
/*   (kpconv.py:1357-1358, 1448-1449):                                         */
/*   out[q] = act( sum_k (sum_h w[q,k,h] x[idx[q,h]]) @ W[k] + bias ),         */
/*   w = influence(|s[idx[q,h]] - q - kp[k]|) (0 constant, 1 linear, 2 gauss). */
/* q_pts [Nq,3], s_pts [Ns,3], neighb_inds int32 [Nq,H] (>= Ns = shadow),      */
/* features [Ns,cin], kernel_points [15,3], weights [15*cin, cout] (the        */
/* reference's [K,cin,cout] tensor, BN scale folded into the last axis),       */
/* bias [cout] (may be NULL), act 0 none / 1 leaky(slope) / 2 relu.            */
/* ------------------------------------------------------------------------- */
size_t ml3d_kpconv_workspace_bytes(int64_t n_queries, int cin, int cout, int num_kernel_points);

int ml3d_kpconv_rigid(const float* q_pts, const float* s_pts, const int32_t* neighb_inds,
                      int64_t n_queries, int64_t n_supports, int64_t max_neighbors,
                      const float* features, int cin, const float* kernel_points,
                      int num_kernel_points, float kp_extent, int kp_influence_mode,
                      const float* weights, const float* bias, int act, float slope, int cout,
                      float* out, void* workspace, size_t workspace_bytes, void* stream);

/* The same convolution with the [15 cin, cout] contraction on the bf16 matrix  */
/* pipe: `packed` = ml3d_gemm_pack_bf16x3(weights, 15 * cin, cout) (see the      */
/* PointPillars section: exact three-way bf16 split, float32-equivalent).  Both  */
/* matrices are passed: convolutions served by the fused small-channel kernels   */
/* (cin <= 32) and ineligible shapes use `weights`.                              */
int ml3d_kpconv_rigid_bf16x3(const float* q_pts, const float* s_pts, const int32_t* neighb_inds,
                             int64_t n_queries, int64_t n_supports, int64_t max_neighbors,
                             const float* features, int cin, const float* kernel_points,
                             int num_kernel_points, float kp_extent, int kp_influence_mode,
                             const float* weights, const void* packed, const float* bias, int act,
                             float slope, int cout, float* out, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ml3d_kpconv_deformable replaces the deformable branch of KPConv.forward                      */
/*   (kpconv.py:1011-1066, 1139-1159) once the inner convolution has run: offset_features        */
/*   [Nq, offset_dim] = offset_conv(q, s, idx, x) + offset_bias (one ml3d_kpconv_rigid call with */
/*   the inner weights, act 0); offset_dim = 45 (3 per kernel point, in units of KP_extent) or   */
/*   60 (+ 15 modulation logits, modulations = 2 sigmoid).  Kernel point k of query q sits at    */
/*   kernel_points[k] + offsets[q, k] * kp_extent; everything else as ml3d_kpconv_rigid, same    */
/*   workspace.  The neighbour pruning of kpconv.py:1071-1103 only removes neighbours whose      */
/*   LINEAR influence is zero, so it is not performed; kp_influence_mode != 1 (linear) and cin   */
/*   outside {16, 32, 64, 128, 256, 512} return ML3D_E_UNSUPPORTED.                              */
int ml3d_kpconv_deformable(const float* q_pts, const float* s_pts, const int32_t* neighb_inds,
                           int64_t n_queries, int64_t n_supports, int64_t max_neighbors,
                           const float* features, int cin, const float* kernel_points,
                           int num_kernel_points, float kp_extent, int kp_influence_mode,
                           const float* offset_features, int offset_dim, const float* weights,
                           const float* bias, int act, float slope, int cout, float* out,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- training side of the rigid KPConv (ABI 5) ---------------------------------- */
/* ml3d_kpconv_weighted: wf[q, k, c] = sum_h w[q, k, h] * features[inds[q, h], c] -- the first half of */
/*   ml3d_kpconv_rigid (kpconv.py:1105-1137) into a caller buffer [n_queries, 15 * cin]; with it     */
/*   KPConv.forward is `wf . weights` (kpconv.py:1139-1159) and its parameter gradient               */
/*   `wf^T . grad_out`, both plain GEMMs of the caller.                                              */
/* ml3d_kpconv_weighted_backward: the adjoint of that sum with respect to the features (the          */
/*   influences depend on geometry only -- kernel points are buffers, kpconv.py:959-963):            */
/*   grad_features[inds[q, h], c] += sum_k w[q, k, h] * grad_wf[q, k, c]; grad_features [n_supports, */
/*   cin] is ZEROED here, then accumulated with float atomics (loss.backward() of                   */
/*   semantic_segmentation.py:423 reaches it through ml3d.ops.KPConvFunction).                       */
int ml3d_kpconv_weighted(const float* q_pts, const float* s_pts, const int32_t* neighb_inds,
                         int64_t n_queries, int64_t n_supports, int64_t max_neighbors,
                         const float* features, int cin, const float* kernel_points,
                         int num_kernel_points, float kp_extent, int kp_influence_mode,
                         float* out_wf, void* stream);

int ml3d_kpconv_weighted_backward(const float* q_pts, const float* s_pts, const int32_t* neighb_inds,
                                  int64_t n_queries, int64_t n_supports, int64_t max_neighbors, int cin,
                                  const float* kernel_points, int num_kernel_points, float kp_extent,
                                  int kp_influence_mode, const float* grad_wf, float* grad_features,
                                  void* stream);

/* ml3d_linear: out = act([gather(a) | a2] @ weights_t + bias + residual) on f32 MFMA.       */
/* Replaces UnaryBlock (Linear + BatchNormBlock + LeakyReLU, kpconv.py:1288-1293), the       */
/* residual add of ResnetBottleneckBlock (kpconv.py:1461) and, with a_gather = upsamples[:,0]*/
/* and a2 = the skip features, NearestUpsampleBlock + torch.cat + UnaryBlock of the decoder  */
/* (kpconv.py:283-286, 821-838, 1468-1481).  a [*, lda] uses k1 columns, row m of the        */
/* product reads a[a_gather[m*a_gather_stride]] (rows >= a_rows are zeros) or a[m] when      */
/* a_gather is NULL; a2 [m, lda2] contributes k2 more columns; weights_t [k1+k2, n].         */
/* residual [*, ldr] is added before the activation: row m, or -- residual_gather != NULL -- */
/* row residual_gather[m * residual_gather_stride] when that lies in [0, residual_rows), else*/
/* nothing.  The latter is the decoder step split by linearity:                              */
/*   W . [x[up[m,0]] ; skip[m]] = (x W_x)[up[m,0]] + skip[m] W_skip                          */
/* -- the upsampled half is multiplied on the COARSE level (a quarter of the rows) and comes */
/* back through the upsampling index (shadow index = no coarse neighbour = zero features).   */
size_t ml3d_linear_workspace_bytes(int64_t m, int n, int k);

int ml3d_linear(const float* a, int64_t lda, int k1, const int32_t* a_gather,
                int64_t a_gather_stride, int64_t a_rows, const float* a2, int64_t lda2, int k2,
                const float* weights_t, const float* bias, const float* residual, int64_t ldr,
                const int32_t* residual_gather, int64_t residual_gather_stride,
                int64_t residual_rows, int act, float slope, float* out, int64_t ldc, int64_t m,
                int n, void* workspace, size_t workspace_bytes, void* stream);

/* ml3d_gather_pool: mode 0 = max_pool (kpconv.py:841-858, shadow rows count as zeros),     */
/* mode 1 = closest_pool (kpconv.py:821-838, feature of the FIRST listed neighbour).         */
int ml3d_gather_pool(const float* features, int64_t n_supports, int channels,
                     const int32_t* inds, int64_t n_queries, int64_t max_neighbors, int mode,
                     float* out, void* stream);


/* ------------------------------------------------------------------------- */
/* PointPillars inference blocks — BatchNorm folded by the caller              */
/* ------------------------------------------------------------------------- */
/* ml3d_pillar_features replaces, for a whole batch, the tail of               */
/*   PointPillarsVoxelization.forward (ragged_to_dense + feats[idx] gather +   */
/*   out-of-bounds filter, ml3d/torch/models/point_pillars.py:359-382),        */
/*   PillarFeatureNet.forward + PFNLayer.forward (:512-555, 417-453) and       */
/*   PointPillarsScatter.forward (:577-616).                                   */
/* Inputs are the ragged result of ml3d_voxelize_* (voxel_coords (x,y,z),      */
/* point_indices, point_row_splits, batch_splits) and the raw point rows       */
/* [N, point_stride] (first in_channels floats used, xyz first).  Pillars with */
/* x >= nx or y >= ny are dropped as the reference does.  Layer l:             */
/* weights_host[l] = folded Linear [c_in_l, units_l] (c_in_0 = in_channels+5,  */
/* c_in_l = 2*units_{l-1}), bias_host[l] [units_l] (HOST arrays of DEVICE      */
/* pointers).  x_offset = vx/2 + range_min_x, y_offset likewise.               */
/* canvas: NHWC [batch, ny, nx, canvas_channels] f32, zeroed by this call;     */
/* pixel (b, y, x) receives the pillar's feature (canvas_channels must equal   */
/* units of the last layer).                                                   */
/* ------------------------------------------------------------------------- */
size_t ml3d_pillar_features_workspace_bytes(int64_t n_pillars, int max_num_points,
                                            int num_layers, const int32_t* units_host);

int ml3d_pillar_features(const float* points, int64_t point_stride, int in_channels,
                         const int32_t* voxel_coords, const int64_t* point_indices,
                         const int64_t* point_row_splits, const int64_t* batch_splits,
                         int64_t batch, int64_t n_pillars, int max_num_points,
                         float vx, float vy, float x_offset, float y_offset, int nx, int ny,
                         int num_layers, const int32_t* units_host,
                         const float* const* weights_host, const float* const* bias_host,
                         float* canvas, int canvas_channels,
                         void* workspace, size_t workspace_bytes, void* stream);

/* Conv2d + folded BatchNorm + activation on an NHWC map (SECOND backbone,      */
/*   point_pillars.py:640-682), implicit GEMM on f32 MFMA.                      */
/* in [batch, h, w, cin] (cin % 4 == 0); weights [(ky*kw + kx)*cin + ci, cout]  */
/* (= the reference's [cout, cin, kh, kw] tensor permuted, BN scale folded);    */
/* out pixel (b, oy, ox) channels [0, cout) at out + pixel * out_pixel_stride.  */
size_t ml3d_conv2d_workspace_bytes(int64_t batch, int out_h, int out_w, int cin, int cout,
                                   int kh, int kw);

int ml3d_conv2d_nhwc(const float* in, int64_t batch, int h, int w, int cin,
                     const float* weights, const float* bias, int kh, int kw, int stride,
                     int pad, int act, float slope, int cout, float* out,
                     int64_t out_pixel_stride, void* workspace, size_t workspace_bytes,
                     void* stream);

/* The same convolution on the bf16 matrix pipe, float32-equivalent: both      */
/* operands are split exactly into three bf16 (x = h + m + l) and the product   */
/* is accumulated in float from the six largest cross terms -- error below the  */
/* float32 rounding of the sum itself ("bf16x3"), 2.7x the MFMA rate.*/
/* ml3d_gemm_pack_bf16x3 splits the SAME weight matrix ml3d_conv2d_nhwc takes   */
/* ([k = kh*kw*cin, n = cout] float, k % 32 == 0 else ML3D_E_UNSUPPORTED) once  */
/* into `packed` (ml3d_gemm_pack_bf16x3_bytes(k, n) bytes, 16-byte aligned).    */
/* ml3d_conv2d_nhwc_bf16x3: cin % 32 == 0 and 32-bit element offsets, else      */
/* ML3D_E_UNSUPPORTED (callers keep ml3d_conv2d_nhwc for those); needs no       */
/* workspace.  Non-finite inputs give NaN where the f32 kernel gives inf.       */
size_t ml3d_gemm_pack_bf16x3_bytes(int k, int n);

int ml3d_gemm_pack_bf16x3(const float* weights, int k, int n, void* packed, size_t packed_bytes,
                          void* stream);

int ml3d_conv2d_nhwc_bf16x3(const float* in, int64_t batch, int h, int w, int cin,
                            const void* packed, const float* bias, int kh, int kw, int stride,
                            int pad, int act, float slope, int cout, float* out,
                            int64_t out_pixel_stride, void* stream);

/* The same path for dense rows: ml3d_linear_bf16x3 =                            */
/*   act([a[rows, k1] | a2[rows, k2]] . W + bias + residual)                     */
/* with `packed` = ml3d_gemm_pack_bf16x3 of W [k1 + k2, n] (a2 may be NULL with  */
/* k2 = 0; each block float4-addressable: lda % 4 == 0, 16-byte aligned; k1 and  */
/* k1 + k2 multiples of 32; else ML3D_E_UNSUPPORTED -- callers keep ml3d_linear).*/
/* Small-rows / deep-k problems are split along k into `workspace`               */
/* (ml3d_linear_bf16x3_workspace_bytes; NULL: unsplit).                          */
/* ml3d_deconv2d_nhwc_bf16x3 = ml3d_deconv2d_nhwc (below) with `packed` of its   */
/* [cin, s*s*cout] matrix.                                                        */
size_t ml3d_linear_bf16x3_workspace_bytes(int64_t rows, int n, int k);

int ml3d_linear_bf16x3(const float* a, int64_t lda, int k1, const float* a2, int64_t lda2, int k2,
                       int64_t rows, const void* packed, const float* bias, const float* residual,
                       int64_t ldr, int n, int act, float slope, float* out, int64_t ldc,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ... and with a GATHERED residual (ABI 12): output row m adds residual[residual_gather[m * residual_gather_stride]] (global row     */
/* index; rows outside [0, residual_rows) add nothing) -- the decoder step of KPFCNN split by linearity,                                */
/* (x W_x)[up[:, 0]] + skip W_skip (NearestUpsampleBlock + torch.cat + UnaryBlock, kpconv.py:283-286, 821-838, 1468-1481).             */
int ml3d_linear_bf16x3_gathered(const float* a, int64_t lda, int k1, const float* a2, int64_t lda2, int k2,
                                int64_t rows, const void* packed, const float* bias, const float* residual,
                                int64_t ldr, const int32_t* residual_gather, int64_t residual_gather_stride,
                                int64_t residual_rows, int n, int act, float slope, float* out, int64_t ldc,
                                void* workspace, size_t workspace_bytes, void* stream);

int ml3d_deconv2d_nhwc_bf16x3(const float* in, int64_t batch, int h, int w, int cin,
                              const void* packed, const float* bias, int stride, int act,
                              float slope, int cout, float* out, int64_t out_pixel_stride,
                              void* stream);

/* ConvTranspose2d with kernel == stride + folded BN + activation (SECONDFPN    */
/*   deblocks, point_pillars.py:712-717, 749): GEMM + pixel-shuffle store.      */
/* weights [ci, (dy*stride + dx)*cout + co] (= the reference's [cin, cout, k, k]*/
/* permuted); out is an NHWC map [batch, h*stride, w*stride, *] with pixel      */
/* stride out_pixel_stride — pass out + channel_offset to write one slice of    */
/* the concatenated neck map.                                                   */
int ml3d_deconv2d_nhwc(const float* in, int64_t batch, int h, int w, int cin,
                       const float* weights, const float* bias, int stride, int act,
                       float slope, int cout, float* out, int64_t out_pixel_stride,
                       void* workspace, size_t workspace_bytes, void* stream);

/* channel slice of an NHWC map -> NCHW tensor (the layout Anchor3DHead.forward  */
/* returns, point_pillars.py:836-841)                                            */
int ml3d_nhwc_to_nchw(const float* in, int64_t in_pixel_stride, int channel_offset,
                      int channels, int64_t batch, int64_t hw, float* out, void* stream);


/* ------------------------------------------------------------------------- */
/* rotated-BEV NMS — replaces open3d.ml.torch.ops.nms                          */
/*   ml3d/torch/utils/objdet_helper.py:346 (multiclass_nms, called from        */
/*   Anchor3DHead.get_bboxes_single, ml3d/torch/models/point_pillars.py:1004). */
/* boxes [n,5] (x0,y0,x1,y1,r) f32, scores [n] f32.  Greedy by descending      */
/* score (ties: lower index first), a box is suppressed when its rotated IoU   */
/* with a kept box is > iou_threshold.  out_keep int64[n] receives the kept    */
/* indices in descending-score order, out_count int64[1] their number          */
/* (n <= 65536).                                                               */
/* ------------------------------------------------------------------------- */
size_t ml3d_nms_workspace_bytes(int64_t n);

int ml3d_nms(const float* boxes, const float* scores, int64_t n, float iou_threshold,
             int64_t* out_keep, int64_t* out_count, void* workspace, size_t workspace_bytes,
             void* stream);

/* ------------------------------------------------------------------------- */
/* Anchor3DHead.get_bboxes for a whole batch — replaces the */
/* per-sample, per-class loop of ml3d/torch/models/point_pillars.py:945-1025   */
/* (sigmoid / top nms_pre / BBoxCoder.decode, objdet_helper.py:286-313 /       */
/* multiclass_nms, objdet_helper.py:316-350 / direction fix) without a host    */
/* read-back per class.  Head maps: cls [B, A*C, H, W], reg [B, A*7, H, W],    */
/*   dir [B, A*2, H, W] in ANY strided layout -- each map comes with three     */
/*   ELEMENT strides (batch, channel, pixel): the reference's NCHW tensors     */
/*   (C*H*W, H*W, 1) or channel slices of a fused NHWC head tensor             */
/*   (H*W*Ctot, 1, Ctot); HOST arrays cls_strides[3] / strides9[9] = cls, reg, */
/*   dir.  anchors [H*W*A, 7] in (h, w, a) order                               */
/*   (Anchor3DRangeGenerator.grid_anchors).                                    */
/* ml3d_pp_anchor_scores: out_scores [B, H*W*A] = max_c sigmoid(cls) — the key */
/*   of the nms_pre top-k (point_pillars.py:985-992), which the caller takes.  */
/* ml3d_pp_boxes: candidates [B, k] int64 anchor indices (k <= 4096) ->        */
/*   out_rows [B, C*k, 9] f32 = (x, y, z, w, l, h, yaw, score, label) of the   */
/*   kept boxes, class-major and in NMS order like the reference's torch.cat,  */
/*   out_total [B] int32 = rows used per sample.  A candidate takes part in    */
/*   class c iff score_c > score_threshold; rotated-BEV IoU > iou_threshold    */
/*   suppresses (greedy, descending score, ties by candidate index).           */
/* ------------------------------------------------------------------------- */
int ml3d_pp_anchor_scores(const float* cls, const int64_t* cls_strides, int64_t batch,
                          int num_anchors, int num_classes, int64_t hw, float* out_scores,
                          void* stream);

size_t ml3d_pp_boxes_workspace_bytes(int64_t batch, int64_t k, int num_classes);

int ml3d_pp_boxes(const float* cls, const float* reg, const float* dir, const int64_t* strides9,
                  const float* anchors, const int64_t* candidates, int64_t batch, int64_t k,
                  int num_anchors, int num_classes, int64_t hw, float score_threshold,
                  float iou_threshold, float dir_offset, float* out_rows, int32_t* out_total,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* the nms_pre top-k of Anchor3DHead.get_bboxes_single (ABI 7) — replaces      */
/* `max_scores.topk(self.nms_pre)`, ml3d/torch/models/point_pillars.py:985-992 */
/* (the reference calls torch.topk per sample; this takes every sample of the */
/* batch in one call).  values [rows, n] f32 row-major, 0 <= k <= min(n, 4096) */
/* (k beyond 4096: ML3D_E_UNSUPPORTED; the reference's configs use 100..4096). */
/* out_index int64 [rows, k]: the k largest of every row in DESCENDING value,  */
/* equal values by ASCENDING index (also at the k-th value: of its ties the    */
/* lowest indices are taken — torch leaves that order unspecified), NaN above  */
/* +inf (torch.topk's convention).  out_value f32 [rows, k] or NULL.           */
/* ------------------------------------------------------------------------- */
size_t ml3d_topk_rows_workspace_bytes(int64_t rows, int64_t n, int64_t k);

int ml3d_topk_rows(const float* values, int64_t rows, int64_t n, int64_t k, int64_t* out_index,
                   float* out_value, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------- */
/* pairwise rotated box IoU for the detection metric —       */
/* replaces open3d.ml.contrib.iou_bev_{cpu,cuda} / iou_3d_{cpu,cuda}           */
/*   ml3d/metrics/mAP.py:85-88, ml3d/metrics/__init__.py:3-9,                  */
/*   ml3d/datasets/utils/operations.py:7.                                      */
/* iou_bev: boxes [n,5] / [m,5] = (x, z, w, l, yaw): centre and size in the    */
/*   ground plane of the camera frame.  iou_3d: boxes [n,7] / [m,7] =          */
/*   (x, y, z, w, h, l, yaw), y = bottom face (the y axis points down: the box */
/*   spans [y - h, y]).  out_iou float32 [n, m] row-major.  The rotated        */
/*   intersection is the NMS's (same corner / clipping arithmetic).            */
/* ------------------------------------------------------------------------- */
int ml3d_iou_bev(const float* boxes_a, const float* boxes_b, int64_t n, int64_t m, float* out_iou, void* stream);

int ml3d_iou_3d(const float* boxes_a, const float* boxes_b, int64_t n, int64_t m, float* out_iou, void* stream);

/* ------------------------------------------------------------------------- */
/* patch sampler / vote accumulation                         */
/* ml3d_nearest_to_center: the k points nearest to a centre — replaces         */
/*   search_tree.query(center_point, k=num_points) of                          */
/*   SemSegSpatiallyRegularSampler (ml3d/datasets/samplers/                    */
/*   semseg_spatially_regular.py:90-91); k may be as large as n_points.        */
/*   The reference's tree there is sklearn.neighbors.KDTree (randlanet.py:142, */
/*   kpconv.py:384): float64 copies of the float32 points, neighbours ordered  */
/*   by the float64 reduced distance (dx*dx + dy*dy) + dz*dz.  The ORDER is    */
/*   observable (random.shuffle of the patch, then prefix subsampling), so     */
/*   this entry orders by exactly that float64 value, ties by ascending index; */
/*   pinned against scikit-learn in tests/test_emulated_prims.py.              */
/*   center_host: HOST float[3].  out_dist2: DOUBLE [k] reduced distances      */
/*   (sklearn's distance = sqrt of it), may be NULL.                           */
/* ml3d_vote_update: probs[inds[i]] = smooth*probs[inds[i]] + (1-smooth)*      */
/*   softmax(logits[i]) on a float16 accumulator [n_cloud, classes] with the   */
/*   numpy promotion of ml3d/torch/models/randlanet.py:420-421, 457-462        */
/*   (float16 product, float32 sum, float16 store).  inds must be distinct    */
/*   (the caller keeps the last occurrence of a repeated point).               */
/* ------------------------------------------------------------------------- */
size_t ml3d_nearest_to_center_workspace_bytes(int64_t n_points);

int ml3d_nearest_to_center(const float* points, int64_t n_points, const float* center_host,
                           int64_t k, int32_t* out_index, double* out_dist2,
                           void* workspace, size_t workspace_bytes, void* stream);

int ml3d_vote_update(const float* logits, const int32_t* point_inds, int64_t n, int num_classes,
                     float smooth, void* probs_f16, int64_t n_cloud, void* stream);

/* ---- the patch loop of one cloud with NOTHING read back (ABI 5) ------------------------------- */
/* The sampler + transform of a test cloud (semseg_spatially_regular.py:64-111,                    */
/* randlanet.py:156-212) touch the whole patch on the host: crop, float32 distances to the centre, */
/* possibility bump, recentring.  These entries do the same arithmetic, in numpy's order, on device */
/* buffers, so that consecutive patches of a cloud queue up without a host synchronisation.         */
/* ml3d_nearest_to_center_dev: ml3d_nearest_to_center with the centre read from DEVICE memory       */
/*   (center_dev: float[3], e.g. the row of the cloud an argmin kernel picked).                     */
/* ml3d_patch_crop: patch row j = points[cand[perm[j]]] (cand: the k nearest in query order, perm:  */
/*   the host's shuffle of 0..k-1 -- random.shuffle / Generator.permutation draw no data);          */
/*   out_sel[j] = that cloud index; d_j = ((dx*dx)+(dy*dy))+(dz*dz) in float32 (np.sum over the     */
/*   three squares); possibility[out_sel[j]] += (float64)((1 - d_j / max_j d_j)^2) with the float32 */
/*   quotient / difference / square of semseg_spatially_regular.py:104-107.  Indices must be        */
/*   distinct (a cloud smaller than the patch stays on the host path).  scratch: k floats + 64 B.   */
/* ml3d_patch_recenter: pts[:, d] -= mean_d for every axis d in dims_mask (bit d), mean_d = the     */
/*   SEQUENTIAL float32 sum of column d over rows 0..k-1, divided by k -- numpy's mean(0) of a      */
/*   C-contiguous float32 [k, 3] array accumulates row by row, and the recentred coordinates feed   */
/*   the neighbour search, so the order is reproduced, not approximated (augmentation.py:16-24).    */
/*   Then features[j] = [pts[j] | (extra[j] - feat_bias) / feat_scale] (randlanet.py:203-206 with   */
/*   the 'normalize.feat' augmentation; extra / n_extra may be NULL / 0).  scratch: 16 bytes.       */
int ml3d_nearest_to_center_dev(const float* points, int64_t n_points, const float* center_dev,
                               int64_t k, int32_t* out_index, double* out_dist2,
                               void* workspace, size_t workspace_bytes, void* stream);

int ml3d_patch_crop(const float* points, int64_t n_points, const int32_t* cand, const int32_t* perm,
                    const float* center_dev, int64_t k, float* out_pts, int32_t* out_sel,
                    double* possibility, void* scratch, size_t scratch_bytes, void* stream);

int ml3d_patch_recenter(float* pts, int64_t k, int dims_mask, const float* extra, int n_extra,
                        float feat_bias, float feat_scale, float* out_features,
                        void* scratch, size_t scratch_bytes, void* stream);

/* ---- RandLA-Net random_sample as a differentiable op (training side, SURVEY.md §8 f4, ABI 5) ------------------ */
/* ml3d_randla_gather_max: out[b, i, c] = max_k features[b, pool_idx[b, i, k], c], i < n_out (randlanet.py:300-327;   */
/*   pool_idx = the level's WHOLE neighbour matrix [batch, n_in, 16] (batch-local indices): item b's pooling rows are  */
/*   its first n_out, row stride n_in (randlanet.py:222-223).                                                          */
/* ml3d_randla_gather_max_backward: grad_features [batch, n_in, c] (zeroed here) += grad_out at the FIRST maximal    */
/*   neighbour of every (b, i, c), float atomics (loss.backward() of semantic_segmentation.py:423 through              */
/*   ml3d.ops.GatherMaxFunction).                                                                                       */
int ml3d_randla_gather_max(const float* features, const int32_t* pool_idx, int64_t batch, int64_t n_in,
                           int64_t n_out, int channels, float* out, void* stream);

int ml3d_randla_gather_max_backward(const float* features, const int32_t* pool_idx, const float* grad_out,
                                    int64_t batch, int64_t n_in, int64_t n_out, int channels,
                                    float* grad_features, void* stream);

/* ---- RandLA-Net attentive pooling as a differentiable op (training side, SURVEY.md §8 f4, ABI 6) ------------------ */
/* ml3d_randla_attentive_pool: out[r, c] = sum_k softmax_k(scores[r, :, c])[k] * x[r, k, c] for point-major               */
/*   scores, x [rows, k, channels] (k <= 32): AttentivePooling.forward up to its SharedMLP (randlanet.py:622-637 --        */
/*   `scores = softmax(score_fn(x), dim=-2); features = sum(scores * x, dim=-2)` in the reference's channel-major layout). */
/* ml3d_randla_attentive_pool_backward: grad_x = p * grad_out, grad_scores = p * (x - out) * grad_out with p the softmax  */
/*   recomputed from `scores`; `out` = the forward result (ml3d.ops.AttentivePoolFunction).                               */
int ml3d_randla_attentive_pool(const float* scores, const float* x, int64_t rows, int k, int channels,
                               float* out, void* stream);

int ml3d_randla_attentive_pool_backward(const float* scores, const float* x, const float* out,
                                        const float* grad_out, int64_t rows, int k, int channels,
                                        float* grad_scores, float* grad_x, void* stream);

/* ml3d_argmax_labels: out_labels[i] = argmax_c scores[i, c] as uint8 (num_classes <= 256; first  */
/*   maximum, NaN = maximum, like torch.argmax) -- the predicted labels that leave the GPU          */
/*   (SURVEY.md §8d "forward + softmax/argmax", §8e gather of predictions): 1 byte per point.       */
int ml3d_argmax_labels(const float* scores, int64_t n, int num_classes, uint8_t* out_labels,
                       void* stream);

/* ------------------------------------------------------------------------- */
/* RandLA-Net neighbour pyramid: the whole loop of                             */
/*   ml3d/torch/models/randlanet.py:218-229 for a batch of equally sized       */
/*   clouds in ONE call.  Layer l has n_l = n_{l-1} / ratio[l-1] points, the   */
/*   sub-cloud being the PREFIX pc[:n_l] (randlanet.py:222).                   */
/* points [batch, n0, 3] f32.                                                  */
/* neighbor_idx[l] [batch, n_l, k] i32  (k-NN of layer l onto itself)          */
/* interp_idx[l]   [batch, n_l, 1] i32  (1-NN of layer l in layer l+1)         */
/* sub_idx[l] is the prefix neighbor_idx[l][:, :n_{l+1}] — not materialised.   */
/* All indices are item-local, as RandLANet.forward consumes them.             */
/* The pointer tables neighbor_idx_host / interp_idx_host are HOST arrays of   */
/* device pointers, length num_layers.  1 <= num_layers <= 8                   */
/* (ML3D_RANDLA_MAX_LAYERS; the reference's configs use 4 or 5): more is       */
/* ML3D_E_INVALID / 0 workspace bytes.                                         */
/* ------------------------------------------------------------------------- */
size_t ml3d_randla_pyramid_workspace_bytes(int64_t batch, int64_t n0, int num_layers,
                                           const int32_t* ratios_host);

int ml3d_randla_knn_pyramid(const float* points, int64_t batch, int64_t n0, int num_layers,
                            const int32_t* ratios_host, int k,
                            int32_t* const* neighbor_idx_host, int32_t* const* interp_idx_host,
                            void* workspace, size_t workspace_bytes, void* stream);


/* ------------------------------------------------------------------------- */
/* RandLA-Net inference forward (eval mode, BatchNorm folded by the caller)   */
/* replaces the PyTorch op chain of RandLANet.forward                          */
/*   ml3d/torch/models/randlanet.py:241-298 — fc0/bn0 (:266-271), per layer    */
/*   LocalFeatureAggregation.forward (:667-692) = mlp1, LocalSpatialEncoding   */
/*   (:533-605), AttentivePooling (:622-639) x2, mlp2 + shortcut, then         */
/*   random_sample (:300-327); mlp (:285); decoder nearest_interpolation       */
/*   (:329-350) + cat + ConvTranspose2d (:287-293); fc1 (:296).                */
/*                                                                             */
/* Every Linear/Conv1x1 is y = act(W^T-packed x + b) with BatchNorm (eval)     */
/* folded into W, b by the host.  Packed weights are [C_in][C_out] row-major   */
/* ("WT").  `params` is ONE device buffer; slot offsets (in floats) come from  */
/* ml3d_randla_param_layout, in this fixed slot order:                         */
/*   0 fc0_WT[in][F] 1 fc0_b[F]                                                */
/*   per encoder layer l (base 2 + 18*l), d_in -> d=dim_output[l], h=d/2:      */
/*     +0 mlp1_WT[d_in][h] +1 mlp1_b  +2 lse1_WT[10][h] +3 lse1_b              */
/*     +4 pool1_score_WT[d][d] +5 pool1_score_b +6 pool1_mlp_WT[d][h] +7 b     */
/*     +8 lse2_WT[h][h] +9 lse2_b +10 pool2_score_WT[d][d] +11 pool2_score_b   */
/*     +12 pool2_mlp_WT[d][d] +13 b +14 mlp2_WT[d][2d] +15 shortcut_WT[d_in][2d]*/
/*     (adjacent: one stacked [d + d_in][2d] matrix) +16 mlp2_b +17 shortcut_b */
/*   then mlp_WT[D][D], mlp_b (D = 2*dim_output[L-1]);                         */
/*   then per decoder stage i: dec_WT[C_in_i][C_out_i], dec_b;                 */
/*   then fc1_0_WT[C][64], b, fc1_1_WT[64][32], b, fc1_3_WT[32][classes], b.   */
/* ------------------------------------------------------------------------- */
#define ML3D_RANDLA_MAX_LAYERS 8

typedef struct ml3d_randla_desc {
    int32_t num_layers;
    int32_t in_channels;   /* 3 + feature dims (randlanet.py:208-211) */
    int32_t dim_features;  /* fc0 width */
    int32_t num_classes;
    int32_t num_neighbors; /* must be 16 (every reference config) */
    int32_t dim_output[ML3D_RANDLA_MAX_LAYERS];
    int32_t sub_sampling_ratio[ML3D_RANDLA_MAX_LAYERS];
    int64_t batch;
    int64_t num_points;    /* n0 per cloud */
} ml3d_randla_desc;

/* number of parameter slots; offsets_out[i] = start of slot i in floats,       */
/* offsets_out[n_slots] = total floats.  Returns n_slots or <0.                 */
int ml3d_randla_param_layout(const ml3d_randla_desc* desc_host, int64_t* offsets_out, int max_slots);

size_t ml3d_randla_forward_workspace_bytes(const ml3d_randla_desc* desc_host);

/* features [batch, n0, in_channels] f32; points [batch, n0, 3] f32;            */
/* neighbor_idx[l] [batch, n_l, 16] i32 and interp_idx[l] [batch, n_l, 1] i32   */
/* as produced by ml3d_randla_knn_pyramid (item-local indices);                 */
/* out_scores [batch, n0, num_classes] f32 (the tensor RandLANet.forward        */
/* returns, randlanet.py:298).                                                  */
int ml3d_randla_forward(const ml3d_randla_desc* desc_host, const float* params,
                        const float* features, const float* points,
                        const int32_t* const* neighbor_idx_host,
                        const int32_t* const* interp_idx_host, float* out_scores,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Same as ml3d_randla_forward, plus an optional measurement hook: when trace_host is    */
/* non-NULL the library records ev_start / ev_stop (hipEvent_t, created by the caller)   */
/* on `stream` immediately before / after the ONE kernel launch whose tag matches, so a  */
/* harness can time a single kernel inside a full forward without a profiler.            */
/* tag = 8*layer + {0 mlp1, 1 lfa_stage1, 2 lfa_stage2, 3 gather_max};                   */
/*       1000 fc0, 1001 mlp, 1100+i decoder stage i, 1200/1201/1202 fc1 layers.          */
/* `next` chains further records (NULL ends the list): every record whose tag matches is */
/* served.  Besides timing, a recorded event is how another stream synchronises with a   */
/* POINT INSIDE the forward (ml3d.engine.PipelinedRandLAEngine starts the next batch's   */
/* neighbour search once the bandwidth-bound first layer of the running forward is done).*/
typedef struct ml3d_trace {
    int32_t tag;
    void* ev_start;
    void* ev_stop;
    const struct ml3d_trace* next;
} ml3d_trace;

int ml3d_randla_forward_traced(const ml3d_randla_desc* desc_host, const float* params,
                               const float* features, const float* points,
                               const int32_t* const* neighbor_idx_host,
                               const int32_t* const* interp_idx_host, float* out_scores,
                               void* workspace, size_t workspace_bytes, void* stream,
                               const ml3d_trace* trace_host);

/* ml3d_randla_knn_pyramid with the same measurement hook: tag = 2*l (k-NN query kernel of  */
/* level l), 2*l+1 (1-NN interpolation query of level l), 100+l (whole grid build of level l). */
int ml3d_randla_knn_pyramid_traced(const float* points, int64_t batch, int64_t n0, int num_layers,
                                   const int32_t* ratios_host, int k,
                                   int32_t* const* neighbor_idx_host, int32_t* const* interp_idx_host,
                                   void* workspace, size_t workspace_bytes, void* stream,
                                   const ml3d_trace* trace_host);

/* Optional spatial tile order for the forward's attention kernels.                                  */
/* ml3d_randla_knn_pyramid_ordered additionally writes, for every level l < num_layers whose entry of  */
/* tile_order_host is non-NULL, tile_order[l] [batch * n_l] i32 = the point rows of level l in the     */
/* cell-sorted order of that level's search grid (cloud-major; a permutation of 0 .. batch*n_l - 1).   */
/* ml3d_randla_forward_ordered walks each level's attention tiles in that order (entries may be NULL): */
/* same outputs -- the per-point arithmetic does not depend on which points share a tile -- but        */
/* consecutive tiles gather neighbouring rows (RandLA point order inside a cloud is random,            */
/* randlanet.py:218-229, so the natural row order has no locality).                                    */
int ml3d_randla_knn_pyramid_ordered(const float* points, int64_t batch, int64_t n0, int num_layers,
                                    const int32_t* ratios_host, int k,
                                    int32_t* const* neighbor_idx_host, int32_t* const* interp_idx_host,
                                    int32_t* const* tile_order_host,
                                    void* workspace, size_t workspace_bytes, void* stream,
                                    const ml3d_trace* trace_host);

int ml3d_randla_forward_ordered(const ml3d_randla_desc* desc_host, const float* params,
                                const float* features, const float* points,
                                const int32_t* const* neighbor_idx_host,
                                const int32_t* const* interp_idx_host,
                                const int32_t* const* tile_order_host, float* out_scores,
                                void* workspace, size_t workspace_bytes, void* stream,
                                const ml3d_trace* trace_host);

/* ------------------------------------------------------------------------------------------------------------------------------ */
/* Training side on hand-written HIP, forward AND backward (ABI 10; open3d-ml_amd/csrc/train.hip).  What the       */
/* reference leaves to torch.autograd around loss.backward() (ml3d/torch/pipelines/semantic_segmentation.py:412-437) for the      */
/* modules of the two segmentation models: every Linear / 1x1 convolution (SharedMLP randlanet.py:503-518, UnaryBlock             */
/* kpconv.py:1288-1293), BatchNorm on the batch statistics (+ LeakyReLU), the index gathers / pools and the attentive pooling     */
/* stage.  ml3d.ops.{LinearFunction, BatchNormActFunction, GatherRowsFunction, GatherPoolFunction, AttentionStageFunction} bind   */
/* them into autograd graphs; outputs are caller-owned, reductions across workgroups are float atomics.                           */
/* ------------------------------------------------------------------------------------------------------------------------------ */
/* ml3d_gemm_tn: c[i, j] = sum_r a[r, i] * b[r, j]  (a [m, lda] k columns, b [m, ldb] n columns, c [k, ldc]; c is zeroed here, the    */
/*   row slices meet in it through float atomics).                                                                                 */
/*   Weight gradient of y = x W^T: grad_W [out, in] = gemm_tn(a = grad_y, b = x); of KPConv's contraction: grad_W [15 cin, cout]  */
/*   = gemm_tn(a = wf, b = grad_out) (kpconv.py:1139-1159).  col_sums_a (optional, [k]) <- sum_r a[r, :] (the bias gradient).     */
int ml3d_gemm_tn(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m, int k, int n,
                 float* c, int64_t ldc, float* col_sums_a, void* stream);

/* ml3d_batchnorm_train_forward: y = act(gamma * (x - mean) * invstd + beta) with mean / biased variance of THIS batch over the    */
/*   rows of x [rows, channels] (torch.nn.functional.batch_norm(training=True); BatchNorm2d over [B, C, N, K] = rows B*N*K).       */
/*   act: 0 none, 1 LeakyReLU(slope).  save_mean / save_var (biased) / save_invstd [channels] are returned for the backward and    */
/*   for the caller's running-statistics update.  gamma / beta may be NULL (1 / 0).                                                */
/* ml3d_batchnorm_train_backward: grad_x, grad_gamma, grad_beta from (x, y, grad_y, saved statistics).                             */
size_t ml3d_batchnorm_train_workspace_bytes(int channels);

int ml3d_batchnorm_train_forward(const float* x, int64_t rows, int channels, const float* gamma,
                                 const float* beta, float eps, int act, float slope, float* y,
                                 float* save_mean, float* save_var, float* save_invstd,
                                 void* workspace, size_t workspace_bytes, void* stream);

int ml3d_batchnorm_train_backward(const float* x, const float* y, const float* grad_y, int64_t rows,
                                  int channels, const float* gamma, const float* save_mean,
                                  const float* save_invstd, int act, float slope, float* grad_x,
                                  float* grad_gamma, float* grad_beta, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* ml3d_gather_rows: out[r, :] = x[index[r * index_stride], :] (rows outside [0, n_src) read zeros: the shadow neighbour);         */
/*   nearest_interpolation (randlanet.py:329-350), closest_pool (kpconv.py:821-838).  ml3d_scatter_add_rows is its adjoint         */
/*   (grad_x [n_src, channels] zeroed here).  ml3d_gather_pool_backward: adjoint of ml3d_gather_pool (mode 0 max_pool: the        */
/*   gradient goes to the first maximal neighbour in list order, dropped when that is the shadow row; mode 1 closest_pool).        */
int ml3d_gather_rows(const float* x, int64_t n_src, int channels, const int32_t* index,
                     int64_t index_stride, int64_t m, float* out, void* stream);

int ml3d_scatter_add_rows(const float* grad_out, int64_t n_src, int channels, const int32_t* index,
                          int64_t index_stride, int64_t m, float* grad_x, void* stream);

int ml3d_gather_pool_backward(const float* features, int64_t n_supports, int channels,
                              const int32_t* inds, int64_t n_queries, int64_t max_neighbors, int mode,
                              const float* grad_out, float* grad_features, void* stream);

/* ml3d_kpconv_deformed_weighted: wf[q, k, c] = sum_h max(0, 1 - |s[inds[q, h]] - q - dkp[q, k]| / extent) * features[inds[q, h], c]    */
/*   -- the aggregation of a DEFORMABLE KPConv in training (kpconv.py:1011-1066, 1105-1137; KP_influence linear, sum aggregation)      */
/*   with the query's own kernel points deformed_kernel_points [n_queries, 15, 3] (= kernel_points + extent * offsets, the caller's    */
/*   arithmetic); out_wf [n_queries, 15 * cin].  The backward returns grad_features [n_supports, cin] (zeroed here, atomics) and       */
/*   grad_kernel_points [n_queries, 15, 3] (written): the gradient that reaches the offset convolution (ml3d.ops.KPConvDeformedFunction). */
int ml3d_kpconv_deformed_weighted(const float* q_pts, const float* s_pts, const int32_t* neighb_inds,
                                  int64_t n_queries, int64_t n_supports, int64_t max_neighbors,
                                  const float* features, int cin, const float* deformed_kernel_points,
                                  int num_kernel_points, float kp_extent, float* out_wf, void* stream);

int ml3d_kpconv_deformed_weighted_backward(const float* q_pts, const float* s_pts, const int32_t* neighb_inds,
                                           int64_t n_queries, int64_t n_supports, int64_t max_neighbors,
                                           const float* features, int cin, const float* deformed_kernel_points,
                                           int num_kernel_points, float kp_extent, const float* grad_wf,
                                           float* grad_features, float* grad_kernel_points, void* stream);

/* ml3d_kpconv_offset_regulariser (ABI 12): p2p_fitting_regularizer of the deformable blocks (kpconv.py:2167-2206) with the min_d2 of */
/*   kpconv.py:1058-1074, value and gradient in one pass without the [Nq, H, K] distance tensor.  Per (query, kernel point):          */
/*   fitting = min_h |s[inds[q, h]] - q - dkp[q, k]|^2 / extent^2 (shadow neighbours at 1e6), repulsive = sum_{j != k} min(|l_j - l_k|  */
/*   - repulse_extent, 0)^2 with l = dkp / extent and the other points detached.  out_partial_sums double[2 * blocks] (blocks =        */
/*   ml3d_kpconv_offset_regulariser_blocks(n_queries)): per-workgroup (sum fitting, sum repulsive) -- add them and divide by            */
/*   n_queries * K for the two L1 means; optional out_min_d2 [Nq, K] (unnormalised), out_grad_fitting / _repulsive [Nq, K, 3] =        */
/*   d fitting[q, k] / d dkp[q, k], d repulsive[q, k] / d dkp[q, k].  At most 16 kernel points.                                          */
int64_t ml3d_kpconv_offset_regulariser_blocks(int64_t n_queries);
int ml3d_kpconv_offset_regulariser(const float* q_pts, const float* s_pts, const int32_t* neighb_inds, int64_t n_queries,
                                   int64_t n_supports, int64_t max_neighbors, const float* deformed_kernel_points,
                                   int num_kernel_points, float kp_extent, float repulse_extent, float* out_min_d2,
                                   float* out_grad_fitting, float* out_grad_repulsive, double* out_partial_sums, void* stream);

/* ml3d_randla_attention_stage: one attentive pooling of LocalFeatureAggregation in training form, fused (randlanet.py:596-605,     */
/*   617, 631-637): x[p, k, :] = [ f[b, idx[p, k], :c1] | enc[p, k, :c2] ], s = x W^T + bias, out[p, c] = sum_k softmax_k(s)[k, c]   */
/*   x[p, k, c].  f [batch, n, c1], enc [batch, n, k, c2], neighbor_idx int32 [batch, n, k] (item-local), weight [d, d] (the         */
/*   Linear's [out, in]) and weight_t its transpose, d = c1 + c2; out [batch, n, d].  k must be 16, d even and <= 256               */
/*   and, above d = 128, c1 <= 160 (ML3D_E_UNSUPPORTED otherwise: the caller keeps the unfused ops).  The backward recomputes x, s   */
/*   and the softmax and returns grad_f [batch, n, c1] (zeroed here; the scatter through neighbor_idx uses float atomics),           */
/*   grad_enc [batch, n, k, c2] (written), grad_weight [d, d] (the persistent workgroups' private partial sums in `workspace`,      */
/*   added up by a second kernel: deterministic) and grad_bias [d]: no [batch, n, k, d] tensor exists in either pass.                */
int ml3d_randla_attention_stage(const float* f, const float* enc, const int32_t* neighbor_idx,
                                const float* weight_t, const float* bias, int64_t batch, int64_t n,
                                int k, int c1, int c2, float* out, void* stream);

size_t ml3d_randla_attention_stage_backward_workspace_bytes(int64_t batch, int64_t n, int c1, int c2);

int ml3d_randla_attention_stage_backward(const float* f, const float* enc, const int32_t* neighbor_idx,
                                         const float* weight, const float* weight_t, const float* bias,
                                         const float* out, const float* grad_out, int64_t batch,
                                         int64_t n, int k, int c1, int c2, float* grad_f,
                                         float* grad_enc, float* grad_weight, float* grad_bias,
                                         void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ML3D_HIP_H */
