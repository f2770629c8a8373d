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
 *  - no global state: re-entrant, one stream per call.
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
/* device pointers, length num_layers.                                         */
/* ------------------------------------------------------------------------- */
size_t ml3d_randla_pyramid_workspace_bytes(int64_t batch, int64_t n0, int num_layers,
                                           const int32_t* ratios_host);

int ml3d_randla_knn_pyramid(const float* points, int64_t batch, int64_t n0, int num_layers,
                            const int32_t* ratios_host, int k,
                            int32_t* const* neighbor_idx_host, int32_t* const* interp_idx_host,
                            void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ML3D_HIP_H */
