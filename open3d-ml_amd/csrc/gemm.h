// gemm.h — f32 GEMM on v_mfma_f32_32x32x2_f32 with fused epilogue (gfx950), shared by the KPConv and
// PointPillars forward passes.
//
//   C[m, n] = act( sum_k A(m, k) * B[k, n] + bias[n] + residual[m, n] )
//
// A is produced by a LOADER functor, so the same tile loop serves
//   * dense rows  [M, K1]           (UnaryBlock Linear, kpconv.py:1288-1293; 1x1 convs; KPConv's
//                                     [Nq, 15*Cin] x [15*Cin, Cout] contraction, kpconv.py:1147-1159)
//   * gathered + concatenated rows  ([x[up_idx[m, 0]] | skip[m]]: NearestUpsampleBlock + cat + unary of
//                                     the KPFCNN decoder, kpconv.py:283-286, 821-838)
//   * implicit im2col of an NHWC image (3x3 / strided convs of SECOND, point_pillars.py:619-682).
// f32-input MFMA keeps the reference's float32 arithmetic (tolerance 1e-4 on logits); roofline = the
// f32 matrix peak 157.3 TFLOP/s.  A workgroup (256 threads = 4 waves) owns a 64 x 64 tile of C, K is
// walked in chunks of 32 staged through LDS with register prefetch of the next chunk; small-M / deep-K
// problems are split along K across gridDim.z into a caller-provided partial buffer and reduced by a
// second kernel (deterministic — no float atomics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ml3d {

struct Epilogue {
    const float* bias;        // [N] or null
    const float* residual;    // [M, ldr] or null (added before the activation)
    int64_t ldr;
    int act;                  // 0 none, 1 leaky relu (slope), 2 relu
    float slope;
    // pixel-shuffle store (ConvTranspose2d with kernel == stride, point_pillars.py:712-717): row m = input pixel
    // (b, y, x) of a [B, ps_h, ps_w] map, column n = (dy, dx, co); the value goes to output pixel
    // (b, y*ps + dy, x*ps + dx), channel co, of an NHWC map whose pixel stride is ldc.  bias is indexed by co.
    int ps, ps_h, ps_w, ps_cout;
    const float* bias2;       // [N] or null: a second bias (two Linears summed into one GEMM over concatenated inputs)
    // gathered residual: when set, the residual row of output row m is
    //   (m / rg_rows_per_item) * rg_src_rows_per_item + res_gather[m]      (rg_rows_per_item >= 64)
    // -- the per-COARSE-point half of "nearest upsample + concat + Linear" added back through the upsampling index
    const int32_t* res_gather;
    int64_t rg_rows_per_item, rg_src_rows_per_item;
    // rg_stride: element stride of res_gather (0 = 1; KPConv: the first column of an [M, H] neighbour matrix);
    // rg_limit: gathered rows outside [0, rg_limit) contribute nothing (the shadow neighbour of a point with no coarse point
    // in reach); must be set whenever res_gather is
    int64_t rg_stride, rg_limit;
};

// dense / gathered / concatenated rows
struct RowsA {
    const float* a; int64_t lda; int k1;
    const int32_t* gather; int64_t gather_stride; int64_t a_rows;   // optional: row = gather[m * stride]; >= a_rows -> zeros
    const float* a2; int64_t lda2; int k2;                           // optional second block of columns
    // gather variants: gather_on_a2 != 0 applies the row gather to the SECOND block instead of the first;
    // g_rows_per_item > 0 makes the gathered index item-local: source row =
    // (m / g_rows_per_item) * g_src_rows_per_item + gather[m * stride]   (RandLA nearest_interpolation)
    int gather_on_a2;
    int64_t g_rows_per_item, g_src_rows_per_item;
};

// implicit im2col of an NHWC image: m = (b, oy, ox), k = (ky, kx, ci), ci fastest
struct ConvA {
    const float* in;     // [B, H, W, C]
    int B, H, W, C;
    int OH, OW;
    int KH, KW, stride, pad;
};

size_t gemm_partial_bytes(int64_t M, int N, int K);   // workspace for the split-K partials (may be 0)

int gemm_rows(const RowsA& A, const float* Bm, int64_t M, int N, int K, const Epilogue& ep, float* C, int64_t ldc,
              void* partial_ws, size_t partial_bytes, hipStream_t stream);
int gemm_conv(const ConvA& A, const float* Bm, int N, const Epilogue& ep, float* C, int64_t ldc, void* partial_ws,
              size_t partial_bytes, hipStream_t stream);

// ---- the same convolution on the bf16 matrix pipe (gemm.hip, gemm_tile_bf3): float32-equivalent products from three-way bf16
// splits of both operands.  The weights are split once by gemm_pack_bf16x3 ([K, N] float, K % 32 == 0 -> gemm_pack_bf16x3_bytes
// bytes, 16-byte aligned); gemm_conv_bf16x3_ok tells whether a problem is eligible (cin % 32 == 0, 32-bit offsets).
size_t gemm_pack_bf16x3_bytes(int K, int N);
int gemm_pack_bf16x3(const float* Bm, int K, int N, void* packed, hipStream_t stream);
bool gemm_conv_bf16x3_ok(const ConvA& A);
// dense rows [a[M, k1] | a2[M, k2]] (each block float4-addressable, k1 and k1 + k2 multiples of 32; a gathered residual needs
// rg_rows_per_item >= 128; else ML3D_E_UNSUPPORTED): Linears, KPConv's contraction, kernel == stride deconvolutions.  Small-M / deep-K problems are split along K into partial_ws (gemm_partial_bytes_bf16x3 bytes; without it
// the problem runs unsplit)
size_t gemm_partial_bytes_bf16x3(int64_t M, int N, int K);
int gemm_rows_bf16x3(const float* a, int64_t lda, int k1, const float* a2, int64_t lda2, int k2, int64_t M, const void* packed,
                     int N, const Epilogue& ep, float* C, int64_t ldc, void* partial_ws, size_t partial_bytes, hipStream_t stream);
int gemm_conv_bf16x3(const ConvA& A, const void* packed, int N, const Epilogue& ep, float* C, int64_t ldc, hipStream_t stream);

}  // namespace ml3d
