/* coda_gemm.h -- fp32 library GEMMs (hipBLASLt) behind the same C ABI as the kernels.
 *
 * What it replaces: the `F.linear` / `torch.mm` calls of the reference's attention
 * projections and feed-forward layers (models/transformer.py:461-479, 556-580;
 * nn.MultiheadAttention in_proj / out_proj) as the host-side mirrors issue them.  These are
 * PLAIN GEMMs: the arithmetic stays in the vendor library (hipBLASLt, the library PyTorch-ROCm
 * itself calls).  The entry point exists because PyTorch's wrapper re-creates the matmul
 * descriptors and re-runs the algorithm heuristic on every call (~22 us of host time per GEMM,
 * ~225 launch-sized GEMMs per step); here descriptor, layouts and the chosen algorithm are
 * cached per problem shape.
 *
 * Row-major convention: C (m x n, row stride ldc) = op(A) op(B) [+ bias] [+ C], with
 * op(A) m x k and op(B) k x n.  transa != 0: A is stored k x m (row stride lda) and used
 * transposed; likewise transb (B stored n x k).  bias: optional n floats added to every row.
 * accumulate != 0: the product is added to the existing C (beta = 1) instead of overwriting it.
 *
 * Return values and stream semantics as in coda_pointnet2.h.  A problem hipBLASLt cannot PLAN (no
 * heuristic result, unsupported leading dimension) is reported as -(3000 + hipblasStatus_t) -- the
 * verdict is cached per shape, descriptors released -- and a failure of the matmul call itself as
 * -(2000 + hipblasStatus_t).  State (handle, plans, one workspace per stream: 32 MiB, grown up to 256 MiB if an algorithm asks for more) is kept per device.
 */
#ifndef CODA_GEMM_H
#define CODA_GEMM_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

int coda_gemm_f32(int transa, int transb, int m, int n, int k, const float *a,
                  long long lda, const float *b, long long ldb, float *c, long long ldc,
                  const float *bias, int accumulate, void *stream);

/* The general form behind coda_gemm_f32 (same library, same plan cache):
 *   C = epilogue(alpha * op(A) op(B) + beta * C [+ bias])
 * dtype: CODA_DTYPE_F32, or CODA_DTYPE_F16 (A, B, C are IEEE half, products accumulated in fp32: the frozen CLIP
 * image tower, include/coda_clip_tower.h).  `bias` is fp32 for either dtype.  epilogue: 0 none (bias must be
 * NULL), 1 "+ bias", 2 "swish(. + bias)" with swish(x) = x * sigmoid(x). */
#define CODA_DTYPE_F32 0
#define CODA_DTYPE_F16 1
int coda_gemm_ex(int dtype, int epilogue, int transa, int transb, int m, int n, int k, const void *a,
                 long long lda, const void *b, long long ldb, void *c, long long ldc, const float *bias,
                 float alpha, float beta, void *stream);

/* First-use timing of the library's candidate algorithms for a shape (the heuristic's first answer is not always
 * the fastest): the candidates run on the caller's operands into a scratch output and the fastest is kept for the
 * process.  On for CODA_DTYPE_F16, off for CODA_DTYPE_F32 (a process then always runs the same fp32 kernels); the
 * environment variable CODA_GEMM_TUNE=0|1, read once, forces it for both.  (No setter: the library keeps no mutable
 * process-wide state.) */

/* Own fp32-MFMA kernel for the same product (csrc/gemm_nn.hip), used for the launch-sized problems of the
 * transformer stacks where the library costs ~14 us of host time per call:
 *   transb != 0:  C (m x n) [+]= A (m x k) . B^T + bias,  B (n x k)     (y = x W^T + b)
 *   transb == 0:  C (m x n) [+]= A (m x k) . B,           B (k x n)     (dx = dy W)
 * Constraints: m, n multiples of 64, k a multiple of 32, lda / ldb multiples of 4 floats, a and b 16-B
 * aligned; any other problem returns CODA_ENOSPC ("not this kernel's shape": use coda_gemm_f32). */
int coda_sgemm_f32(int transb, int m, int n, int k, const float *a, long long lda, const float *b,
                   long long ldb, float *c, long long ldc, const float *bias, int accumulate, void *stream);
/* The same product with the transformer feed-forward's element-wise pass in the epilogue:
 *   C = dropout(relu(A . op(B) + bias)),  keep decisions = the counter hash of (seed, row * n + col) that
 *   coda_tok_bias_relu_dropout_fwd_f32 uses, so C equals that call on the plain product.
 * Launch-sized problems only (m * n <= 2048 * 1024, m, n multiples of 64, k a multiple of 128, ldc == n);
 * CODA_ENOSPC otherwise: run the two passes. */
int coda_sgemm_relu_dropout_f32(int transb, int m, int n, int k, const float *a, long long lda, const float *b,
                                long long ldb, float *c, long long ldc, const float *bias, float dropout_p,
                                uint64_t seed, void *stream);

/* The backward of that layer's activation in the epilogue of the product that feeds it (round 6): dz (m x n, dense) =
 * relu-dropout backward of da (m x k) . w (k x n) -- dz = product / (1 - p) where the saved activation act (m x n,
 * dense; zero where clamped or dropped) is positive, else 0 -- and partials (m / 32, n) = the column sums of dz over
 * each 32-row tile (the bias gradient: reduce the coda_sgemm_relu_dropout_bwd_blocks(m) rows in fixed order, e.g.
 * with coda_tok_colsum_finalize_grouped_f32).  One launch where the feed-forward's backward ran a product and
 * coda_tok_bias_relu_dropout_bwd_f32.  Launch-sized problems only (as above); CODA_ENOSPC otherwise. */
int coda_sgemm_relu_dropout_bwd_blocks(int m);
int coda_sgemm_relu_dropout_bwd_f32(int m, int n, int k, const float *da, long long ldda, const float *w,
                                    long long ldw, const float *act, float dropout_p, float *dz,
                                    float *partials, void *stream);

/* Many weight gradients in one launch: out_p (m x n, row stride ldout) = dy_p^T x_p for every problem of the
 * array (dy_p rows x m, x_p rows x n; replaces one `torch.mm(dy.t(), x)` per linear layer, models/transformer.py's
 * decoder layers as autograd differentiates them).  `problems` is HOST memory, read during the call (the
 * descriptors travel as kernel arguments, 64 problems per launch).  Deterministic (fixed summation order), out
 * is overwritten.  Constraints: m, n multiples of 64, rows a multiple of 8; CODA_ENOSPC otherwise. */
typedef struct CodaTnProblem {
  const float *dy;
  const float *x;
  float *out;
  int rows, m, n;
  long long lddy, ldx, ldout;
} CodaTnProblem;

int coda_grouped_gemm_tn_f32(const CodaTnProblem *problems, int count, void *stream);

/* ---- fp32-accurate GEMMs on the bf16 matrix cores (csrc/gemm_x3.hip) ------------------------------------------
 * Same products as coda_gemm_f32 for the token-wise linear layers (models/helpers.py:45-112,
 * models/transformer.py:461-479, 556-580, models/model_3detr.py:409-419, 475-511), computed with every fp32 operand
 * carried as three bf16 pieces (x = hi + mid + lo exactly) and the six piece products of order <= 2 accumulated in
 * fp32: error against float64 within 2x of the native fp32 GEMM (tests/test_gemm_x3_gpu.py), 6/16 of its matrix time.
 *
 * The WEIGHT operand is split ahead of time, once per optimizer step, into TILED plane sets: a matrix of R output
 * columns x C contraction indices is stored as blocks [ceil(R / 128)][C / 32] of 3 planes x 128 rows x 32 k bf16 (24 KB,
 * contiguous: a stage's weight operand is one sequential run of whole cache lines); rows past R stay as the caller zeroed
 * them.  Size: 3 * ceil(R / 128) * 128 * C bf16.
 *   coda_gemm_x3_split_f32: for every item, src (rows x cols fp32, row stride ld) -> nt = the tiled set of src itself
 *   (R = rows, C = cols; needs cols % 32 == 0) and / or nn = the tiled set of its TRANSPOSE (R = cols, C = rows; needs
 *   rows % 32 == 0); either may be NULL.  `items` is HOST memory read during the call; 48 items per launch.
 * The product:
 *   coda_gemm_x3_nt_f32: C (m x n, row stride ldc) [+]= A (m x k fp32, row stride lda) . W^T [+ bias], W = rows
 *   w_row0 .. w_row0 + n - 1, columns w_col0 .. w_col0 + k - 1 of the matrix whose tiled set `w_tiled` is (w_cols = that
 *   matrix's column count).  y = x W^T + b takes the `nt` set of W (n x k); dx = dy W takes the `nn` set (rows = dx's
 *   columns, contraction over W's rows) -- a row slice [r0, r0 + n') of a packed weight is w_row0 = r0 in nt and
 *   w_col0 = r0, k = n' in nn.
 *   Constraints: m a multiple of 128, n of 64, k and w_col0 of 32, w_row0 of 64 (of 128 for the 128-wide tiles), lda
 *   a multiple of 4, a and w_tiled 16-byte aligned; CODA_ENOSPC otherwise ("not this kernel's shape": use coda_gemm_f32). */
typedef struct CodaX3SplitItem {
  const float *src;
  void *nt;
  void *nn;
  int rows, cols;
  long long ld;
} CodaX3SplitItem;
int coda_gemm_x3_split_f32(const CodaX3SplitItem *items, int count, void *stream);
int coda_gemm_x3_nt_f32(int m, int n, int k, const float *a, long long lda, const void *w_tiled, int w_cols, int w_row0,
                        int w_col0, float *c, long long ldc, const float *bias, int accumulate, void *stream);
/* Weight gradients: part (slices x m x n, dense) [s] = dY_s^T X_s over the s-th of `slices` equal token ranges of
 * dY (rows x m, row stride lddy) and X (rows x n, row stride ldx) -- `torch.mm(dy.t(), x)` of a linear layer over its
 * 16 384 token rows as the library path computes it (row chunks as a batched GEMM, then a sum over the chunks), both
 * operands split in the kernel.  The sum over the slices is the caller's (coda_tok_colsum_finalize_grouped_f32).
 * Constraints: m, n multiples of 128, rows / slices a multiple of 32, lddy / ldx multiples of 4, 16-byte aligned
 * operands; CODA_ENOSPC otherwise. */
int coda_gemm_x3_tn_f32(int rows, int m, int n, const float *dy, long long lddy, const float *x, long long ldx,
                        float *part, int slices, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_GEMM_H */
