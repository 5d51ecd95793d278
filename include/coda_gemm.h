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
 * Return values and stream semantics as in coda_pointnet2.h; a hipBLASLt failure is reported as
 * -(2000 + hipblasStatus_t).  The library keeps one 32 MiB workspace per stream it is called on.
 */
#ifndef CODA_GEMM_H
#define CODA_GEMM_H

#ifdef __cplusplus
extern "C" {
#endif

int coda_gemm_f32(int transa, int transb, int m, int n, int k, const float *a,
                  long long lda, const float *b, long long ldb, float *c, long long ldc,
                  const float *bias, int accumulate, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_GEMM_H */
