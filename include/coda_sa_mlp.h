/*
 * coda_sa_mlp.h -- C ABI of the streaming kernels of the set-abstraction shared MLP.
 *
 * Together with plain library GEMMs they replace the per-layer
 * Conv2d(1x1) + BatchNorm2d + ReLU stack and the max-pool over nsample of
 * PointnetSAModuleVotes (pointnet2_modules.py:247-253, pytorch_utils.py:8-117) on
 * CHANNELS-LAST activations: a matrix of P = B*npoint*nsample rows and C channels,
 * row p = (b, centre, sample).  C must be a multiple of 4 that divides 1024.
 *
 * Conventions as in coda_pointnet2.h (raw device pointers, stream, status codes).
 * `w1` selects the first layer: when non-NULL, `src` is the grouped xyz (P,3) and the
 * pre-BN activation is recomputed on the fly as x . w1[c] (w1 is (C,3) row-major);
 * when NULL, `src` is the stored pre-BN activation (P,C).
 * Statistics (`sums`, 2*C doubles: per-channel sum, then sum of squares / or sum d,
 * then sum d*xhat in the backward) are zeroed by the call.
 *
 * De-duplicated groups.  ball_query pads a group that has fewer than nsample points in its
 * radius with copies of its first hit (ball_query_gpu.cu:35-48), and every layer of the MLP
 * maps identical rows to identical rows, so the caller may store each group as its DISTINCT
 * rows only (variable length, `group_offsets` (G+1) row offsets) with a per-row multiplicity
 * `row_weight` (1, or 1 + number of copies for the first row of a group): statistics count a
 * row `row_weight` times, pooling is unaffected, and in the backward a row carries the SUM of
 * the gradients of the rows it stands for.  Both pointers NULL = dense groups of s_len rows.
 */
#ifndef CODA_SA_MLP_H
#define CODA_SA_MLP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sums[0:C] = sum_p y, sums[C:2C] = sum_p y^2 */
int coda_sa_col_stats_f32(const float *src, const float *w1, long long p, int c,
                          const float *row_weight, double *sums, void *stream);
/* dst = relu(y * scale + shift) */
int coda_sa_bn_relu_apply_f32(const float *src, const float *w1, const float *scale,
                              const float *shift, long long p, int c, float *dst,
                              void *stream);
/* last layer: statistics + per-(group, channel) max / min over the s_len rows of a group
 * and the row index (0..s_len-1) where they occur (lowest on ties) */
int coda_sa_col_stats_pool_f32(const float *y, long long groups, int s_len, int c,
                               const float *row_weight, const int32_t *group_offsets,
                               double *sums, float *ymax, float *ymin,
                               int32_t *amax, int32_t *amin, void *stream);
/* backward of max-pool + ReLU + BN of the last layer; coef = [a, m1, m2, mean, invstd][C]:
 * dy[p][c] = a * ((row == sel ? d : 0) - w_p * (m1 + (y - mean) * invstd * m2)) */
int coda_sa_bn_bwd_sparse_f32(const float *y, const float *d, const int32_t *sel,
                              const float *coef, long long groups, int s_len, int c,
                              const float *row_weight, const int32_t *group_offsets,
                              float *dy, void *stream);
/* hidden layers, prm = [scale, shift, mean, invstd][C]; d = da where scale*y+shift > 0:
 * sums[0:C] = sum d, sums[C:2C] = sum d * (y - mean) * invstd */
int coda_sa_relu_bn_bwd_stats_f32(const float *da, const float *src, const float *w1,
                                  const float *prm, long long p, int c, double *sums,
                                  void *stream);
/* prm = [scale, shift, mean, invstd, a, m1, m2][C]; dy = a * (d - w_p * (m1 + xhat * m2)).
 * w1 == NULL: dy (P,C) is written (may alias da).  w1 != NULL (first layer): dy is not
 * stored, dw1[k*C + c] = sum_p dy[p][c] * x[p][k] (3*C doubles, zeroed by the call). */
int coda_sa_relu_bn_bwd_apply_f32(const float *da, const float *src, const float *w1,
                                  const float *prm, long long p, int c,
                                  const float *row_weight, float *dy, double *dw1,
                                  void *stream);

/* Batch-norm bookkeeping between the streaming kernels (pytorch_utils.py:8-117's BatchNorm2d in train mode):
 * stats = [scale, shift, mean, invstd][C] from sums = [sum y, sum y^2] over n rows; running statistics updated
 * with `momentum` (running_mean / running_var / num_batches may be NULL: no update). */
int coda_sa_bn_finalize_f32(const double *sums, double n, double eps, float momentum, const float *gamma,
                            const float *beta, float *running_mean, float *running_var, long long *num_batches,
                            float *stats, int c, void *stream);
/* sums = [sum d, sum d * xhat]: dbeta / dgamma = float(sums) (NULL: skip); coef (NULL: skip) in layout 0 =
 * coda_sa_relu_bn_bwd_apply_f32's prm[7][C], layout 1 = coda_sa_bn_bwd_sparse_f32's coef[5][C]; n <= 0: m1 = m2 = 0. */
int coda_sa_bn_bwd_coef_f32(const double *sums, double n, const float *gamma, const float *stats, float *coef,
                            int layout, float *dbeta, float *dgamma, int c, void *stream);
/* pooled last layer, forward: ysel / sel = the group's max (scale >= 0) or min pre-BN value and its row,
 * out = relu(ysel * scale + shift); all (groups, C). */
int coda_sa_pool_select_f32(const float *ymax, const float *ymin, const int32_t *amax, const int32_t *amin,
                            const float *stats, float *ysel, int32_t *sel, float *out, long long groups, int c,
                            void *stream);
/* pooled last layer, backward: d = gout where out > 0; sums = [sum d, sum d * (ysel - mean) * invstd]. */
int coda_sa_pool_bwd_stats_f32(const float *gout, const float *out, const float *ysel, const float *stats,
                               float *d, long long groups, int c, double *sums, void *stream);

/* De-duplicated groups of a ball query in one launch: ball_query pads a group with copies of its FIRST hit behind its
 * distinct hits (ball_query_gpu.cu:35-48), so group g's distinct rows are its first cnt[g] slots.
 *   grouped (groups, s_len, 3) float32 (channels-last grouped xyz), cnt (groups) int64, goff (groups + 1) int64 =
 *   exclusive prefix sum of cnt (goff[groups] = total).
 *   x (rows_padded, 3): row goff[g] + j = grouped[g][j] for j < cnt[g]; rows >= total are zero.
 *   row_weight (rows_padded): s_len - cnt[g] + 1 for the first row of a group (it stands for the copies), 1 for the
 *   other distinct rows, 0 for the padding.  goff32 (groups + 1) int32 = goff.
 * rows_padded >= total (the caller rounds it up to the GEMM's row granularity). */
int coda_sa_compact_groups_f32(const float *grouped, const int64_t *cnt, const int64_t *goff, float *x,
                               float *row_weight, int32_t *goff32, long long groups, int s_len, long long total,
                               long long rows_padded, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_SA_MLP_H */
