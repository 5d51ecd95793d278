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


/* ---- MFMA pipeline (csrc/sa_mfma.hip) -------------------------------------------------------------------------
 * The shared MLP [3 -> C1 -> C2 -> C3] + BN + ReLU + max-pool of the pre-encoder (pointnet2_modules.py:247-253,
 * pytorch_utils.py:8-33; C1, C2, C3 = 64, 128, 256, models/model_3detr.py:3935-3944) with the 1x1 convolutions as
 * hand-written fp32-MFMA GEMMs (v_mfma_f32_32x32x2_f32) fused with what surrounds them, instead of library GEMMs
 * between streaming kernels:
 *   forward   layer 1 is never formed in memory: its batch statistics follow from the 3x3 second moments of the
 *             grouped xyz (y1 = x W1^T is linear), its activations are recomputed from x in the prologue of layer 2;
 *             BN + ReLU of layer l is applied in the prologue of layer l + 1; per-channel sum / sum of squares of
 *             the GEMM's output in its epilogue; the last layer pools (max or min by the sign of gamma, with the
 *             arg) in its epilogue, so no kernel ever reads the 256-channel activation for statistics or pooling.
 *             y2 and y3 (pre-BN) are stored once for the backward.
 *   backward  dy of a layer is formed on the fly from (y, upstream gradient, BN coefficients) while the tile is
 *             staged -- the pooled sparse gradient of the last layer never becomes a dense tensor -- and feeds
 *             dA = dy W (epilogue: ReLU mask of the layer below + its BN-backward sums) and dW = dy^T A (per-workgroup
 *             partial tiles, fixed-order reduction).  Layer 1's dW / dgamma / dbeta follow in closed form from
 *             five sums per channel of layer 2's dA epilogue and the xyz moments.
 * Rows: the DISTINCT rows of the ball-query groups (see "De-duplicated groups" above), packed on the device without
 * a host read-back: every kernel takes the row count from group_offsets[groups] in device memory and splits the
 * rows evenly over `nblocks` workgroups (coda_sa_mfma_blocks()).
 */
/* workgroups (= row ranges) of the forward / dx kernels (kind 0) and of the dw kernels (kind 1, = partial tiles) */
int coda_sa_mfma_blocks(int kind);
/* Supported widths of coda_sa_mfma_*: (cin, cout) = (64, 128) [first: w1 != NULL] and (128, 256) [pooled]. */
int coda_sa_mfma_supported(int c1, int c2, int c3, int s_len);

/* grouped (groups, s_len, 3) channels-last grouped xyz, idx (groups, s_len) its ball-query indices.
 * dedup != 0: group g keeps its distinct rows (slot 0 and the slots whose index differs from slot 0's);
 * dedup == 0: all s_len rows.  Outputs (capacity groups * s_len rows):
 *   x (rows,3), row_weight (rows), group_offsets (groups + 1) int32, row_group (rows) int32 = (group << 6) | row
 *   index inside the group (s_len <= 64),
 *   moments (10 doubles): sum w, sum w x_j (3), sum w x_i x_j (xx, xy, xz, yy, yz, zz) over the packed rows.
 * counts: scratch of `groups` int32.  `zero` / `nzero`: optional buffer of doubles zeroed by the call (the
 * statistics accumulators of the forward kernels that follow: saves their memset launches). */
int coda_sa_pack_groups_f32(const float *grouped, const int32_t *idx, int dedup, float *x, float *row_weight,
                            int32_t *group_offsets, int32_t *row_group, double *moments, int32_t *counts,
                            double *zero, int nzero, long long groups, int s_len, void *stream);
/* sums[0:C] = sum_p w_p y1, sums[C:2C] = sum_p w_p y1^2 for y1 = x . w1[c] from the moments (w1 (C,3)). */
int coda_sa_l1_sums_f32(const double *moments, const float *w1, double *sums, int c, void *stream);

/* One layer, forward: y_out (rows, cout) = act_in W^T, W (cout, cin) row-major (the conv weight),
 *   act_in = relu(scale_in * y_in + shift_in), y_in = src (rows, cin), or, first layer (w1 != NULL, (cin,3)):
 *   y_in = src (rows,3) . w1^T recomputed on the fly;  stats_in = [scale, shift, ...][cin].
 * sums (2*cout doubles, ACCUMULATED: zero them before, e.g. through coda_sa_pack_groups_f32) += [sum w y, sum w y^2].
 * Pooled layer (ysel != NULL): per (group, channel) the pre-BN value with the largest sign(gamma[c]) * y and its row
 * index inside the group (lowest on ties): ysel / sel (groups, cout) for the part of a group inside the workgroup
 * that holds its first row, part_y / part_sel (nblocks, cout) + part_gid (nblocks) for the part a group has in the
 * next workgroup; coda_sa_pool_finish_f32 merges them.  y_out may be NULL (not stored). */
int coda_sa_mfma_fwd_f32(const float *src, const float *w1, const float *stats_in, const float *w,
                         const float *row_weight, const int32_t *group_offsets, const int32_t *row_group,
                         long long groups, int s_len, int cin, int cout, float *y_out, double *sums,
                         const float *gamma, float *ysel, int32_t *sel, float *part_y, int32_t *part_sel,
                         int32_t *part_gid, int nblocks, void *stream);
/* merges the partial pools, then out = relu(ysel * scale + shift) (stats = [scale, shift, ..][c]) */
int coda_sa_pool_finish_f32(float *ysel, int32_t *sel, const float *part_y, const int32_t *part_sel,
                            const int32_t *part_gid, const int32_t *group_offsets, const float *gamma,
                            const float *stats, float *out, long long groups, int c, int nblocks, void *stream);

/* One layer, backward.  dy (rows, cout) is never stored:
 *   dy[p][c] = a[c] * (dsel[p][c] - w_p * (m1[c] + (y_out[p][c] - mean[c]) * invstd[c] * m2[c]))
 *   dsel = d[g][c] if row-in-group == sel[g][c] else 0   (pooled layer: d, sel (groups, cout), layout 1 coef)
 *        = dmid[p][c]                                    (hidden layer: dmid (rows, cout), layout 0 coef)
 *   coef as written by coda_sa_bn_bwd_coef_f32 (layout 1: [a, m1, m2, mean, invstd], layout 0: [scale, shift,
 *   mean, invstd, a, m1, m2]).
 * dx:  dmid_in (rows, cin) = (dy W) where relu'(act_in) else 0 (NULL for the first layer: not stored),
 *      sums_in (zeroed by the call): [sum dmid_in, sum dmid_in * xhat_in][cin], first layer: + [sum dmid_in * x_j]
 *      (3 more blocks of cin: 5 * cin doubles).  stats_in = [scale, shift, mean, invstd][cin].
 * dw:  dw (cout, cin) = dy^T act_in through `partials` (nblocks, cout, cin) floats, summed in workgroup order. */
int coda_sa_mfma_bwd_dx_f32(const float *y_out, const float *dmid, const float *d, const int32_t *sel,
                            const float *coef, int layout, const float *w, const float *src_in, const float *w1,
                            const float *stats_in, const float *row_weight, const int32_t *group_offsets,
                            const int32_t *row_group, long long groups, int s_len, int cin, int cout,
                            float *dmid_in, double *sums_in, int nblocks, void *stream);
int coda_sa_mfma_bwd_dw_f32(const float *y_out, const float *dmid, const float *d, const int32_t *sel,
                            const float *coef, int layout, const float *src_in, const float *w1,
                            const float *stats_in, const float *row_weight, const int32_t *group_offsets,
                            const int32_t *row_group, long long groups, int s_len, int cin, int cout,
                            float *partials, float *dw, int nblocks, void *stream);
/* First layer, closed form.  sums5 = the 5*c doubles of coda_sa_mfma_bwd_dx_f32 (local rows); sums_bn = [sum d,
 * sum d xhat] over the whole batch (= sums5 unless SyncBatchNorm reduced a copy across ranks); n <= 0: eval mode.
 * dw1 (c,3), dbeta / dgamma (c) (NULL: skip; from the LOCAL sums). */
int coda_sa_l1_bwd_f32(const double *sums5, const double *sums_bn, double n, const float *gamma, const float *stats,
                       const double *moments, const float *w1, float *dw1, float *dbeta, float *dgamma, int c,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_SA_MLP_H */
