/*
 * coda_token_ops.h -- C ABI of the token-wise streaming kernels around the library GEMMs of
 * the model's MLP stacks and transformer layers.
 *
 * Part 1, batch-norm MLP blocks.  The reference builds its prediction heads, the
 * encoder->decoder projection and the query projection from GenericMLP
 * (models/helpers.py:45-112): Conv1d(k=1, no bias) -> BatchNorm1d -> ReLU [-> Dropout] on
 * (batch, C, tokens) tensors, six heads over the same decoder output
 * (models/model_3detr.py:1566-1601, 1617-1660).  Here the activations are kept
 * CHANNELS-LAST as z (G, R, C): G independent stacks ("groups": the heads), R tokens,
 * C channels; the 1x1 convolutions are (batched) library GEMMs on that layout and these
 * kernels do everything between two GEMMs in one pass each:
 *
 *   forward    stats -> finalize -> act        a = dropout(relu(z * scale + shift))
 *   backward   act_bwd_stats -> bwd_finalize -> act_bwd_apply
 *
 * Batch statistics are summed in fp64: the two statistics kernels leave one partial sum per row block
 * ("parts", coda_tok_bn_parts() of them), the finalize kernels add them in index order (no atomics: 512 blocks
 * queueing on 2*C result addresses took 6 x the streaming time, and the result is deterministic).  With
 * SyncBatchNorm the caller sums the parts itself, all-reduces the (G,2,C) result and passes it with nparts = 1.  Dropout masks come from a
 * counter hash of (seed, element index) and are regenerated, not stored, in the backward.
 *
 * C must be a multiple of 4 with C/4 dividing 256 (4 ... 1024).  Conventions as in
 * coda_pointnet2.h: raw device pointers, `stream` is a hipStream_t, 0 / CODA_EINVAL /
 * hipError_t status.
 */
#ifndef CODA_TOKEN_OPS_H
#define CODA_TOKEN_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* number of row blocks (= partial sums per group and channel) of the two statistics kernels for this shape (>= 1),
 * or CODA_EINVAL */
int coda_tok_bn_parts(int groups, long long rows, int c);

/* parts (P,G,2,C) doubles, P = coda_tok_bn_parts(), every entry written by the call:
 * sum_p parts[p][g][0][c] = sum_r z, sum_p parts[p][g][1][c] = sum_r z^2 */
int coda_tok_bn_stats_f32(const float *z, int groups, long long rows, int c, double *parts,
                          void *stream);

/* Batch-norm parameters of this step from the sums over `count` tokens -- sums (nparts,G,2,C), added over the
 * leading index in order (nparts = 1: sums that are complete already, e.g. all-reduced):
 * prm (G,4,C) = scale (gamma*invstd), shift (beta - mean*scale), mean, invstd;
 * stat (G,2,C) = mean, unbiased variance (what the running statistics take; may be NULL).
 * gamma / beta (G,C). */
int coda_tok_bn_finalize_f32(const double *sums, int nparts, const float *gamma, const float *beta,
                             int groups, int c, double count, float eps, float *prm,
                             float *stat, void *stream);

/* a = dropout_p(relu(z * scale + shift)); `relu` 0/1; a may alias z.  seed_dev: optional
 * device-resident 64-bit word folded into the seed (hipGraph replays), may be NULL. */
int coda_tok_bn_act_f32(const float *z, const float *prm, int groups, long long rows, int c,
                        int relu, float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                        float *a, void *stream);

/* d = da * keep/(1-p) where the activation is positive (relu) else da * keep/(1-p);
 * parts (P,G,2,C) doubles as above: sum_r d, sum_r d * xhat */
int coda_tok_bn_act_bwd_stats_f32(const float *da, const float *z, const float *prm,
                                  int groups, long long rows, int c, int relu,
                                  float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                                  double *parts, void *stream);

/* From the local sums (nparts_local,G,2,C) (-> dgamma = sum d*xhat, dbeta = sum d, both (G,C) fp32) and the
 * all-reduced sums (G,2,C) over `count` tokens (NULL: the local sums, single process):
 * prmb (G,3,C) = gamma*invstd, sum d / count, sum d*xhat / count. */
int coda_tok_bn_bwd_finalize_f32(const double *sums_local, int nparts_local, const double *sums_total,
                                 const float *gamma, const float *prm, int groups, int c,
                                 double count, float *prmb, float *dgamma, float *dbeta,
                                 void *stream);

/* dz = prmb[0] * (d - prmb[1] - xhat * prmb[2]); dz may alias da */
int coda_tok_bn_act_bwd_apply_f32(const float *da, const float *z, const float *prm,
                                  const float *prmb, int groups, long long rows, int c,
                                  int relu, float dropout_p, uint64_t seed,
                                  const uint64_t *seed_dev, float *dz, void *stream);

/*
 * Part 2, the glue of the pre-norm transformer layers (models/transformer.py:457-494,
 * 558-594): between two GEMMs the reference runs bias add, Dropout, residual add, LayerNorm
 * and the positional-embedding add as separate passes over the (tokens, C) activations,
 * forward and backward.  One kernel each way here:
 *
 *   v = dropout_p(x + bias)        bias, dropout optional
 *   s = res + v                    res optional (s == v without it)
 *   y = LayerNorm(s) * gamma + beta            gamma == NULL: no norm, only s is produced
 *   yp = y + pos                   pos optional
 *
 * x, res, pos, s, y, yp are (rows, C) row-major; C a multiple of 4, C <= 1024; mean / rstd
 * (rows).  s_out may be NULL when s == x (no bias, no dropout, no residual).
 */
int coda_tok_add_ln_fwd_f32(const float *x, const float *bias, const float *res, const float *pos,
                            const float *gamma, const float *beta, long long rows, int c, float eps,
                            float dropout_p, uint64_t seed, const uint64_t *seed_dev, float *s_out,
                            float *y_out, float *yp_out, float *mean, float *rstd, void *stream);

/* number of (3,C) partial-sum slots the backward needs in `partials` */
int coda_tok_add_ln_bwd_blocks(long long rows, int c);

/* Backward of the above.  dy / dyp / ds are the gradients of y / yp / s (each may be NULL);
 *   dres = ds + LayerNorm_backward(dy + dyp)      (written to dres_out; the gradient of res)
 *   dx   = dropout-masked dres                    (written to dx_out; may be NULL when there is
 *                                                  no dropout: dx == dres)
 * partials (blocks,3,C): per-block column sums of [(dy+dyp)*xhat, (dy+dyp), dx] = the
 * contributions to dgamma, dbeta, dbias.  With sums_out (3,C) != NULL the call also reduces
 * them (fixed order: deterministic) with a second launch; with sums_out == NULL the caller does,
 * e.g. with coda_tok_colsum_finalize_f32. */
int coda_tok_add_ln_bwd_f32(const float *dy, const float *dyp, const float *ds, const float *s,
                            const float *mean, const float *rstd, const float *gamma, long long rows,
                            int c, float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                            float *dres_out, float *dx_out, float *partials, float *sums_out,
                            void *stream);

/* Two LayerNorms of ONE residual stream (round 6: the decoder's layer-output norm and the next layer's norm1 normalise the
 * same s with the same mean / rstd and differ only in their affine maps -- models/transformer.py:213-240 calls them
 * back to back).  Forward: the first kernel's (s, y, yp, mean, rstd) plus y2 = LayerNorm(s) * gamma2 + beta2 and
 * yp2 = y2 + pos2.  Backward: head 1 (dy, dyp, gamma) and head 2 (dy2, dyp2, gamma2; gamma2 == NULL: one head) enter the
 * SAME correction terms -- dres = rstd (g - mean(g) - xhat mean(g xhat)) + ds with g = d gamma + d2 gamma2 --, head 2's
 * [sum d2 xhat | sum d2 | 0] partials go to partials2 (same (blocks, 3, C) layout).  dpos_from = 1 / 2: the gradient of the
 * positional embedding that entered through yp / yp2 is accumulated on the way, dpos_acc (=, when dpos_init, else +=)
 * dyp|dyp2 + dpos_extra (dpos_extra optional); 0: none. */
int coda_tok_add_ln_fwd2_f32(const float *x, const float *bias, const float *res, const float *pos,
                             const float *gamma, const float *beta, const float *gamma2,
                             const float *beta2, const float *pos2, long long rows, int c, float eps,
                             float dropout_p, uint64_t seed, const uint64_t *seed_dev, float *s_out,
                             float *y_out, float *yp_out, float *y2_out, float *yp2_out, float *mean,
                             float *rstd, void *stream);
int coda_tok_add_ln_bwd2_f32(const float *dy, const float *dyp, const float *dy2, const float *dyp2,
                             const float *ds, const float *s_in, const float *mean, const float *rstd,
                             const float *gamma, const float *gamma2, long long rows, int c,
                             float dropout_p, uint64_t seed, const uint64_t *seed_dev, int dpos_from,
                             const float *dpos_extra, float *dpos_acc, int dpos_init, float *dres_out,
                             float *dx_out, float *partials, float *partials2, void *stream);

/* out[j] = sum_b partials[b][j], j < n (n = 3*C above) */
int coda_tok_colsum_finalize_f32(const float *partials, int blocks, int n, float *out, void *stream);

/* The same reduction for many (partials, out) pairs in one launch: a stack-level backward (the decoder as one
 * autograd node) leaves ~70 of them open -- LayerNorm / bias gradients that nothing reads before the node
 * returns -- and closes them together.  `items` is HOST memory read during the call (kernel arguments, 120
 * items per launch); every item: out[g][j] = sum_b partials[g][b][j] for g < groups, j < n. */
typedef struct CodaColsumItem {
  const float *partials;
  float *out;
  int blocks, n, groups, pad_;
} CodaColsumItem;
int coda_tok_colsum_finalize_grouped_f32(const CodaColsumItem *items, int count, void *stream);

/* Column sums of x (G, rows, C) -> out (G, C) (the bias gradients of the projections):
 * per-block partials (G, blocks, C) with blocks = coda_tok_colsum_blocks(rows, c), then a
 * fixed-order reduction (two launches of one call; out == NULL: partials only, the caller reduces them later,
 * e.g. with coda_tok_colsum_finalize_grouped_f32).  C a multiple of 4, C <= 1024. */
int coda_tok_colsum_blocks(long long rows, int c);
int coda_tok_colsum_f32(const float *x, int groups, long long rows, int c, float *partials,
                        float *out, void *stream);

/* Feed-forward activation: a = dropout_p(relu(h + bias)) on (rows, C); a may alias h; C/4 must
 * divide 256.  Backward: dz = da / (1-p) where a > 0, else 0 (a dropped or clamped element has
 * a == 0 either way); partials (blocks,1,C) column sums of dz, reduced into dbias (C) by the
 * same call when dbias != NULL. */
int coda_tok_bias_relu_dropout_fwd_f32(const float *h, const float *bias, long long rows, int c,
                                       float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                                       float *a, void *stream);
int coda_tok_bias_relu_dropout_bwd_blocks(long long rows, int c);
int coda_tok_bias_relu_dropout_bwd_f32(const float *da, const float *a, long long rows, int c,
                                       float dropout_p, float *dz, float *partials, float *dbias,
                                       void *stream);

/* Part 3, the Fourier coordinate embedding of the decoder (models/position_embedding.py:97-130) in one launch:
 *   u = (xyz - range_lo[b]) / (range_hi[b] - range_lo[b])   (skipped when the range pointers are NULL)
 *   phase = (u * 2*pi) @ gauss[:, :half]                     (gauss (3, >= half) row-major, row stride ld_gauss)
 *   out[b][i][0:half] = sin(phase), out[b][i][half:2*half] = cos(phase)
 * xyz (b, n, 3), range_lo / range_hi (b, 3), out (b, n, 2*half) float32 -- the reference returns its transpose
 * (b, 2*half, n) as a view of exactly this buffer.  One rounding per operation, products summed left to right. */
int coda_fourier_pos_embed_f32(const float *xyz, const float *range_lo, const float *range_hi, const float *gauss,
                               int ld_gauss, float *out, int b, int n, int half, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_TOKEN_OPS_H */
