/*
 * coda_align_loss.h -- C ABI of the CLIP-space alignment losses.
 *
 * For every decoder layer l the reference evaluates, on the 512-d region embedding e of each
 * proposal (criterion.py:924-943 and :598-644):
 *
 *   l1[l]  = sum_{b,q}  sum_c | e*w - gt*w |                       (w = crop mask of (b,q))
 *   ce[l]  = sum_{b,q}  conf * CE( t * <e/(|e|+1e-32), text_j>_j , label )
 *
 * as ~40 element-wise / reduction passes over (B,nq,512) tensors per layer, forward and again
 * in backward.  Here all L layers are one pass each way: a wave owns a proposal row, keeps it
 * in registers, and produces the row's two partial losses (forward) or its gradient (backward;
 * the forward quantities are recomputed, nothing is stored).  The normalisers
 * (sum(w)*512, #(conf > 1e-32)) and loss weights stay with the caller: it scales the per-layer
 * sums and hands the resulting upstream gradients `g` to the backward.
 *
 * emb is addressed through element strides (ld_l, ld_b, ld_q; the channel axis is contiguous),
 * so the (layer, scene, query) view of the heads' (layer, query, scene) buffer is read in place.
 * E must be a multiple of 64 with E <= 1024.  Conventions as in coda_pointnet2.h.
 */
#ifndef CODA_ALIGN_LOSS_H
#define CODA_ALIGN_LOSS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* partial (L*B*nq, 2): [row][0] = sum_c |e*w - gt*w|, [row][1] = conf * CE; rows ordered (l,b,q).
 * gt (B,nq,E), wmask (B,nq), text (B,ncls,E), logit_scale (1, device), labels (L,B,nq) int64,
 * conf (L,B,nq). */
int coda_align_loss_fwd_f32(const float *emb, long long ld_l, long long ld_b, long long ld_q,
                            const float *gt, const float *wmask, const float *text,
                            const float *logit_scale, const int64_t *labels, const float *conf,
                            int nl, int b, int nq, int e, int ncls, float *partial, void *stream);

/* demb (L,B,nq,E) dense = g[l][0] * d l1[l] / d emb + g[l][1] * d ce[l] / d emb;  g (L,2) device. */
int coda_align_loss_bwd_f32(const float *emb, long long ld_l, long long ld_b, long long ld_q,
                            const float *gt, const float *wmask, const float *text,
                            const float *logit_scale, const int64_t *labels, const float *conf,
                            const float *g, int nl, int b, int nq, int e, int ncls, float *demb,
                            void *stream);

/* ---- the same two terms at the stage-2 class counts (232 / 1201 prompts, models/model_3detr.py:321): the class logits
 * and their gradient are DENSE PRODUCTS on the matrix cores (coda_gemm.h; the host side, align_loss.py, issues them), and
 * these three entry points are the row-wise pieces around them.  Rows are ordered (l, b, q) as above.
 *   coda_align_rows_fwd_f32: partial[row][0] = sum_c |e w - gt w|; stat[row] = (|e|, 1 / (|e| + 1e-32));
 *                            ehat (rows x E, dense) = e / (|e| + 1e-32)                     -> logits = ehat . text^T
 *   coda_align_ce_f32:       logits (rows, row stride ld; the first ncls of ncols columns are classes).  g == NULL:
 *                            partial[row][1] = conf * CE(t * logits[row], label).  g (L,2) given: the row is overwritten
 *                            with g[l][1] * conf * t * (softmax - onehot), zero in columns ncls .. ncols - 1
 *                                                                                          -> dh = dlogits . text
 *   coda_align_rows_bwd_f32: demb = g[l][0] * d l1 / d e + (dh / (n + eps) - e <dh, e> / (n (n + eps)^2)) */
int coda_align_rows_fwd_f32(const float *emb, long long ld_l, long long ld_b, long long ld_q, const float *gt,
                            const float *wmask, int nl, int b, int nq, int e, float *ehat, float *stat,
                            float *partial, void *stream);
int coda_align_rows_bwd_f32(const float *emb, long long ld_l, long long ld_b, long long ld_q, const float *gt,
                            const float *wmask, const float *stat, const float *dh, const float *g, int nl, int b,
                            int nq, int e, float *demb, void *stream);
int coda_align_ce_f32(float *logits, long long ld, int ncls, int ncols, const float *logit_scale,
                      const int64_t *labels, const float *conf, const float *g, float *partial, long long rows,
                      long long rows_per_layer, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_ALIGN_LOSS_H */
