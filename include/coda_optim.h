/*
 * coda_optim.h -- C ABI of the tail of the training step (engine.py:161-164): gradient-norm clipping
 * (torch.nn.utils.clip_grad_norm_(model.parameters(), args.clip_gradient), default 0.1, main.py:52) and the AdamW
 * update (optimizer.py:35: torch.optim.AdamW over one or two parameter groups) -- three launches for all parameter
 * tensors of a group instead of the framework's per-tensor / multi-tensor kernel sequences.
 *
 * A parameter list is a device array of CodaOptTensor (float32, dense; g == NULL: no gradient this step, the tensor
 * is skipped as torch's optimizers skip it) plus a chunk map built once from the sizes: int32 pairs
 * (tensor index, chunk index), one per block, tensor i contributing ceil(n_i / coda_opt_chunk_elems()) chunks.
 *
 *   coda_opt_grad_sumsq_f32   *sumsq (double, device) = sum over all gradients of g^2   (zeroed by the call)
 *   coda_opt_grad_scale_f32   g *= min(1, max_norm / (sqrt(*sumsq) + 1e-6)); *total_norm = sqrt(*sumsq) (may be
 *                             NULL): clip_grad_norm_'s in-place scaling (norm type 2)
 *   coda_opt_adamw_f32        torch.optim.AdamW's update (amsgrad=False, maximize=False) with each tensor's own
 *                             step count:  p -= lr wd p;  m += (1-b1)(g-m);  v = b2 v + (1-b2) g^2;
 *                             p -= (lr / (1-b1^step)) m / (sqrt(v) / sqrt(1-b2^step) + eps)
 *
 * Everything is enqueued on `stream`; nothing is read back.  Summation order of the norm is not fixed (double
 * atomics).
 */
#ifndef CODA_OPTIM_H
#define CODA_OPTIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct CodaOptTensor {
  float *p, *g, *m, *v;  /* parameter, gradient, exp_avg, exp_avg_sq (m / v unused by the clipping calls) */
  long long n;
  double step;           /* this tensor's 1-based update count (AdamW only): torch keeps one per parameter */
} CodaOptTensor;

int coda_opt_chunk_elems(void);
int coda_opt_grad_sumsq_f32(const CodaOptTensor *table, const int32_t *chunks, int nchunks, double *sumsq,
                            void *stream);
int coda_opt_grad_scale_f32(const CodaOptTensor *table, const int32_t *chunks, int nchunks, const double *sumsq,
                            float max_norm, float *total_norm, void *stream);
/* Data-parallel gradient packing (main.py:993-996 wraps the model in DistributedDataParallel, which copies every
 * gradient into its buckets with one kernel per tensor): p[i] = g[i] * scale for every tensor of the table (p = the
 * tensor's slice of a flat all-reduce buffer; zeros where g == NULL), one launch. */
int coda_opt_pack_f32(const CodaOptTensor *table, const int32_t *chunks, int nchunks, float scale, void *stream);
int coda_opt_adamw_f32(const CodaOptTensor *table, const int32_t *chunks, int nchunks, float lr, float beta1,
                       float beta2, float eps, float weight_decay, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_OPTIM_H */
