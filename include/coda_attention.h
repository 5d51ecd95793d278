/*
 * coda_attention.h -- C ABI of the fused multi-head attention core (gfx950, fp32 MFMA).
 *
 * Replaces the scaled-dot-product core inside torch.nn.MultiheadAttention as the
 * reference's 3DETR layers use it (models/transformer.py:422,470-471,506-507,
 * 566-573 -> torch.nn.functional.multi_head_attention_forward):
 *     P = softmax(scale * Q K^T  [masked -> -inf]);  O = dropout_p(P) V
 * for the three shapes of the path: encoder self-attention (2048 x 2048), decoder
 * self-attention (nq x nq) and decoder cross-attention (nq x 2048), 4 heads,
 * head_dim 64 (dec_dim 256) or 128 (dec_dim 512).
 *
 * Layout: the SEQUENCE-FIRST projections are read in place,
 *     q (L, B, H, D),  k / v (S, B, H, D),  out (L, B, H, D)   float32,
 * element (l, b, h, c) of q at (l*B + b)*ldq + h*D + c (same for k / ldk, v / ldv):
 * ld* = H*D for dense (L,B,E) tensors, 3*H*D for the slices of a packed in-projection
 * (no copy of the chunks is needed).  out, dout, dq, dk, dv are dense (ld = H*D).  lse (B, H, L) float32 keeps log-sum-exp of the
 * scaled scores for the backward pass.  mask: optional uint8 (B, H, L, S), non-zero
 * = key not attended (the boolean radius mask of MaskedTransformerEncoder,
 * transformer.py:154-190), or NULL.
 *
 * Dropout: keep(b,h,l,s) is a counter-based hash of (seed, b*H+h, l, s) evaluated
 * identically in forward and backward; kept probabilities are scaled by 1/(1-p).
 * p = 0 disables it (eval mode).  `seed_dev` (optional, device memory) is XOR-folded
 * into `seed` inside the kernels: a captured hipGraph whose replays bump that word
 * draws a fresh mask per replay while `seed` keeps the calls of one step apart.
 *
 * Numerics: fp32 inputs, fp32 MFMA (v_mfma_f32_32x32x2_f32, bit-equal to an fmaf
 * chain), fp32 online softmax; parity target 1e-3 relative vs the fp32 reference.
 * Rows whose keys are ALL masked produce 0 (torch produces NaN).
 *
 * Return values and stream semantics as in coda_pointnet2.h.  head_dim must be 64
 * or 128 (CODA_EINVAL otherwise).
 */
#ifndef CODA_ATTENTION_H
#define CODA_ATTENTION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int coda_mha_fwd_f32(const float *q, const float *k, const float *v,
                     const uint8_t *mask, float *out, float *lse, int b, int h,
                     int l, int s, int d, int ldq, int ldk, int ldv, float scale,
                     float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                     void *stream);

/* dq (L,B,H,D), dk / dv (S,B,H,D) are fully written.  lddq / lddk / lddv: floats between
 * consecutive batch rows of the gradient outputs (0 = dense, H*D), so that they can be
 * written straight into column slices of a packed buffer, like ldq / ldk / ldv for the
 * inputs.  `delta` is a (B,H,L) float scratch provided by the caller (rowsum(dout * out)). */
int coda_mha_bwd_f32(const float *q, const float *k, const float *v,
                     const uint8_t *mask, const float *out, const float *lse,
                     const float *dout, float *dq, float *dk, float *dv,
                     float *delta, int b, int h, int l, int s, int d, int ldq,
                     int ldk, int ldv, int lddq, int lddk, int lddv, float scale,
                     float dropout_p, uint64_t seed, const uint64_t *seed_dev, void *stream);

/* The same backward, issued in parts: bit 0 = the delta pre-pass (rowsum(dout * out) -> `delta`), bit 1 = the
 * dK/dV kernel, bit 2 = the dQ kernel (7 = coda_mha_bwd_f32).  dK/dV and dQ both read `delta`; a caller that
 * needs dQ first (the decoder's cross-attention: only dQ is on the dependency chain of its backward, dK/dV of all
 * layers are consumed at the very end) issues parts 1|4 on its stream and part 2 on a second stream behind an
 * event.  Outputs of the parts not requested may be NULL. */
int coda_mha_bwd_parts_f32(const float *q, const float *k, const float *v,
                           const uint8_t *mask, const float *out, const float *lse,
                           const float *dout, float *dq, float *dk, float *dv,
                           float *delta, int b, int h, int l, int s, int d, int ldq,
                           int ldk, int ldv, int lddq, int lddk, int lddv, float scale,
                           float dropout_p, uint64_t seed, const uint64_t *seed_dev, int parts, void *stream);

/* MFMA operand type -- a per-call argument of the *_opt entry points (the plain ones use the library default;
 * the library keeps no mutable process-wide state, see coda_pointnet2.h): 0 = fp32 operands
 * (v_mfma_f32_32x32x2_f32), 1 = bf16 operands
 * (v_mfma_f32_32x32x16_bf16): Q, K, V, dO and the probabilities are rounded to bf16 on their way
 * into the matrix cores, accumulation / softmax / lse / every tensor in memory stay fp32.  This is
 * BASELINE.json configs[4] ("bf16 MFMA attention"); parity target there is 2e-2 relative to the
 * fp32 reference (bf16 has 8 significand bits).  Dropout masks are identical in all modes.
 * 2 = every fp32 operand carried as THREE bf16 pieces (hi + mid + lo = the fp32 value exactly) and every
 * product evaluated as the six piece products of order <= 2 with fp32 accumulation: fp32-level results (what
 * is dropped is <= 2^-24 relative, the size of fp32's own product rounding; tests/test_attention_x3_gpu.py
 * measures the error against float64 next to the fp32-MFMA kernels') at 6/16 of the fp32 MFMA cost, on the
 * matrix cores, which -- unlike the fp32 MFMA -- run concurrently with the soft-max VALU work.  EXPERIMENTAL and
 * off by default: the operand splitting is itself VALU work, and the measured gain is 1.2-1.3x on the
 * long-sequence forward and dQ kernels only; every other problem (dK/dV, the decoder shapes, head_dim 128) runs the
 * fp32-MFMA kernels in this mode too.
 * -1 = the library default: environment variable CODA_ATTN_DTYPE ("bf16" / "1" -> 1, "bf16x3" / "x3" / "2" -> 2,
 * else 0), read once; coda_mha_get_mfma_dtype() returns it. */
int coda_mha_fwd_opt_f32(const float *q, const float *k, const float *v,
                         const uint8_t *mask, float *out, float *lse, int b, int h,
                         int l, int s, int d, int ldq, int ldk, int ldv, float scale,
                         float dropout_p, uint64_t seed, const uint64_t *seed_dev,
                         int mfma_dtype, void *stream);
int coda_mha_bwd_parts_opt_f32(const float *q, const float *k, const float *v,
                               const uint8_t *mask, const float *out, const float *lse,
                               const float *dout, float *dq, float *dk, float *dv,
                               float *delta, int b, int h, int l, int s, int d, int ldq,
                               int ldk, int ldv, int lddq, int lddk, int lddv, float scale,
                               float dropout_p, uint64_t seed, const uint64_t *seed_dev, int parts,
                               int mfma_dtype, void *stream);
int coda_mha_get_mfma_dtype(void);

/* The whole backward with a caller-provided workspace of coda_mha_bwd_ws_bytes() bytes (0: this problem does not use
 * one -- NULL / 0 may be passed and the call equals coda_mha_bwd_parts_opt_f32 with parts = 7).  Long unmasked
 * sequences at head width 64 (the encoder's 2048 x 2048 self-attention), fp32 MFMA operands: the dK/dV kernel
 * leaves dS = P (dP - delta)
 * in the workspace, (B, H, L, S) floats, and dQ = scale dS K is a plain GEMM -- the backward then executes S, dP, dV,
 * dK, dQ once each (10 L S d flops per head) instead of recomputing S and dP in a second kernel (14).  Same results
 * as the two-kernel form up to the summation order of dQ.  The workspace is scratch: nothing is kept in it.
 * Round 6: of those five products, S and dP (inside the dK/dV kernel: both operands are staged tiles) and the dS K GEMM
 * run on the bf16 matrix cores with every operand in three bf16 pieces (x = hi + mid + lo exactly, six piece products,
 * fp32 accumulation; errors below the fp32 MFMA's own, drift of the bf16 accumulate cancelled by alternating signs);
 * the GEMM's K^T pieces take another 24 KB per head and 64 keys behind dS (when s is a multiple of 64), which
 * coda_mha_bwd_ws_bytes includes.  CODA_ATTN_DKV_X3=0 / CODA_ATTN_DQ_X3=0: the fp32-MFMA forms (A/B).
 *
 * Short query sequences (round 6; l < 1024, whole 32-row tiles, head width 64, no mask, fp32 MFMA operands -- the
 * decoder's cross-attention, s >= 1024 and a multiple of 128, and its self-attention, s < 1024): ONE kernel forms dK, dV
 * and, from the same dS, the dQ tile of every (key block, query tile) pair; the workspace holds those partial tiles,
 * (B, H, key blocks of 128 | 32 keys, L, 64) floats, and a second small kernel adds the key blocks in FIXED order and
 * scales: 10 L S d flops per head instead of 14, rowsum(dO * O) formed inside (`delta` is not written on this route),
 * no atomics and no waiting among workgroups, results bit-identical from call to call.  Without a workspace these
 * problems run the two-kernel form.  CODA_ATTN_FUSED_BWD=0 switches the route off (A/B). */
size_t coda_mha_bwd_ws_bytes(int b, int h, int l, int s, int d);
int coda_mha_bwd_ws_f32(const float *q, const float *k, const float *v, const uint8_t *mask, const float *out,
                        const float *lse, const float *dout, float *dq, float *dk, float *dv, float *delta, int b,
                        int h, int l, int s, int d, int ldq, int ldk, int ldv, int lddq, int lddk, int lddv,
                        float scale, float dropout_p, uint64_t seed, const uint64_t *seed_dev, void *workspace,
                        size_t workspace_bytes, int mfma_dtype, void *stream);

/* Measurement aid (bench.py's live roofline figures; no reference counterpart).  While
 * enabled, each kernel launched by coda_mha_fwd_f32 / coda_mha_bwd_f32 for a problem with
 * l >= min_len and s >= min_len carries a HIP event pair as the start / stop events OF ITS DISPATCH (hipExtLaunchKernelGGL:
 * the kernel's own begin / end timestamps, what a rocprofv3 kernel trace reports; events recorded around the launch also
 * measured the launch gap and two marker packets, +3 us on a 10-20 us kernel).
 * coda_mha_timing_enable(min_len < 0) disables; every call drops the records taken so far.
 * coda_mha_timing_collect synchronises the recorded events and writes up to `cap` records
 * (kind: 0 forward, 1 delta, 2 dK/dV, 3 dQ, 4 dQ as the dS K GEMM, 5 the one-kernel backward, 6 the sum of its partial dQ
 * tiles, 7 the bf16 K^T pieces of the bf16x3 dS K GEMM; the call's l and s; milliseconds); returns the
 * number written, CODA_EINVAL, or -(1000 + hipError_t) if an event query failed.
 * coda_mha_timing_enable_kinds additionally restricts the records to the kinds whose bit is set in `kind_mask`
 * (bit k = kind k above).  A dispatch that carries events is not free for its NEIGHBOURS: the queue runs it with a
 * completion signal of its own, and the kernel trace shows 5-9 us of idle queue before and after it (round 6:
 * twelve timed launches per step cost the headline 1.5 %), so bench.py's timed region records the dominant kernel only. */
int coda_mha_timing_enable(int min_len);
int coda_mha_timing_enable_kinds(int min_len, unsigned kind_mask);
int coda_mha_timing_collect(int *kind, int *l, int *s, float *ms, int cap);

#ifdef __cplusplus
}
#endif
#endif /* CODA_ATTENTION_H */
