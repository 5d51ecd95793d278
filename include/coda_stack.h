/*
 * coda_stack.h -- the 3DETR decoder stack as ONE call each way.
 *
 * What it replaces: the launch sequence of TransformerDecoder.forward over its pre-norm layers
 * (models/transformer.py:97-143 driving TransformerDecoderLayer.forward_pre :556-580, return_intermediate with the
 * shared final LayerNorm :126-135) and of its autograd backward -- per layer ~12 launches forward and ~23 backward
 * (LayerNorm / residual / dropout kernels of coda_token_ops.h, the attention core of coda_attention.h, the
 * projection GEMMs of coda_gemm.h).  The host-side mirror used to issue them one by one from Python (~5 ms of
 * host time per training step for ~280 launches); these two entry points run the identical sequence from C++.
 * No new arithmetic: every launch is one of the entry points of the headers named above, in the order and with
 * the operands the Python mirror (coda_neurips2023_amd/fused_blocks.py, _DecoderStack) uses.
 *
 * Scope: pre-norm layers, query positional embedding present, no attention masks, LayerNorm everywhere (the
 * configuration of every CoDA recipe); anything else stays on the per-launch path.
 *
 * Layout conventions: activations (tokens, batch, E) row-major = (rows, E) with rows = nq * bsz.  `params` holds,
 * per layer, 18 device pointers in this order: norm1.weight, norm1.bias, self_attn.in_proj_weight (3E,E),
 * self_attn.in_proj_bias, self_attn.out_proj.weight, self_attn.out_proj.bias, norm2.weight, norm2.bias,
 * multihead_attn.in_proj_weight, multihead_attn.in_proj_bias, multihead_attn.out_proj.weight,
 * multihead_attn.out_proj.bias, norm3.weight, norm3.bias, linear1.weight (F,E), linear1.bias, linear2.weight (E,F),
 * linear2.bias.  k_all / v_all: the memory's key / value projections of ALL layers, (ns * bsz, nl * E) (layer l =
 * columns l*E ..), computed by the caller (two large GEMMs), with a row stride of `ld_kv` floats (0: nl * E; a
 * multiple of 4).  A stride of nl * E + 64 is worth using: with 8 layers of width 256 the rows of a layer's slice
 * are 8 KB apart and land on a fraction of the memory channels -- the cross-attention dK/dV kernel takes 120 us on
 * such slices and 96 us with the padded stride (tools/probe_attn_layout.py).  dk_all / dv_all of the backward use the
 * same stride.  Dropout: counter-based, op seeds are derived from `seed`; the backward must be given the forward's
 * value.
 */
#ifndef CODA_STACK_H
#define CODA_STACK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct CodaDecoderStack {
  int nl, nq, bsz, e, ns, nheads, ffn;
  float eps, p_attn, p1, p2, p_ffn, p3;
  uint64_t seed;
  const float *tgt;        /* (nq, bsz, E): the stream entering layer 0 */
  const float *query_pos;  /* (nq, bsz, E) */
  const float *k_all, *v_all;
  const float *norm_g, *norm_b; /* decoder.norm */
  const float *const *params;   /* HOST array of nl * 18 device pointers */
  float *outs;             /* (nl, nq, bsz, E): decoder-normed output of every layer */
  float *ws;               /* saved activations: coda_decoder_stack_ws_floats() floats, written by fwd, read by bwd */
  int ld_kv;               /* row stride of k_all / v_all / dk_all / dv_all in floats (0: nl * E) */
  int mfma_dtype;          /* MFMA operand type of the attention core (coda_attention.h): -1 library default, 0 fp32, 1 bf16, 2 bf16x3 */
  float *attn_ws;          /* backward only, optional: workspace of the attention core's backward (coda_mha_bwd_ws_bytes of
                              the cross-attention problem -- the partial dQ tiles of the one-kernel backward; the
                              self-attention's need is smaller; shared by the layers), NULL = the two-kernel backward */
  size_t attn_ws_bytes;
} CodaDecoderStack;

/* floats of `ws` / of the backward's scratch for these dimensions (0 on invalid dimensions) */
size_t coda_decoder_stack_ws_floats(int nl, int nq, int bsz, int e, int nheads, int ffn);
size_t coda_decoder_stack_bwd_ws_floats(int nl, int nq, int bsz, int e, int nheads, int ffn);

int coda_decoder_stack_fwd_f32(const CodaDecoderStack *d, void *stream);

/* dstack (nl, nq, bsz, E): gradient of `outs`.  Outputs: d_tgt (nq,bsz,E), d_query_pos (nq,bsz,E), dk_all / dv_all
 * (ns*bsz, nl*E: layer l's columns, row stride ld_kv), grads: HOST array of nl * 18 device pointers in the order of `params` for the
 * MATRIX-shaped gradients and the projection biases -- entries 2, 3, 4, 8, 9, 10, 14, 15, 16 (of
 * multihead_attn.in_proj_weight / bias only the query rows [0, E) are written: the key / value rows follow from
 * dk_all / dv_all on the caller's side); the other entries are ignored.  The LayerNorm and output-bias gradients
 * come packed, as the reductions produce them: sums (nl, 4, 3E), per layer
 *   [0] d norm1.weight | d norm1.bias | unused        [1] d norm2.weight | d norm2.bias | d self_attn.out_proj.bias
 *   [2] d norm3.weight | d norm3.bias | d multihead_attn.out_proj.bias
 *   [3] layer l's share of d decoder.norm.weight | of d decoder.norm.bias | d linear2.bias
 * (the caller sums the decoder.norm shares over the layers).  bwd_ws: coda_decoder_stack_bwd_ws_floats() floats. */
int coda_decoder_stack_bwd_f32(const CodaDecoderStack *d, const float *dstack, float *d_tgt, float *d_query_pos,
                               float *dk_all, float *dv_all, float *const *grads, float *sums, float *bwd_ws,
                               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_STACK_H */
