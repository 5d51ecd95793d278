/*
 * coda_clip_tower.h -- C ABI of the frozen CLIP image tower: the ViT the reference calls as
 * `self.clip_model.encode_image(crops)` once per scene (models/model_3detr.py:1091, :1605), i.e.
 * VisionTransformer.forward (CLIP/clip/model.py:612-659) over ResidualAttentionBlock (:295-316) -- inference only,
 * no gradients: the tower is frozen (`requires_grad=False`, models/model_3detr.py:331-333).
 *
 *   tokens  = [class_embedding ; conv1(image) as (grid^2, width)] + positional_embedding      (:613-619)
 *   x       = ln_pre(tokens)                                                                    (:620)
 *   12 x :    x += out_proj(attention(in_proj(ln_1(x))));  x += c_proj(QuickGELU(c_fc(ln_2(x)))) (:313-316)
 *   cls     = ln_post(x[class token]) @ proj ;  all = ln_post(x) @ proj                         (:655-659)
 *
 * One call runs ALL crops of a step (the reference: one call per scene, batch 32).  Per transformer block this
 * is 7 launches with no stand-alone element-wise pass: LayerNorm, GEMM(+bias), attention, GEMM(+bias, +residual
 * in place), LayerNorm, GEMM(+bias, QuickGELU as the library's swish epilogue: QuickGELU(u) = swish(1.702 u) /
 * 1.702, the two constants ride on the GEMMs' alpha), GEMM(+bias, +residual).  GEMMs are hipBLASLt
 * (coda_gemm_ex); LayerNorm, the patch gather, the token assembly and the attention core are this library's
 * kernels (csrc/vit_tower.hip).  Tokens are kept sequence-first (L, n, width), as the reference permutes them.
 *
 * dtype CODA_DTYPE_F16 (the reference's: clip.load converts the tower to fp16, CLIP/clip/model.py:1146-1167):
 * activations and weight matrices are IEEE half, every product accumulates in fp32 on the matrix cores
 * (v_mfma_f32_32x32x16_f16), LayerNorm / soft-max statistics are fp32.  The attention kernel keeps the whole
 * (32 query x L) score block of a wave in registers: L <= 288 tokens (ViT-B/32, B/16, L/14 at 224 px), head
 * width 64 (every CLIP ViT); CODA_ENOSPC otherwise.
 * dtype CODA_DTYPE_F32: everything in float32 (attention = coda_mha_fwd_f32, any L) -- the parity mode.
 *
 * Weight matrices are of `dtype`; LayerNorm parameters, biases, class / positional embeddings are float32.
 * `layers` is a HOST array; all other pointers are device memory.  `images` (n,3,res,res) float32, already
 * normalised (coda_crop_resize_f32's output).  Outputs are of `dtype`: cls (n, out_dim); all_tokens
 * (L, n, out_dim) sequence-first, or NULL to skip it (the distillation branch only uses cls).
 * `workspace`: coda_vit_workspace_bytes(desc, n, all_tokens != NULL) bytes, 256-byte aligned; CODA_ENOSPC if
 * smaller.  Everything is enqueued on `stream`; nothing is read back.
 */
#ifndef CODA_CLIP_TOWER_H
#define CODA_CLIP_TOWER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct CodaVitLayer {
  const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
  const void *in_w;    /* (3 width, width)  attn.in_proj_weight */
  const float *in_b;   /* (3 width)         attn.in_proj_bias */
  const void *out_w;   /* (width, width)    attn.out_proj.weight */
  const float *out_b;
  const void *fc_w;    /* (mlp, width)      mlp.c_fc.weight */
  const float *fc_b;
  const void *proj_w;  /* (width, mlp)      mlp.c_proj.weight */
  const float *proj_b;
} CodaVitLayer;

typedef struct CodaVit {
  int32_t dtype;       /* CODA_DTYPE_F32 | CODA_DTYPE_F16 (include/coda_gemm.h) */
  int32_t resolution, patch, width, nlayers, heads, mlp, out_dim;
  float eps;           /* LayerNorm epsilon (1e-5) */
  int32_t pad_;
  const void *conv_w;  /* (width, 3 * patch * patch)  conv1.weight flattened */
  const float *cls;    /* (width)      class_embedding */
  const float *pos;    /* (L, width)   positional_embedding, L = (resolution / patch)^2 + 1 */
  const float *ln_pre_g, *ln_pre_b, *ln_post_g, *ln_post_b;
  const void *proj;    /* (width, out_dim) */
  const CodaVitLayer *layers;
} CodaVit;

size_t coda_vit_workspace_bytes(const CodaVit *desc, int n, int with_tokens);

int coda_vit_fwd(const CodaVit *desc, const float *images, int n, void *cls, void *all_tokens, void *workspace,
                 size_t workspace_bytes, void *stream);

/* The attention core of the fp16 tower on its own (parity tests, tools/bench_clip_tower.py):
 * qkv (L, n, 3 * heads * 64) half, sequence-first packed in-projection; out (L, n, heads * 64) half =
 * softmax(q k^T / 8) v per image and head.  L <= 288. */
int coda_vit_attention_f16(const void *qkv, void *out, int n, int l, int heads, void *stream);

/* 1 / 0: the c_fc GEMMs of coda_vit_fwd calls of this dtype carry QuickGELU as the library's swish epilogue / run
 * bias-only followed by an activation pass (the library had no such kernel); -1: no call yet. */
int coda_vit_quickgelu_fused(int dtype);

#ifdef __cplusplus
}
#endif
#endif /* CODA_CLIP_TOWER_H */
