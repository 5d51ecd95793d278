// decoder_stack.hip -- host-side driver: the decoder stack's launch sequence from C++ (include/coda_stack.h).
// Mirrors fused_blocks._DecoderStack.forward / .backward launch for launch; every launch goes through the public
// entry points of the other headers.
#include "coda_stack.h"
#include "coda_attention.h"
#include "coda_gemm.h"
#include "coda_token_ops.h"
#include "common.hip.h"

#include <cstdlib>

#include <vector>

namespace coda {
namespace {

struct Dims {
  int nl, nq, b, e, ns, h, f;
  size_t R, RE, RF, LSE;  // rows, rows*E, rows*F, bsz*heads*nq
};
Dims dims_of(int nl, int nq, int bsz, int e, int ns, int nheads, int ffn) {
  Dims d{nl, nq, bsz, e, ns, nheads, ffn, 0, 0, 0, 0};
  d.R = static_cast<size_t>(nq) * bsz;
  d.RE = d.R * e;
  d.RF = d.R * ffn;
  d.LSE = static_cast<size_t>(bsz) * nheads * nq;
  return d;
}
bool bad_dims(int nl, int nq, int bsz, int e, int nheads, int ffn) {
  return nl <= 0 || nq <= 0 || bsz <= 0 || e <= 0 || nheads <= 0 || ffn <= 0 || e % nheads != 0 ||
         (e / nheads != 64 && e / nheads != 128) || e % 4 != 0 || ffn % 4 != 0;
}
size_t up4(size_t x) { return (x + 3) & ~static_cast<size_t>(3); }  // keep every sub-buffer 16-byte aligned

// saved activations of one layer (offsets in floats)
struct LayerWs {
  size_t y1, y1p, mean1, rstd1, qk, v1, attn1, lse1, s2, y2p, mean2, rstd2, q2, attn2, lse2, s3, y3, mean3, rstd3, h, s4,
      mean4, rstd4, total;
};
LayerWs layer_ws(const Dims &d) {
  LayerWs w{};
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += up4(n); return at; };
  w.y1 = take(d.RE); w.y1p = take(d.RE); w.mean1 = take(d.R); w.rstd1 = take(d.R);
  w.qk = take(2 * d.RE); w.v1 = take(d.RE); w.attn1 = take(d.RE); w.lse1 = take(d.LSE);
  w.s2 = take(d.RE); w.y2p = take(d.RE); w.mean2 = take(d.R); w.rstd2 = take(d.R);
  w.q2 = take(d.RE); w.attn2 = take(d.RE); w.lse2 = take(d.LSE);
  w.s3 = take(d.RE); w.y3 = take(d.RE); w.mean3 = take(d.R); w.rstd3 = take(d.R);
  w.h = take(d.RF);
  w.s4 = take(d.RE); w.mean4 = take(d.R); w.rstd4 = take(d.R);
  w.total = o;
  return w;
}
// forward temporaries shared by the layers: a1 | a2 | o | y2
size_t fwd_tmp_floats(const Dims &d) { return 4 * up4(d.RE); }

uint64_t op_seed(uint64_t base, int layer, int op) {
  uint64_t x = base + 0x9E3779B97F4A7C15ull * static_cast<uint64_t>(layer * 8 + op + 1);
  x ^= x >> 31;
  return x & 0x7fffffffffffffffull;
}

// the routing of gemm.py's _run: own split-K kernel for the launch-sized products, the library otherwise
int gemm_auto(int transb, int m, int n, int k, const float *a, long long lda, const float *b, long long ldb, float *c,
              long long ldc, const float *bias, int accumulate, void *stream) {
  if (static_cast<long long>(m) * n <= 2048LL * 256 && m % 64 == 0 && n % 64 == 0 && k % 128 == 0) {
    const int st = coda_sgemm_f32(transb, m, n, k, a, lda, b, ldb, c, ldc, bias, accumulate, stream);
    if (st != CODA_ENOSPC) return st;
  }
  return coda_gemm_f32(0, transb, m, n, k, a, lda, b, ldb, c, ldc, bias, accumulate, stream);
}
// y = x W^T (+ bias): x (m,k), W (n,k)
int linear(int m, int n, int k, const float *x, const float *w, long long ldw, const float *bias, float *y, long long ldy,
           void *s) {
  return gemm_auto(1, m, n, k, x, k, w, ldw, y, ldy, bias, 0, s);
}
// dx = dy W: dy (m,n'), W (n',k)
int dgrad(int m, int k, int nprime, const float *dy, long long lddy, const float *w, long long ldw, float *dx, int accumulate,
          void *s) {
  return gemm_auto(0, m, k, nprime, dy, lddy, w, ldw, dx, k, nullptr, accumulate, s);
}

// CODA_DEC_LN2=0: the layer-output norm and the next layer's norm1 as separate launches, forward and backward (A/B)
inline bool fuse_ln2() {
  static const bool on = [] { const char *e = getenv("CODA_DEC_LN2"); return !e || atoi(e) != 0; }();
  return on;
}

// CODA_DEC_FFN_BWD=0: the feed-forward's activation backward as its own pass behind the product (A/B)
inline bool ffn_bwd_epilogue() {
  static const bool on = [] { const char *v = getenv("CODA_DEC_FFN_BWD"); return !v || atoi(v) != 0; }();
  return on;
}

// CODA_DEC_QKV_ROWS=0: the self-attention's dq, dk, dv as three separate (R, E) matrices (A/B)
// (3E columns must fit the column-sum kernel: E <= 341, i.e. the 256-wide decoder; the 512-wide one keeps three matrices)
inline bool qkv_rows(int e) {
  static const bool on = [] { const char *v = getenv("CODA_DEC_QKV_ROWS"); return !v || atoi(v) != 0; }();
  return on && 3 * e <= 1024;
}

#define CODA_TRY(expr)           \
  do {                           \
    const int st__ = (expr);     \
    if (st__ != CODA_OK) return st__; \
  } while (0)

// acc (n floats) (+)= a + b
__global__ __launch_bounds__(256) void add3_kernel(float *__restrict__ acc, const float *__restrict__ a,
                                                   const float *__restrict__ b, size_t n4, int first) {
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 x = reinterpret_cast<const float4 *>(a)[i], y = reinterpret_cast<const float4 *>(b)[i];
  float4 r = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  if (!first) {
    const float4 o = reinterpret_cast<const float4 *>(acc)[i];
    r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
  }
  reinterpret_cast<float4 *>(acc)[i] = r;
}

}  // namespace
}  // namespace coda

using namespace coda;

CODA_API size_t coda_decoder_stack_ws_floats(int nl, int nq, int bsz, int e, int nheads, int ffn) {
  if (bad_dims(nl, nq, bsz, e, nheads, ffn)) return 0;
  const Dims d = dims_of(nl, nq, bsz, e, 0, nheads, ffn);
  return layer_ws(d).total * nl + fwd_tmp_floats(d);
}

CODA_API int coda_decoder_stack_fwd_f32(const CodaDecoderStack *a, void *stream) {
  if (!a || bad_dims(a->nl, a->nq, a->bsz, a->e, a->nheads, a->ffn) || a->ns <= 0) return CODA_EINVAL;
  if (a->ld_kv != 0 && (a->ld_kv < a->nl * a->e || a->ld_kv % 4 != 0)) return CODA_EINVAL;
  if (a->mfma_dtype < -1 || a->mfma_dtype > 2) return CODA_EINVAL;
  coda::CallOptions opt = coda::call_options();
  opt.mfma_dtype = a->mfma_dtype;  // the attention launches below read it (common.hip.h: per-call options)
  coda::ScopedCallOptions scope(opt);
  if (!a->tgt || !a->query_pos || !a->k_all || !a->v_all || !a->norm_g || !a->norm_b || !a->params || !a->outs || !a->ws)
    return CODA_EINVAL;
  const Dims d = dims_of(a->nl, a->nq, a->bsz, a->e, a->ns, a->nheads, a->ffn);
  const LayerWs lw = layer_ws(d);
  const int R = static_cast<int>(d.R), E = d.e, F = d.f, hd = E / d.h, ld_kv = a->ld_kv > 0 ? a->ld_kv : d.nl * E;
  const float scale = 1.0f / sqrtf(static_cast<float>(hd));
  float *tmp = a->ws + lw.total * d.nl;
  float *a1 = tmp, *a2 = tmp + up4(d.RE), *o = tmp + 2 * up4(d.RE), *y2 = tmp + 3 * up4(d.RE);
  const float *res = a->tgt;
  for (int l = 0; l < d.nl; ++l) {
    const float *const *P = a->params + 18 * l;
    const float *g1 = P[0], *b1n = P[1], *in1 = P[2], *ib1 = P[3], *ow1 = P[4], *ob1 = P[5], *g2 = P[6], *b2n = P[7],
                *in2 = P[8], *ib2 = P[9], *ow2 = P[10], *ob2 = P[11], *g3 = P[12], *b3n = P[13], *w1 = P[14], *fb1 = P[15],
                *w2 = P[16], *fb2 = P[17];
    float *W = a->ws + lw.total * l;
    // 1. y1 = LN1(res), y1p = y1 + query_pos -- for l > 0 already written by the layer below (step 7: the decoder's norm
    //    and this norm1 normalise the same s4 with the same mean / rstd; CODA_DEC_LN2=0: one launch each, A/B)
    if (l == 0 || !fuse_ln2())
      CODA_TRY(coda_tok_add_ln_fwd_f32(res, nullptr, nullptr, a->query_pos, g1, b1n, R, E, a->eps, 0.f, 0, nullptr, nullptr,
                                       W + lw.y1, W + lw.y1p, W + lw.mean1, W + lw.rstd1, stream));
    // 2. self-attention: [q|k] from y1p, v from y1
    CODA_TRY(linear(R, 2 * E, E, W + lw.y1p, in1, E, ib1, W + lw.qk, 2 * E, stream));
    CODA_TRY(linear(R, E, E, W + lw.y1, in1 + static_cast<size_t>(2) * E * E, E, ib1 + 2 * E, W + lw.v1, E, stream));
    CODA_TRY(coda_mha_fwd_f32(W + lw.qk, W + lw.qk + E, W + lw.v1, nullptr, W + lw.attn1, W + lw.lse1, d.b, d.h, d.nq, d.nq,
                              hd, 2 * E, 2 * E, E, scale, a->p_attn, op_seed(a->seed, l, 0), nullptr, stream));
    CODA_TRY(linear(R, E, E, W + lw.attn1, ow1, E, nullptr, a1, E, stream));
    // 3. s2 = res + drop(a1 + ob1), y2 = LN2(s2), y2p = y2 + query_pos
    CODA_TRY(coda_tok_add_ln_fwd_f32(a1, ob1, res, a->query_pos, g2, b2n, R, E, a->eps, a->p1, op_seed(a->seed, l, 1), nullptr,
                                     W + lw.s2, y2, W + lw.y2p, W + lw.mean2, W + lw.rstd2, stream));
    // 4. cross-attention on the pre-projected memory
    CODA_TRY(linear(R, E, E, W + lw.y2p, in2, E, ib2, W + lw.q2, E, stream));
    CODA_TRY(coda_mha_fwd_f32(W + lw.q2, a->k_all + static_cast<size_t>(l) * E, a->v_all + static_cast<size_t>(l) * E, nullptr,
                              W + lw.attn2, W + lw.lse2, d.b, d.h, d.nq, d.ns, hd, E, ld_kv, ld_kv, scale, a->p_attn,
                              op_seed(a->seed, l, 2), nullptr, stream));
    CODA_TRY(linear(R, E, E, W + lw.attn2, ow2, E, nullptr, a2, E, stream));
    // 5. s3 = s2 + drop(a2 + ob2), y3 = LN3(s3)
    CODA_TRY(coda_tok_add_ln_fwd_f32(a2, ob2, W + lw.s2, nullptr, g3, b3n, R, E, a->eps, a->p2, op_seed(a->seed, l, 3), nullptr,
                                     W + lw.s3, W + lw.y3, nullptr, W + lw.mean3, W + lw.rstd3, stream));
    // 6. feed-forward: h = drop(relu(y3 W1^T + fb1)), o = h W2^T
    {  // one launch where the own kernel takes the shape (bias + ReLU + dropout in its epilogue), else two
      const int st = coda_sgemm_relu_dropout_f32(1, R, F, E, W + lw.y3, E, w1, E, W + lw.h, F, fb1, a->p_ffn,
                                                 op_seed(a->seed, l, 4), stream);
      if (st == CODA_ENOSPC) {
        CODA_TRY(linear(R, F, E, W + lw.y3, w1, E, nullptr, W + lw.h, F, stream));
        CODA_TRY(coda_tok_bias_relu_dropout_fwd_f32(W + lw.h, fb1, R, F, a->p_ffn, op_seed(a->seed, l, 4), nullptr, W + lw.h,
                                                    stream));
      } else {
        CODA_TRY(st);
      }
    }
    CODA_TRY(linear(R, E, F, W + lw.h, w2, F, nullptr, o, E, stream));
    // 7. s4 = s3 + drop(o + fb2); the layer's output = decoder.norm(s4) -- and, in the same pass, the next layer's
    //    y1 = norm1(s4), y1p = y1 + query_pos
    if (l + 1 < d.nl && fuse_ln2()) {
      const float *const *Pn = a->params + 18 * (l + 1);
      float *Wn = a->ws + lw.total * (l + 1);
      CODA_TRY(coda_tok_add_ln_fwd2_f32(o, fb2, W + lw.s3, nullptr, a->norm_g, a->norm_b, Pn[0], Pn[1], a->query_pos, R, E,
                                        a->eps, a->p3, op_seed(a->seed, l, 5), nullptr, W + lw.s4, a->outs + d.RE * l, nullptr,
                                        Wn + lw.y1, Wn + lw.y1p, W + lw.mean4, W + lw.rstd4, stream));
    } else {
      CODA_TRY(coda_tok_add_ln_fwd_f32(o, fb2, W + lw.s3, nullptr, a->norm_g, a->norm_b, R, E, a->eps, a->p3,
                                       op_seed(a->seed, l, 5), nullptr, W + lw.s4, a->outs + d.RE * l, nullptr, W + lw.mean4,
                                       W + lw.rstd4, stream));
    }
    res = W + lw.s4;
  }
  return CODA_OK;
}

namespace coda {
namespace {
// backward scratch of one layer that must survive until the grouped launches at the end of the call
struct LayerBwd {
  size_t d_o, dh0, da2, dq, da1, dqkv, p_cn, p_c5, p_c3, p_c1, p_ffn, p_dq, p_dqkv, total;
};
LayerBwd layer_bwd(const Dims &d) {
  LayerBwd w{};
  size_t o = 0;
  auto take = [&](size_t n) { const size_t at = o; o += up4(n); return at; };
  const size_t bl = static_cast<size_t>(coda_tok_add_ln_bwd_blocks(static_cast<long long>(d.R), d.e));
  const size_t bf = static_cast<size_t>(coda_tok_bias_relu_dropout_bwd_blocks(static_cast<long long>(d.R), d.f));
  const size_t bc = static_cast<size_t>(coda_tok_colsum_blocks(static_cast<long long>(d.R), d.e));
  w.d_o = take(d.RE); w.dh0 = take(d.RF); w.da2 = take(d.RE); w.dq = take(d.RE); w.da1 = take(d.RE); w.dqkv = take(3 * d.RE);
  w.p_cn = take(bl * 3 * d.e); w.p_c5 = take(bl * 3 * d.e); w.p_c3 = take(bl * 3 * d.e); w.p_c1 = take(bl * 3 * d.e);
  const size_t bc3 = qkv_rows(d.e) ? static_cast<size_t>(coda_tok_colsum_blocks(static_cast<long long>(d.R), 3 * d.e)) : 0;  // (R, 3E) as one matrix
  w.p_ffn = take((bf > static_cast<size_t>(d.R / 32) ? bf : static_cast<size_t>(d.R / 32)) * d.f);  // (own product's epilogue: a partial per 32 rows)
  w.p_dq = take(bc * d.e); w.p_dqkv = take(3 * (bc > bc3 ? bc : bc3) * d.e);
  w.total = o;
  return w;
}
// shared temporaries: ds3 | dh (R*F) | dy3 | ds2 | dattn | delta | dxq | ds1 | dqk | dv1 | ds_next[2]
size_t bwd_tmp_floats(const Dims &d) { return 10 * up4(d.RE) + up4(d.RF) + up4(d.LSE); }
}  // namespace
}  // namespace coda

CODA_API size_t coda_decoder_stack_bwd_ws_floats(int nl, int nq, int bsz, int e, int nheads, int ffn) {
  if (bad_dims(nl, nq, bsz, e, nheads, ffn)) return 0;
  const Dims d = dims_of(nl, nq, bsz, e, 0, nheads, ffn);
  return layer_bwd(d).total * nl + bwd_tmp_floats(d);
}

CODA_API int coda_decoder_stack_bwd_f32(const CodaDecoderStack *a, const float *dstack, float *d_tgt, float *d_query_pos,
                                        float *dk_all, float *dv_all, float *const *grads, float *sums, float *bwd_ws,
                                        void *stream) {
  if (!a || bad_dims(a->nl, a->nq, a->bsz, a->e, a->nheads, a->ffn) || a->ns <= 0) return CODA_EINVAL;
  if (a->ld_kv != 0 && (a->ld_kv < a->nl * a->e || a->ld_kv % 4 != 0)) return CODA_EINVAL;
  if (a->mfma_dtype < -1 || a->mfma_dtype > 2) return CODA_EINVAL;
  coda::CallOptions opt = coda::call_options();
  opt.mfma_dtype = a->mfma_dtype;
  opt.attn_ds_ws = a->attn_ws;  // the attention backward calls below pick the dS workspace up (coda_mha_bwd_ws_f32's route)
  opt.attn_ds_bytes = a->attn_ws ? a->attn_ws_bytes : 0;
  coda::ScopedCallOptions scope(opt);
  if (!a->tgt || !a->query_pos || !a->k_all || !a->v_all || !a->norm_g || !a->params || !a->ws || !dstack || !d_tgt ||
      !d_query_pos || !dk_all || !dv_all || !grads || !sums || !bwd_ws)
    return CODA_EINVAL;
  const Dims d = dims_of(a->nl, a->nq, a->bsz, a->e, a->ns, a->nheads, a->ffn);
  const LayerWs lw = layer_ws(d);
  const LayerBwd lb = layer_bwd(d);
  const int R = static_cast<int>(d.R), E = d.e, F = d.f, hd = E / d.h, ld_kv = a->ld_kv > 0 ? a->ld_kv : d.nl * E;
  const float scale = 1.0f / sqrtf(static_cast<float>(hd));
  const int bl = coda_tok_add_ln_bwd_blocks(R, E), bf = coda_tok_bias_relu_dropout_bwd_blocks(R, F),
            bc = coda_tok_colsum_blocks(R, E);
  float *tmp = bwd_ws + lb.total * d.nl;
  size_t to = 0;
  auto ttake = [&](size_t n) { float *p = tmp + to; to += up4(n); return p; };
  float *ds3 = ttake(d.RE), *dh = ttake(d.RF), *dy3 = ttake(d.RE), *ds2 = ttake(d.RE), *dattn = ttake(d.RE),
        *delta = ttake(d.LSE), *dxq = ttake(d.RE), *ds1 = ttake(d.RE), *dqk = ttake(d.RE), *dv1 = ttake(d.RE);
  float *dsn[2] = {ttake(d.RE), ttake(d.RE)};
  std::vector<CodaTnProblem> tn;
  std::vector<CodaColsumItem> cs;
  tn.reserve(static_cast<size_t>(d.nl) * 8);
  cs.reserve(static_cast<size_t>(d.nl) * 7);
  // the grouped kernel's shape constraints; otherwise every weight gradient is a library call on the spot
  const bool grouped = R % 8 == 0 && E % 64 == 0 && F % 64 == 0;
  int tn_status = CODA_OK;
  auto add_tn_ld = [&](float *out, long long ldout, const float *dy, int m, long long lddy, const float *x, int n) {
    if (grouped) {
      tn.push_back(CodaTnProblem{dy, x, out, R, m, n, lddy, n, ldout});
    } else if (tn_status == CODA_OK) {
      tn_status = coda_gemm_f32(1, 0, m, n, R, dy, lddy, x, n, out, ldout, nullptr, 0, stream);
    }
  };
  auto add_tn = [&](float *out, long long ldout, const float *dy, int m, const float *x, int n) {
    add_tn_ld(out, ldout, dy, m, m, x, n);
  };
  auto add_cs = [&](const float *partials, float *out, int blocks, int n, int groups) {
    cs.push_back(CodaColsumItem{partials, out, blocks, n, groups, 0});
  };
  const float *ds_next = nullptr;
  for (int l = d.nl - 1; l >= 0; --l) {
    const float *const *P = a->params + 18 * l;
    const float *g1 = P[0], *in1 = P[2], *ow1 = P[4], *g2 = P[6], *in2 = P[8], *ow2 = P[10], *g3 = P[12], *w1 = P[14], *w2 = P[16];
    float *const *G = grads + 18 * l;
    const float *W = a->ws + lw.total * l;
    float *B = bwd_ws + lb.total * l;
    const float *res_in = l == 0 ? a->tgt : a->ws + lw.total * (l - 1) + lw.s4;
    float *S = sums + static_cast<size_t>(12) * E * l;  // [c1 | c3 | c5 | cn] x 3E
    // 7'. decoder.norm + residual + dropout of the layer output
    float *d_o = a->p3 > 0.f ? B + lb.d_o : ds3;  // dx == dres without dropout
    if (l + 1 < d.nl && fuse_ln2()) {
      // ... together with norm1 of the layer above (same s4, mean, rstd; upstream dv1 through y1, dqk through y1p, the
      // stream's own gradient ds1) and the positional embedding's gradient of that layer (dqk through y1p, dxq through
      // y2p): what were three launches at the end of the previous iteration (LN1 backward, add3) and this one
      const float *g1n = (a->params + 18 * (l + 1))[0];
      float *Bn = bwd_ws + lb.total * (l + 1);
      CODA_TRY(coda_tok_add_ln_bwd2_f32(dstack + d.RE * l, nullptr, dv1, dqk, ds1, W + lw.s4, W + lw.mean4, W + lw.rstd4,
                                        a->norm_g, g1n, R, E, a->p3, op_seed(a->seed, l, 5), nullptr, 2, dxq, d_query_pos,
                                        l + 1 == d.nl - 1 ? 1 : 0, ds3, a->p3 > 0.f ? B + lb.d_o : nullptr, B + lb.p_cn,
                                        Bn + lb.p_c1, stream));
    } else {
      CODA_TRY(coda_tok_add_ln_bwd_f32(dstack + d.RE * l, nullptr, ds_next, W + lw.s4, W + lw.mean4, W + lw.rstd4, a->norm_g, R,
                                       E, a->p3, op_seed(a->seed, l, 5), nullptr, ds3, a->p3 > 0.f ? B + lb.d_o : nullptr,
                                       B + lb.p_cn, nullptr, stream));
    }
    if (!(a->p3 > 0.f)) {  // the deferred weight gradient needs a buffer that survives the loop
      CODA_TRY(static_cast<int>(hipMemcpyAsync(B + lb.d_o, ds3, sizeof(float) * d.RE, hipMemcpyDeviceToDevice,
                                               static_cast<hipStream_t>(stream))));
      d_o = B + lb.d_o;
    }
    add_cs(B + lb.p_cn, S + 9 * E, bl, 3 * E, 1);       // [d decoder.norm.weight | .bias | d linear2.bias]
    // 6'. feed-forward
    {  // dh0 = relu-dropout backward of dh = do W2, with the bias gradient's partials: one launch where the own kernel
       // takes the shape (the activation's backward in its epilogue), else the product and the element-wise pass
      int st = ffn_bwd_epilogue() ? coda_sgemm_relu_dropout_bwd_f32(R, F, E, d_o, E, w2, F, W + lw.h, a->p_ffn, B + lb.dh0,
                                                                    B + lb.p_ffn, stream)
                                  : CODA_ENOSPC;
      if (st == CODA_ENOSPC) {
        CODA_TRY(dgrad(R, F, E, d_o, E, w2, F, dh, 0, stream));                  // dh = do W2
        CODA_TRY(coda_tok_bias_relu_dropout_bwd_f32(dh, W + lw.h, R, F, a->p_ffn, B + lb.dh0, B + lb.p_ffn, nullptr, stream));
        add_cs(B + lb.p_ffn, G[15], bf, F, 1);                                    // d linear1.bias
      } else {
        CODA_TRY(st);
        add_cs(B + lb.p_ffn, G[15], coda_sgemm_relu_dropout_bwd_blocks(R), F, 1);
      }
    }
    add_tn(G[16], F, d_o, E, W + lw.h, F);                                        // d linear2.weight (E,F) = do^T h
    add_tn(G[14], E, B + lb.dh0, F, W + lw.y3, E);                                // d linear1.weight (F,E) = dh0^T y3
    CODA_TRY(dgrad(R, E, F, B + lb.dh0, F, w1, E, dy3, 0, stream));               // dy3 = dh0 W1
    // 5'. LN3 + residual
    float *da2 = a->p2 > 0.f ? B + lb.da2 : nullptr;
    CODA_TRY(coda_tok_add_ln_bwd_f32(dy3, nullptr, ds3, W + lw.s3, W + lw.mean3, W + lw.rstd3, g3, R, E, a->p2,
                                     op_seed(a->seed, l, 3), nullptr, ds2, da2, B + lb.p_c5, nullptr, stream));
    if (!da2) {
      CODA_TRY(static_cast<int>(hipMemcpyAsync(B + lb.da2, ds2, sizeof(float) * d.RE, hipMemcpyDeviceToDevice,
                                               static_cast<hipStream_t>(stream))));
    }
    da2 = B + lb.da2;
    add_cs(B + lb.p_c5, S + 6 * E, bl, 3 * E, 1);       // [d norm3.weight | .bias | d multihead_attn.out_proj.bias]
    // 4'. cross-attention
    add_tn(G[10], E, da2, E, W + lw.attn2, E);                                    // d out_proj.weight = da2^T attn2
    CODA_TRY(dgrad(R, E, E, da2, E, ow2, E, dattn, 0, stream));
    CODA_TRY(coda_mha_bwd_f32(W + lw.q2, a->k_all + static_cast<size_t>(l) * E, a->v_all + static_cast<size_t>(l) * E, nullptr,
                              W + lw.attn2, W + lw.lse2, dattn, B + lb.dq, dk_all + static_cast<size_t>(l) * E,
                              dv_all + static_cast<size_t>(l) * E, delta, d.b, d.h, d.nq, d.ns, hd, E, ld_kv, ld_kv, 0, ld_kv,
                              ld_kv, scale, a->p_attn, op_seed(a->seed, l, 2), nullptr, stream));
    add_tn(G[8], E, B + lb.dq, E, W + lw.y2p, E);                                 // query rows of d in_proj_weight
    CODA_TRY(coda_tok_colsum_f32(B + lb.dq, 1, R, E, B + lb.p_dq, nullptr, stream));
    add_cs(B + lb.p_dq, G[9], bc, E, 1);                                          // query part of d in_proj_bias
    CODA_TRY(dgrad(R, E, E, B + lb.dq, E, in2, E, dxq, 0, stream));
    // 3'. LN2 + residual (the normalised output was used only through y2p)
    float *da1 = a->p1 > 0.f ? B + lb.da1 : nullptr;
    CODA_TRY(coda_tok_add_ln_bwd_f32(nullptr, dxq, ds2, W + lw.s2, W + lw.mean2, W + lw.rstd2, g2, R, E, a->p1,
                                     op_seed(a->seed, l, 1), nullptr, ds1, da1, B + lb.p_c3, nullptr, stream));
    if (!da1) {
      CODA_TRY(static_cast<int>(hipMemcpyAsync(B + lb.da1, ds1, sizeof(float) * d.RE, hipMemcpyDeviceToDevice,
                                               static_cast<hipStream_t>(stream))));
    }
    da1 = B + lb.da1;
    add_cs(B + lb.p_c3, S + 3 * E, bl, 3 * E, 1);       // [d norm2.weight | .bias | d self_attn.out_proj.bias]
    // 2'. self-attention
    add_tn(G[4], E, da1, E, W + lw.attn1, E);                                     // d out_proj.weight
    CODA_TRY(dgrad(R, E, E, da1, E, ow1, E, dattn, 0, stream));
    float *dqkv = B + lb.dqkv;
    if (qkv_rows(E)) {
      // dq | dk | dv as the columns of ONE (R, 3E) matrix (the attention kernels take row strides): the in_proj bias
      // gradient is one column sum over 3E columns, and d y1p = [dq | dk] in_proj_weight[:2E] ONE product over K = 2E
      // instead of two accumulating ones
      CODA_TRY(coda_mha_bwd_f32(W + lw.qk, W + lw.qk + E, W + lw.v1, nullptr, W + lw.attn1, W + lw.lse1, dattn, dqkv, dqkv + E,
                                dqkv + 2 * E, delta, d.b, d.h, d.nq, d.nq, hd, 2 * E, 2 * E, E, 3 * E, 3 * E, 3 * E, scale,
                                a->p_attn, op_seed(a->seed, l, 0), nullptr, stream));
      CODA_TRY(coda_tok_colsum_f32(dqkv, 1, R, 3 * E, B + lb.p_dqkv, nullptr, stream));
      add_cs(B + lb.p_dqkv, G[3], coda_tok_colsum_blocks(R, 3 * E), 3 * E, 1);     // d in_proj_bias (3E)
      add_tn_ld(G[2], E, dqkv, 2 * E, 3 * E, W + lw.y1p, E);                      // rows of q and k: [dq | dk]^T y1p
      add_tn_ld(G[2] + static_cast<size_t>(2) * E * E, E, dqkv + 2 * E, E, 3 * E, W + lw.y1, E);
      CODA_TRY(dgrad(R, E, 2 * E, dqkv, 3 * E, in1, E, dqk, 0, stream));
      CODA_TRY(dgrad(R, E, E, dqkv + 2 * E, 3 * E, in1 + static_cast<size_t>(2) * E * E, E, dv1, 0, stream));
    } else {
      CODA_TRY(coda_mha_bwd_f32(W + lw.qk, W + lw.qk + E, W + lw.v1, nullptr, W + lw.attn1, W + lw.lse1, dattn, dqkv,
                                dqkv + d.RE, dqkv + 2 * d.RE, delta, d.b, d.h, d.nq, d.nq, hd, 2 * E, 2 * E, E, 0, 0, 0, scale,
                                a->p_attn, op_seed(a->seed, l, 0), nullptr, stream));
      CODA_TRY(coda_tok_colsum_f32(dqkv, 3, R, E, B + lb.p_dqkv, nullptr, stream));
      add_cs(B + lb.p_dqkv, G[3], bc, E, 3);                                      // d in_proj_bias (3E)
      add_tn(G[2], E, dqkv, E, W + lw.y1p, E);
      add_tn(G[2] + static_cast<size_t>(E) * E, E, dqkv + d.RE, E, W + lw.y1p, E);
      add_tn(G[2] + static_cast<size_t>(2) * E * E, E, dqkv + 2 * d.RE, E, W + lw.y1, E);
      CODA_TRY(dgrad(R, E, E, dqkv, E, in1, E, dqk, 0, stream));
      CODA_TRY(dgrad(R, E, E, dqkv + d.RE, E, in1 + static_cast<size_t>(E) * E, E, dqk, 1, stream));
      CODA_TRY(dgrad(R, E, E, dqkv + 2 * d.RE, E, in1 + static_cast<size_t>(2) * E * E, E, dv1, 0, stream));
    }
    // 1'. LN1: the block's input WAS the stream, so its gradient is d(stream) + d(LayerNorm path)
    float *out_ds = l == 0 ? d_tgt : dsn[l & 1];
    if (!fuse_ln2()) {
      CODA_TRY(coda_tok_add_ln_bwd_f32(dv1, dqk, ds1, res_in, W + lw.mean1, W + lw.rstd1, g1, R, E, 0.f, 0, nullptr, out_ds,
                                       nullptr, B + lb.p_c1, nullptr, stream));
      // query_pos receives dqk (through y1p) and dxq (through y2p)
      const size_t n4 = d.RE / 4;
      hipLaunchKernelGGL(add3_kernel, dim3(static_cast<unsigned>((n4 + 255) / 256)), dim3(256), 0,
                         static_cast<hipStream_t>(stream), d_query_pos, dqk, dxq, n4, l == d.nl - 1 ? 1 : 0);
    } else if (l == 0) {
      // the first layer's norm1 has no layer below to share a launch with; the positional gradient rides along
      CODA_TRY(coda_tok_add_ln_bwd2_f32(dv1, dqk, nullptr, nullptr, ds1, res_in, W + lw.mean1, W + lw.rstd1, g1, nullptr, R, E,
                                        0.f, 0, nullptr, 1, dxq, d_query_pos, d.nl == 1 ? 1 : 0, out_ds, nullptr, B + lb.p_c1,
                                        nullptr, stream));
    }  // (l > 0: norm1's backward runs at the top of the next iteration, with the layer below's output norm)
    add_cs(B + lb.p_c1, S, bl, 3 * E, 1);               // [d norm1.weight | .bias | unused]
    ds_next = out_ds;
  }
  if (tn_status != CODA_OK) return tn_status;
  CODA_TRY(coda_tok_colsum_finalize_grouped_f32(cs.data(), static_cast<int>(cs.size()), stream));
  CODA_TRY(coda_grouped_gemm_tn_f32(tn.data(), static_cast<int>(tn.size()), stream));
  return launch_status();
}
