// pseudo_label.hip -- the label side of CoDA's image branch (include/coda_clip_labels.h).
//
//  * clip_weak_labels_kernel: soft-max over the class prompts of (unit-norm image embedding . text embedding) *
//    temperature, reduced to (max probability, arg-max) per proposal -- models/model_3detr.py:1153-1172, 1614-1631
//    (weak labels of the alignment loss) and :1110-1123, 1497-1505 (classification of novel-box candidates).  The
//    reference materialises the normalised embeddings, the (B, K, ncls) logits and probabilities; here a workgroup
//    owns 32 proposals, the products run on the fp32 MFMA and only two numbers per proposal are written.
//  * pseudo_box_filter_kernel: torchvision.ops.nms on the projected rectangles + the 3-D IoU test against the
//    ground truth + the objectness threshold of the stage-2 mining loop (:1305-1426), one workgroup per scene.
#include "coda_clip_labels.h"
#include "common.hip.h"

#include <math.h>

namespace coda {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWlD = 512;          // CLIP's joint embedding width (hard-coded 512 in the reference, :975-977)
constexpr int kWlStride = 516;     // LDS row stride in floats: 16-byte aligned rows, conflict-free ds_read_b128
constexpr int kWlThreads = 256;
constexpr int kWlWaves = kWlThreads / kWave;
constexpr size_t kWlLds = sizeof(float) * (32 * kWlStride + 3 * kWlWaves * 32);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// running soft-max state of one row over the classes seen so far: (max logit, sum of exp(logit - max), first arg-max)
struct RowState {
  float m, s;
  int arg;
};
__device__ __forceinline__ void merge(RowState &a, float m2, float s2, int arg2) {
  if (m2 == -INFINITY) return;
  if (a.m == -INFINITY) {
    a = RowState{m2, s2, arg2};
    return;
  }
  if (m2 > a.m) {
    a.s = a.s * expf(a.m - m2) + s2;
    a.m = m2;
    a.arg = arg2;
  } else {
    a.s += s2 * expf(m2 - a.m);
    if (m2 == a.m && arg2 < a.arg) a.arg = arg2;  // first maximum
  }
}

__global__ __launch_bounds__(kWlThreads) void clip_weak_labels_kernel(
    const float *__restrict__ emb, long long emb_stride, const float *__restrict__ text, long long set_stride,
    const float *__restrict__ scale_p, const float *__restrict__ row_mask, float *__restrict__ score,
    int64_t *__restrict__ label, int rows, int rows_per_set, int ncls) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float *s_e = reinterpret_cast<float *>(smem);  // [32][kWlStride]: unit-norm embeddings of this tile
  float *s_m = s_e + 32 * kWlStride;             // [waves][32]
  float *s_s = s_m + kWlWaves * 32;
  int *s_a = reinterpret_cast<int *>(s_s + kWlWaves * 32);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int row0 = blockIdx.x * 32;

  // stage + normalise: wave w owns rows 8w .. 8w+7, a lane 8 of a row's 512 values (two coalesced 16-byte loads)
#pragma unroll 2
  for (int q = 0; q < 8; ++q) {
    const int r = w * 8 + q;
    const int gr = min(row0 + r, rows - 1);
    const float4 *src = reinterpret_cast<const float4 *>(emb + static_cast<long long>(gr) * emb_stride);
    float4 v0 = src[lane], v1 = src[lane + 64];
    float ss = v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w + v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
    const float nrm = sqrtf(wave_sum(ss)) + 1e-32f;  // e / (|e| + 1e-32), :1159-1160
    v0.x /= nrm; v0.y /= nrm; v0.z /= nrm; v0.w /= nrm;
    v1.x /= nrm; v1.y /= nrm; v1.z /= nrm; v1.w /= nrm;
    *reinterpret_cast<float4 *>(s_e + r * kWlStride + lane * 4) = v0;
    *reinterpret_cast<float4 *>(s_e + r * kWlStride + 256 + lane * 4) = v1;
  }
  __syncthreads();

  // logits^T = text . emb^T on v_mfma_f32_32x32x2_f32: A = text (i = class), B = emb (j = proposal), so a lane holds
  // 16 classes of ONE proposal (column j = lane & 31) and the soft-max statistics reduce in registers.  The two
  // k-slots of the instruction take k = s and k = 256 + s (lane half h), so each lane walks 256 contiguous floats of
  // its text row (global, 16-byte loads, a 128-byte line consumed within one unrolled batch) and of its embedding row
  // (LDS, ds_read_b128).  The 4 waves take the 32-class tiles round-robin.
  const int h = lane >> 5, li = lane & 31;
  const float scale = *scale_p;
  const float *tset = text + static_cast<long long>(row0 / rows_per_set) * set_stride;
  const float *erow = s_e + li * kWlStride + h * 256;
  RowState st{-INFINITY, 0.f, 0};
  const int ntiles = (ncls + 31) / 32;
  for (int ct = w; ct < ntiles; ct += kWlWaves) {
    const int cls = min(ct * 32 + li, ncls - 1);
    const float4 *trow = reinterpret_cast<const float4 *>(tset + static_cast<long long>(cls) * kWlD + h * 256);
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int s0 = 0; s0 < 64; s0 += 8) {
      float4 ta[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ta[j] = trow[s0 + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 eb = *reinterpret_cast<const float4 *>(erow + (s0 + j) * 4);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[j].x, eb.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[j].y, eb.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[j].z, eb.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[j].w, eb.w, acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // accumulator r of lane (h, li): class row (r & 3) + 8 (r >> 2) + 4 h
      const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (c < ncls) {
        const float v = acc[r] * scale;
        if (v > st.m) {
          st.s = st.s * expf(st.m - v) + 1.f;  // (first class: 0 * exp(-inf) + 1)
          st.m = v;
          st.arg = c;
        } else {
          st.s += expf(v - st.m);
        }
      }
    }
  }
  // the other half of the classes of this proposal sits in lane ^ 32
  merge(st, __shfl_xor(st.m, 32), __shfl_xor(st.s, 32), __shfl_xor(st.arg, 32));
  if (h == 0) {
    s_m[w * 32 + li] = st.m;
    s_s[w * 32 + li] = st.s;
    s_a[w * 32 + li] = st.arg;
  }
  __syncthreads();
  if (tid < 32 && row0 + tid < rows) {
    RowState t{s_m[tid], s_s[tid], s_a[tid]};
#pragma unroll
    for (int q = 1; q < kWlWaves; ++q) merge(t, s_m[q * 32 + tid], s_s[q * 32 + tid], s_a[q * 32 + tid]);
    const int row = row0 + tid;
    const bool masked = row_mask && row_mask[row] < 1.f;
    score[row] = masked ? 0.f : 1.f / t.s;  // max of the soft-max = exp(0) / sum
    label[row] = t.arg;
  }
}

// ---- stage-2 candidate filter ---------------------------------------------------------------------------------
constexpr int kPfThreads = 256;
constexpr int kPfMaxK = 1024, kPfMaxG = 128;

__device__ __forceinline__ float iou2d(const float4 a, const float4 b) {  // torchvision.ops.nms, float32
  const float w = fmaxf(fminf(a.z, b.z) - fmaxf(a.x, b.x), 0.f);
  const float hgt = fmaxf(fminf(a.w, b.w) - fmaxf(a.y, b.y), 0.f);
  const float inter = w * hgt;
  const float sa = (a.z - a.x) * (a.w - a.y), sb = (b.z - b.x) * (b.w - b.y);
  return inter / (sa + sb - inter);
}

__global__ __launch_bounds__(kPfThreads) void pseudo_box_filter_kernel(
    const int32_t *__restrict__ rects, const unsigned char *__restrict__ valid, const float *__restrict__ objectness,
    const float *__restrict__ pred_corners, const float *__restrict__ gt_corners, const float *__restrict__ gt_present,
    float nms_iou, float gt_iou, float min_objectness, int32_t *__restrict__ sel, int32_t *__restrict__ count, int k,
    int kpow2, int g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4 *s_box = reinterpret_cast<float4 *>(smem);                                   // [k] (x1,y1,x2,y2)
  unsigned long long *s_key = reinterpret_cast<unsigned long long *>(s_box + k);      // [kpow2]
  float *s_score = reinterpret_cast<float *>(s_key + kpow2);                          // [k]
  float *s_gt = s_score + k;                                                          // [g][6] lo xyz, hi xyz
  unsigned char *s_alive = reinterpret_cast<unsigned char *>(s_gt + 6 * g);           // [kpow2] by sorted position
  unsigned char *s_pass = s_alive + kpow2;                                            // [kpow2] by sorted position
  unsigned char *s_gtok = s_pass + kpow2;                                             // [g]
  const int scene = blockIdx.x, tid = threadIdx.x;
  const int32_t *rc = rects + static_cast<size_t>(scene) * k * 4;
  const unsigned char *vd = valid + static_cast<size_t>(scene) * k;
  const float *ob = objectness + static_cast<size_t>(scene) * k;
  const float *pc = pred_corners + static_cast<size_t>(scene) * k * 24;

  for (int j = tid; j < k; j += kPfThreads) {
    const bool ok = vd[j] != 0;
    // a given-up proposal: score -1, rectangle (0,0,2,2) (:1312-1342); else (ymin, xmin, ymax, xmax) (:1343-1346)
    s_box[j] = ok ? make_float4(static_cast<float>(rc[j * 4 + 1]), static_cast<float>(rc[j * 4 + 0]),
                                static_cast<float>(rc[j * 4 + 3]), static_cast<float>(rc[j * 4 + 2]))
                  : make_float4(0.f, 0.f, 2.f, 2.f);
    s_score[j] = ok ? ob[j] : -1.f;
    sel[static_cast<size_t>(scene) * k + j] = -1;
  }
  for (int q = tid; q < g; q += kPfThreads) {
    const float *c = gt_corners + (static_cast<size_t>(scene) * g + q) * 24;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      float lo = c[a], hi = lo;
#pragma unroll
      for (int p = 1; p < 8; ++p) {
        lo = fminf(lo, c[p * 3 + a]);
        hi = fmaxf(hi, c[p * 3 + a]);
      }
      s_gt[q * 6 + a] = lo;
      s_gt[q * 6 + 3 + a] = hi;
    }
    s_gtok[q] = gt_present[static_cast<size_t>(scene) * g + q] > 0.f;  // torch.nonzero(gt_box_present), :1352
  }
  __syncthreads();
  // descending (score, then ascending index) as one unsigned key
  for (int j = tid; j < kpow2; j += kPfThreads) {
    unsigned long long key = 0ull;
    if (j < k) {
      unsigned u = __float_as_uint(s_score[j]);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      key = (static_cast<unsigned long long>(u) << 32) | static_cast<unsigned>(0x7fffffff - j);
      if (key == 0ull) key = 1ull;
    }
    s_key[j] = key;
  }
  __syncthreads();
  for (int size = 2; size <= kpow2; size <<= 1) {  // bitonic sort, descending
    for (int strd = size >> 1; strd > 0; strd >>= 1) {
      for (int t = tid; t < kpow2 / 2; t += kPfThreads) {
        const int lo = 2 * t - (t & (strd - 1)), hi = lo + strd;
        const bool desc = (lo & size) == 0;
        const unsigned long long a = s_key[lo], b = s_key[hi];
        if ((a < b) == desc) {
          s_key[lo] = b;
          s_key[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int j = tid; j < kpow2; j += kPfThreads) s_alive[j] = j < k;
  __syncthreads();
  auto index_at = [&](int pos) { return 0x7fffffff - static_cast<int>(s_key[pos] & 0xffffffffu); };
  for (int i = 0; i < k; ++i) {  // greedy sweep in sorted order (uniform control flow)
    if (!s_alive[i]) continue;
    const float4 bi = s_box[index_at(i)];
    for (int j = i + 1 + tid; j < k; j += kPfThreads) {
      if (!s_alive[j]) continue;
      if (iou2d(bi, s_box[index_at(j)]) > nms_iou) s_alive[j] = 0;
    }
    __syncthreads();
  }
  // survivors: 3-D IoU of the axis-aligned extents against every present ground-truth box (cal_iou, :868-899),
  // validity, objectness threshold
  for (int i = tid; i < k; i += kPfThreads) {
    bool pass = false;
    if (s_alive[i]) {
      const int j = index_at(i);
      pass = vd[j] != 0 && !(s_score[j] < min_objectness);
      if (pass) {
        float lo[3], hi[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          lo[a] = hi[a] = pc[j * 24 + a];
#pragma unroll
          for (int p = 1; p < 8; ++p) {
            lo[a] = fminf(lo[a], pc[j * 24 + p * 3 + a]);
            hi[a] = fmaxf(hi[a], pc[j * 24 + p * 3 + a]);
          }
        }
        const float vol1 = (hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]);
        for (int q = 0; q < g && pass; ++q) {
          if (!s_gtok[q]) continue;
          const float *gb = s_gt + q * 6;
          const float l = fmaxf(0.f, fminf(hi[0], gb[3]) - fmaxf(lo[0], gb[0]));
          const float wd = fmaxf(0.f, fminf(hi[1], gb[4]) - fmaxf(lo[1], gb[1]));
          const float ht = fmaxf(0.f, fminf(hi[2], gb[5]) - fmaxf(lo[2], gb[2]));
          const float vol2 = (gb[3] - gb[0]) * (gb[4] - gb[1]) * (gb[5] - gb[2]);
          const float inter = l * wd * ht;
          if (inter / (vol1 + vol2 - inter) > gt_iou) pass = false;
        }
      }
    }
    s_pass[i] = pass;
  }
  __syncthreads();
  if (tid == 0) {
    int n = 0;
    for (int i = 0; i < k; ++i)
      if (s_pass[i]) sel[static_cast<size_t>(scene) * k + n++] = index_at(i);
    count[scene] = n;
  }
}

}  // namespace
}  // namespace coda

CODA_API int coda_clip_weak_labels_f32(const float *emb, long long emb_stride, const float *text, const float *scale,
                                       const float *row_mask, float *score, int64_t *label, int rows, int rows_per_set,
                                       int nsets, int ncls, int d, void *stream) {
  using namespace coda;
  if (rows < 0 || ncls < 1 || nsets < 1 || d != kWlD || rows_per_set < 1 || emb_stride < d || (emb_stride & 3)) return CODA_EINVAL;
  if (rows == 0) return CODA_OK;
  if (!emb || !text || !scale || !score || !label) return CODA_EINVAL;
  if ((reinterpret_cast<uintptr_t>(emb) | reinterpret_cast<uintptr_t>(text)) & 15) return CODA_EINVAL;
  if (nsets > 1 && (rows_per_set % 32 != 0 || static_cast<long long>(rows_per_set) * nsets < rows)) return CODA_EINVAL;
  if (nsets == 1) rows_per_set = rows > 32 ? ((rows + 31) / 32) * 32 : 32;  // every tile reads set 0
  auto kern = clip_weak_labels_kernel;
  if (int st = raise_dynamic_lds(kern, kWlLds); st != CODA_OK) return st;
  clear_sticky_error();
  hipLaunchKernelGGL(kern, dim3((rows + 31) / 32), dim3(kWlThreads), kWlLds, static_cast<hipStream_t>(stream), emb,
                     emb_stride, text, static_cast<long long>(ncls) * kWlD, scale, row_mask, score, label, rows,
                     rows_per_set, ncls);
  return launch_status();
}

CODA_API int coda_pseudo_box_filter_f32(const int32_t *rects, const unsigned char *valid, const float *objectness,
                                        const float *pred_corners, const float *gt_corners, const float *gt_present,
                                        float nms_iou, float gt_iou, float min_objectness, int32_t *sel, int32_t *count,
                                        int b, int k, int g, void *stream) {
  using namespace coda;
  if (b < 0 || k < 0 || g < 0) return CODA_EINVAL;
  if (b == 0) return CODA_OK;
  if (!count || (k > 0 && (!rects || !valid || !objectness || !pred_corners || !sel)) || (g > 0 && (!gt_corners || !gt_present)))
    return CODA_EINVAL;
  if (k > kPfMaxK || g > kPfMaxG) return CODA_ENOSPC;
  int kpow2 = 2;
  while (kpow2 < k) kpow2 <<= 1;
  const size_t lds = sizeof(float4) * k + sizeof(unsigned long long) * kpow2 + sizeof(float) * (k + 6 * g) + 2 * kpow2 + g + 16;
  auto kern = pseudo_box_filter_kernel;
  if (int st = raise_dynamic_lds(kern, lds); st != CODA_OK) return st;
  clear_sticky_error();
  hipLaunchKernelGGL(kern, dim3(b), dim3(kPfThreads), lds, static_cast<hipStream_t>(stream), rects, valid, objectness,
                     pred_corners, gt_corners, gt_present, nms_iou, gt_iou, min_objectness, sel, count, k, kpow2, g);
  return launch_status();
}
