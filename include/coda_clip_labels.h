/* coda_clip_labels.h -- C ABI of the label side of CoDA's image branch (second half).
 *
 * The reference derives two kinds of labels from CLIP image embeddings of cropped proposals, both as chains of small
 * torch launches inside per-scene Python loops:
 *   - "weak labels" for the alignment loss: softmax over the class prompts of (unit-norm image embedding . text
 *     embedding) * temperature, its arg-max and max probability per proposal
 *     (models/model_3detr.py:1153-1172 and :1614-1631; the same expression classifies novel-box candidates,
 *     :1110-1123 and :1497-1505);
 *   - the stage-2 pseudo-label candidates: torchvision.ops.nms on the proposals' projected 2-D rectangles in
 *     objectness order, minus the proposals whose axis-aligned 3-D extent overlaps a ground-truth box (IoU > 0.25,
 *     cal_iou :868-899), minus those below the objectness threshold (:1305-1426).
 * Here each is one launch for the whole batch.  Same conventions as coda_pointnet2.h: raw device pointers, a
 * hipStream_t, 0 / CODA_E* / hipError_t return values; built into libcoda_hip.so.
 */
#ifndef CODA_CLIP_LABELS_H
#define CODA_CLIP_LABELS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* score[r] = max_c softmax_c( <e_r / (|e_r| + 1e-32), text[set(r)][c]> * scale ),  label[r] = its arg-max (first
 * maximum), for r < rows.  One workgroup per 32 rows; the products run on v_mfma_f32_32x32x2_f32 (fp32 in, fp32
 * accumulate), the soft-max state is kept online per row, nothing but the two result vectors is written.
 *   emb      (rows, 512) float32 row-major, row stride `emb_stride` elements.
 *   text     (nsets, ncls, 512) float32; row r uses set r / rows_per_set (nsets == 1: rows_per_set >= rows).
 *            rows_per_set must be a multiple of 32 when there is more than one set.
 *   scale    device pointer to ONE float32 (the model's clipped exp(logit_scale), a 0-dim tensor).
 *   row_mask (rows) float32 or NULL: where row_mask[r] < 1 the score is written as 0 (:1169-1170, :1627-1629).
 *   score    (rows) float32, label (rows) int64.
 * Returns CODA_EINVAL for d != 512, ncls < 1 or a rows_per_set that cuts a 32-row tile. */
int coda_clip_weak_labels_f32(const float *emb, long long emb_stride, const float *text, const float *scale,
                              const float *row_mask, float *score, int64_t *label, int rows, int rows_per_set, int nsets,
                              int ncls, int d, void *stream);

/* Stage-2 pseudo-label candidates of every scene (models/model_3detr.py:1305-1426), one workgroup per scene.
 *   rects (b,k,4) int32 [xmin,ymin,xmax,ymax] and valid (b,k) uint8 from coda_project_box_rects_f64; a proposal with
 *   valid == 0 gets score -1 and the rectangle (0,0,2,2) like the reference's "give the box up" branches.
 *   objectness (b,k) float32; pred_corners (b,k,8,3), gt_corners (b,g,8,3) float32 (camera frame, "box_corners");
 *   gt_present (b,g) float32 (> 0: a real box).
 * Greedy 2-D NMS in descending (score, then ascending index) order: rectangle j is suppressed by a kept rectangle i
 * when inter / (area_i + area_j - inter) > nms_iou (float32 arithmetic, no +1: torchvision.ops.nms); a survivor is
 * dropped when the IoU of its axis-aligned 3-D extent with any present ground-truth box's exceeds gt_iou, when it
 * is invalid, or when its score < min_objectness.
 *   sel (b,k) int32: the remaining proposals of a scene in NMS (= descending objectness) order, -1 padded;
 *   count (b) int32.
 * Limits: k <= 1024, g <= 128 (CODA_ENOSPC otherwise). */
int coda_pseudo_box_filter_f32(const int32_t *rects, const unsigned char *valid, const float *objectness,
                               const float *pred_corners, const float *gt_corners, const float *gt_present, float nms_iou,
                               float gt_iou, float min_objectness, int32_t *sel, int32_t *count, int b, int k, int g,
                               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_CLIP_LABELS_H */
