/*
 * coda_box_ops.h -- C ABI of the box geometry: the matcher's gIoU and the box decoder
 * (8f rank 3).
 *
 * coda_generalized_box3d_iou_f32 replaces utils/box_util.py:655-745
 * (generalized_box3d_iou_tensor; dispatcher :861-875), which SetCriterion.single_output_forward
 * calls once per decoder layer (criterion.py:1106-1112) for the Hungarian cost matrix
 * (criterion.py:58-66).  The reference evaluates the rotated case with a Python triple loop
 * over (scene, proposal, GT box) and a Sutherland-Hodgman clip per pair on the host; here one
 * thread owns one (scene, proposal, GT) pair and the whole (L*B, K1, K2) matrix is one launch.
 *
 *   corners1 (B, K1, 8, 3), corners2 (B, K2, 8, 3) float32, camera frame (up = -Y), corner order
 *   of get_3d_box_batch_tensor (utils/box_util.py:427-490); nums_k2 (B) int32 = number of real GT
 *   boxes per scene (columns >= nums_k2[b] are 0, :739-744) or NULL.
 *   out (B, K1, K2) float32:
 *     height  = clamp(min(c1[0].y, c2[0].y) - max(c1[4].y, c2[4].y), 0)               :676-678
 *     rect    = corners 3,2,1,0 projected on (x, z)                                    :681-686
 *     area    = rotated ? area(clip(rect1, rect2))   [only where the axis-aligned
 *                         overlap of rect[1] / rect[3] is non-zero, :716-717]
 *                       : that axis-aligned overlap                                    :688-691
 *     giou    = iou - (1 - union / enclosing)  for well-formed pairs, else 0           :733-738
 *   inter_vols_only != 0 returns area * height instead (:729-731); the value 2 additionally skips the reference's
 *   axis-aligned pre-test of the two quadrilaterals (:688-694), i.e. EVERY pair is clipped: the intersection volume
 *   of utils/box_util.py:156-183 (box3d_iou), which the evaluation's eval_det uses.
 *   rotated_k2_limit: < 0 = none.  The reference's Cython fast path (utils/box_intersection.pyx:
 *   181, `K2 = rect2.shape[2]`) only visits GT columns k2 < 4 when boxes are rotated; pass 4 to
 *   reproduce a deployment that built the Cython module.
 * Forward only: the reference asks for gradients only when loss_giou_weight > 0, which no CoDA
 * recipe sets (every scripts/ recipe passes --loss_giou_weight 0).  Conventions as in coda_pointnet2.h.
 */
#ifndef CODA_BOX_OPS_H
#define CODA_BOX_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int coda_generalized_box3d_iou_f32(const float *corners1, const float *corners2,
                                   const int32_t *nums_k2, float *out, int b, int k1, int k2,
                                   int rotated, int inter_vols_only, int rotated_k2_limit,
                                   void *stream);

/* Same, with the rotated / axis-aligned choice read from device memory (one byte, non-zero = rotated): the
 * reference derives it as torch.any(gt_box_angles > 0).item() (criterion.py:1147), a host synchronisation in
 * the middle of the training step; callers that keep the comparison's result on the device pass it here. */
int coda_generalized_box3d_iou_devflag_f32(const float *corners1, const float *corners2, const int32_t *nums_k2,
                                           float *out, int b, int k1, int k2, const unsigned char *rotated_flag,
                                           int inter_vols_only, int rotated_k2_limit, void *stream);

/* ---- box decoding ----------------------------------------------------------
 * Everything `get_box_predictions` computes from the six heads' raw outputs for all decoder layers
 * (models/model_3detr.py:1683-1731 with BoxProcessor :56-127, utils/pc_util.py:38-73 and the corner
 * builders utils/box_util.py:383-490 as the SUN-RGBD / ScanNet dataset configs wire them), one thread
 * per (layer, scene, query) row instead of ~70 element-wise / reduction / tiny-GEMM launches per
 * direction:
 *   center_unnormalized = query_xyz + (sigmoid(center_raw) - 0.5)
 *   center_normalized   = (center_unnormalized - dims_min) / (dims_max - dims_min)
 *   size_normalized     = sigmoid(size_raw);  size_unnormalized = size_normalized * max(dims_max - dims_min, 0.1)
 *   angle_residual      = angle_residual_normalized * (pi / nbin)
 *   angle_continuous    = (2 pi / nbin) * argmax(angle_logits) + angle_residual[argmax], minus 2 pi if > pi
 *   box_corners         = camera-frame corners (flip_axis_to_camera + get_3d_box_batch_tensor)
 *   box_corners_xyz     = depth-frame corners (get_3d_box_batch_tensor_xyz)
 *   sem_cls_prob, objectness_prob = softmax(sem_cls_logits)[:-1], 1 - softmax(...)[-1]
 * Inputs are addressed as (layer, scene, query, channel) through element strides (sl, sb, sq; channels
 * contiguous), so the heads' (layer, query, scene) buffers are read in place; `rows` = nl * b * nq;
 * query_xyz (b, nq, 3), dims_min / dims_max (b, 3).  Outputs are dense (nl, b, nq, ...).
 * The backward takes the gradients of the differentiable outputs (any pointer may be NULL = zero) and
 * returns d center_raw, d size_raw (rows x 3) and d angle_residual_normalized (rows x nbin), dense. */
int coda_box_decode_fwd_f32(const float *center_raw, const float *size_raw, const float *angle_logits,
                            const float *angle_res_norm, const float *cls_logits, const long long *strides,
                            const float *query_xyz, const float *dims_min, const float *dims_max, int nl, int b,
                            int nq, int nbin, int ncls1, float *center_norm, float *center_unnorm,
                            float *size_norm, float *size_unnorm, float *angle_residual, float *angle_cont,
                            float *corners, float *corners_xyz, float *cls_prob, float *obj_prob, void *stream);
int coda_box_decode_bwd_f32(const float *center_raw, const float *size_raw, const float *angle_logits,
                            const float *angle_res_norm, const long long *strides, const float *dims_min,
                            const float *dims_max, int nl, int b, int nq, int nbin, const float *g_center_norm,
                            const float *g_center_unnorm, const float *g_size_norm, const float *g_size_unnorm,
                            const float *g_angle_residual, const float *g_angle_cont, const float *g_corners,
                            const float *g_corners_xyz, float *d_center_raw, float *d_size_raw,
                            float *d_angle_res_norm, void *stream);

/* ---- the matcher's cost matrix (criterion.py:50-66) -------------------------------------------------------
 * gIoU as above plus, in the same pass, the L1 centre distance torch.cdist(center_normalized,
 * gt_box_centers_normalized, p=1) (criterion.py:1153) and
 *   cost = w_class * -cls_prob[label of GT] + w_objectness * -objectness + w_center * L1 + w_giou * -gIoU
 * evaluated with one rounding per operation in the reference's left-to-right order.  Replaces one gather, the
 * cdist kernel (750 us for 64 x 256 x 64 pairs on MI355X) and seven elementwise passes over (b, k1, k2).
 *   center1 (b,k1,3), center2 (b,k2,3), cls_prob (b,k1,ncls), labels (b,k2) int64 in [0,ncls), objectness (b,k1);
 *   outputs gious, center_dist, cost: (b,k1,k2) each.  rotated_flag: device byte or NULL (then `rotated`).
 *   labels are trusted to be in range (they index the dataset's class table). */
int coda_matcher_cost_f32(const float *corners1, const float *corners2, const int32_t *nums_k2, const float *center1,
                          const float *center2, const float *cls_prob, const int64_t *labels, const float *objectness,
                          float w_class, float w_objectness, float w_center, float w_giou, float *gious,
                          float *center_dist, float *cost, int b, int k1, int k2, int ncls, int rotated,
                          const unsigned char *rotated_flag, int rotated_k2_limit, void *stream);

/* ---- Hungarian assignment (criterion.py:27-86, Matcher.forward) -----------------------------------------
 * Replaces the per-scene scipy.optimize.linear_sum_assignment calls on the host (one D2H copy of the cost
 * tensor and one host solve per scene and decoder layer): one workgroup per problem solves the rectangular
 * assignment "every real GT box gets a distinct proposal, minimum total cost" with the shortest-augmenting-path
 * algorithm scipy uses (Crouse's rectangular LSAP: dual variables in fp64, one augmentation per GT box), the
 * proposals spread over the threads.
 *   cost (nprob, nq, ngt) float32 (criterion.py:58-66), nactual (nprob) int64 = real GT boxes per problem.
 *   per_prop_gt_inds (nprob, nq) int64: GT index of a matched proposal, 0 otherwise (criterion.py:72-74, 80);
 *   matched_mask (nprob, nq) float32: 1 for matched proposals (:75-77, 81).
 * The optimum is unique unless costs tie exactly; under exact ties the total cost equals scipy's while the
 * chosen proposals may differ (scipy prefers the lowest row index of its formulation, this kernel the lowest
 * proposal index among equal reduced costs).  Non-finite costs (NaN, +-inf: a diverged model): scipy raises
 * ValueError; here the problem's matched_mask is written as NaN for every proposal, which makes the step's loss
 * NaN, the condition engine.py:155-157 stops on.  Limits: nq <= 1024, ngt <= 128, (nq * ngt * 4 + nq * 24) bytes of LDS <= 160 KiB;
 * CODA_ENOSPC otherwise (callers fall back to the host solver). */
int coda_hungarian_f32(const float *cost, const int64_t *nactual, int64_t *per_prop_gt_inds, float *matched_mask,
                       int nprob, int nq, int ngt, void *stream);

/* ---- matched box losses (criterion.py:219-246, 834-900, 1015-1104) -----------------------------------
 * The four box terms SetCriterion evaluates for every decoder layer once the matcher has assigned
 * proposals to ground-truth boxes, as per-row partial sums (rows = nl * b * nq, 5 floats per row):
 *   [0] w[y] * CE(sem_cls_logits, y) * has_object[b]        y = gt label if matched else background (K-1)
 *   [1] matched * CE(angle_logits, gt_angle_class)           [2] matched * huber(res_norm[gt bin] - gt_res_norm)
 *   [3] matched * |center_norm - gt_center|_1                [4] matched * |size_norm - gt_size|_1
 * with the ground-truth row of a proposal = gt_inds[row].  The caller sums the rows of a layer and applies
 * the reference's normalisers (#scenes-with-objects * nq, num_boxes) and weights; the backward takes the
 * resulting per-layer scalars g (nl x 5) and writes the gradients of the five inputs (dense, (rows, C)).
 * Inputs are addressed through element strides like coda_box_decode_fwd_f32 (5 x 3: sem_cls_logits,
 * angle_logits, angle_res_norm, center_norm, size_norm).  gt_res_norm is already divided by pi / nbin. */
int coda_box_loss_fwd_f32(const float *sem_logits, const float *angle_logits, const float *angle_res_norm,
                          const float *center_norm, const float *size_norm, const long long *strides,
                          const int64_t *gt_inds, const float *matched, const int64_t *gt_sem_label,
                          const int64_t *gt_angle_class, const float *gt_res_norm, const float *gt_center,
                          const float *gt_size, const float *has_object, const float *sem_class_weight, int nl,
                          int b, int nq, int ngt, int nsem, int nbin, float *partial, void *stream);
int coda_box_loss_bwd_f32(const float *sem_logits, const float *angle_logits, const float *angle_res_norm,
                          const float *center_norm, const float *size_norm, const long long *strides,
                          const int64_t *gt_inds, const float *matched, const int64_t *gt_sem_label,
                          const int64_t *gt_angle_class, const float *gt_res_norm, const float *gt_center,
                          const float *gt_size, const float *has_object, const float *sem_class_weight, int nl,
                          int b, int nq, int ngt, int nsem, int nbin, const float *g, float *d_sem, float *d_angle,
                          float *d_res, float *d_center, float *d_size, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_BOX_OPS_H */
