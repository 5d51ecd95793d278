/*
 * coda_box_ops.h -- C ABI of the box geometry the matcher needs.
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
 *   inter_vols_only != 0 returns area * height instead (:729-731).
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

#ifdef __cplusplus
}
#endif
#endif /* CODA_BOX_OPS_H */
