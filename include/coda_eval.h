/*
 * coda_eval.h -- C ABI of the evaluation post-processing: the two loops of
 * utils/ap_calculator.py's parse_predictions* that decide which proposals survive -- "remove empty boxes"
 * (:845-871 / :106-124, a scipy Delaunay triangulation + find_simplex per proposal on the host) and the
 * score-ordered greedy NMS on axis-aligned extents (utils/nms.py:46-175 called from :873-968 / :126-217).
 *
 * coda_box_point_count_f32
 *   counts[b][k] = number of points of scene b inside proposal box k.  corners (b,k,8,3) float32 are the
 *   upright-camera corners the model emits (outputs["box_corners"], corner order of utils/box_util.py:383-416:
 *   0..3 one face in ring order, 4..7 the opposite face, i+4 across from i); points (b,n,point_stride) float32
 *   hold depth-frame xyz in their first three columns (batch_data_label["point_clouds"]).  A point is inside when
 *   its camera-frame image (x, -z, y) satisfies 0 <= (p - c0).e <= e.e for the three edges e = c1-c0, c3-c0,
 *   c4-c0 -- the convex hull of the eight corners that the reference tests with in_hull (utils/box_util.py:22-31);
 *   points exactly on a face are a measure-zero difference (qhull applies a tolerance there).
 *
 * coda_nms_f32
 *   One scene per workgroup.  extents = min / max of the corners per axis, as float64 (the reference fills a
 *   float64 array from the float32 corners, :889-905); candidates = boxes with nonempty[b][k] != 0, or, when a
 *   scene has none, its box with the highest objectness (:869-870; nonempty may be NULL = all boxes);
 *   candidates are visited by decreasing score, a visited box is kept and suppresses every later candidate whose
 *   overlap with it exceeds nms_iou:
 *     mode 0 (use_3d_nms false): 2-D boxes (x, z), utils/nms.py:46-79
 *     mode 1: 3-D boxes, :82-121            mode 2 (cls_nms): 3-D, only boxes of the same class suppress, :124-175
 *     overlap = inter / (vol_i + vol_j - inter), or inter / vol_j when old_type != 0; float64, one rounding per
 *     operation in numpy's order.
 *   Equal scores: the reference's np.argsort (introsort) leaves their order unspecified; here the box with the
 *   larger index is visited first.  keep (b,k) uint8 receives 1 for kept boxes.  k <= 2048.
 */
#ifndef CODA_EVAL_H
#define CODA_EVAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int coda_box_point_count_f32(const float *corners, const float *points, int32_t *counts, int b, int k, int n,
                             int point_stride, void *stream);

int coda_nms_f32(const float *corners, const float *scores, const int32_t *classes, const unsigned char *nonempty,
                 unsigned char *keep, int b, int k, int mode, double nms_iou, int old_type, void *stream);

#ifdef __cplusplus
}
#endif

#endif /* CODA_EVAL_H */
