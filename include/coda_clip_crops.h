/*
 * coda_clip_crops.h -- C ABI of the image side of the CLIP distillation branch up to the tower's input
 * (SURVEY.md 8f rank 2): everything get_predicted_box_clip_embedding (models/model_3detr.py:902-1086) does per
 * proposal in a Python loop with int(tensor) read-backs -- un-augment the predicted corners, project them into the
 * image (datasets/sunrgbd_utils.py:611-635), take the 2-D extent, cut the crop, pad it to a white square, resize
 * it to the tower's resolution and normalise it -- as two launches for all scenes and proposals.  The frozen CLIP
 * tower itself stays the deployment's module (weights are not part of this package).
 *
 * coda_project_box_rects_f64
 *   corners (b,k,8,3) float32: outputs["box_corners_xyz"]; sizes (b,k,3) float32: outputs["size_unnormalized"].
 *   Per scene (float64, as the reference promotes): scale (b,3), rot (b,3,3), flip (b), zx_flip (b) or NULL,
 *   K (b,3,3), Rtilt (b,3,3), ori_wh (b,2) = [ori_width, ori_height], offset_xy (b,2) = [y_offset, x_offset] (the
 *   reference adds y_offset to u and x_offset to v), image_flip (b), flip_length (b).
 *   uv (b,k,8,2), depth (b,k,8) float64 (may be NULL): :907-965.
 *   rects (b,k,4) int32 = [xmin, ymin, xmax, ymax] (int() of the min / max over the 8 corners, :1021-1024);
 *   valid (b,k) uint8 = extent positive in both directions, no corner behind the camera, box not of zero size
 *   (:1015-1036: the three `continue`s of the loop).
 *
 * coda_crop_resize_f32
 *   images (b,h,w,3) uint8 RGB; sel (b,s) int32 proposal indices; rects / valid as above.
 *   out (b*s, 3, res, res) float32: crop -> centred on a white square of the larger edge (:1039-1062) -> bicubic
 *   resize (torchvision 0.9.1's Resize on a uint8 tensor = torch.nn.functional.interpolate(mode="bicubic",
 *   align_corners=False), then clamp to [0,255], round, cast) -> / 255 -> (x - mean) / std with CLIP's constants
 *   (CLIP/clip/clip.py:95-101).  Entries of invalid proposals are filled with the normalised white square (their
 *   embeddings are masked out by the caller).
 */
#ifndef CODA_CLIP_CROPS_H
#define CODA_CLIP_CROPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int coda_project_box_rects_f64(const float *corners, const float *sizes, const double *scale, const double *rot,
                               const double *flip, const double *zx_flip, const double *kmat, const double *rtilt,
                               const double *ori_wh, const double *offset_xy, const double *image_flip,
                               const double *flip_length, double *uv, double *depth, int32_t *rects,
                               unsigned char *valid, int b, int k, void *stream);

int coda_crop_resize_f32(const unsigned char *images, const int32_t *sel, const int32_t *rects,
                         const unsigned char *valid, float *out, int b, int h, int w, int k, int s, int res,
                         void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CODA_CLIP_CROPS_H */
