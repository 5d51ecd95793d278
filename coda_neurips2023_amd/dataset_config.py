"""Minimal dataset-config stand-in for the hot path.

The model only touches four members of the reference's dataset config objects
(datasets/sunrgbd_anonymous_aligned_image.py:86-298): ``num_semcls``,
``num_angle_bin``, ``image_size`` and the two corner builders.  Callers that
have the reference's datasets package pass its config object instead; tests and
``bench.py`` (no dataset here) use this one.
"""
from . import box_util


class HotPathDatasetConfig:
    # the corner builders below are the SUN-RGBD / ScanNet ones (flip to camera + get_3d_box_batch_tensor, and
    # get_3d_box_batch_tensor_xyz): lets the model use the fused decoder (box_decode.py).  A reference dataset
    # config object can opt in by setting the same attribute.
    standard_corner_builders = True

    def __init__(self, num_semcls=1, num_angle_bin=12, max_num_obj=64, image_size=(730, 531)):
        self.num_semcls = num_semcls
        self.num_angle_bin = num_angle_bin
        self.max_num_obj = max_num_obj
        self.image_size = list(image_size)

    def box_parametrization_to_corners(self, box_center_unnorm, box_size, box_angle):
        box_center_upright = box_util.flip_axis_to_camera_tensor(box_center_unnorm)
        return box_util.get_3d_box_batch_tensor(box_size, box_angle, box_center_upright)

    def box_parametrization_to_corners_xyz(self, box_center_unnorm, box_size, box_angle):
        return box_util.get_3d_box_batch_tensor_xyz(box_size, box_angle, box_center_unnorm)
