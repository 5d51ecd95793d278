"""Host-side mirror of the reference's ``pointnet2`` package
(third_party_pointnet2/pointnet2): ``_ext`` (operator module over the C ABI),
``pointnet2_utils`` (autograd functions, QueryAndGroup), ``pytorch_utils``
(SharedMLP) and ``pointnet2_modules`` (PointnetSAModuleVotes, PointnetFPModule).
"""
from . import _ext  # noqa: F401
