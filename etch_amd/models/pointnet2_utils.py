"""MI355X counterpart of /root/reference/src/models/pointnet2_utils.py: square_distance, index_points, PointFeatPropagation."""
from .. import ops
from ..vgtk_functional import index_points, square_distance  # noqa: F401  (pointnet2_utils.py:4-43, HIP kernels)


def PointFeatPropagation(xyz1, xyz2, points2):
    """pointnet2_utils.py:45-74.  xyz1 [B,3,N], xyz2 [B,3,S], points2 [B,D,S] with D = C*60 flattened as c*60+a
    (models_pointcloud.py:161) -> [B,N,D].  Reference-layout wrapper; the model itself uses
    `propagate_cl` to stay channels-last."""
    B, D, S = points2.shape
    C = D // 60
    feats_cl = points2.view(B, C, 60, S).permute(0, 3, 2, 1).contiguous()
    out_cl, _ = propagate_cl(xyz1.permute(0, 2, 1).contiguous(), xyz2.contiguous(), feats_cl)
    return out_cl.permute(0, 1, 3, 2).reshape(B, -1, D)


def propagate_cl(hitpts_bn3, xyz2_b3s, feats_cl, order=None):
    """-> (point_equiv_feat (B,N,60,C) channels-last, point_inv_feat (B,N,C)).  order: spatial processing order of the N points."""
    idx, w = ops.prop3nn(hitpts_bn3, xyz2_b3s)
    return ops.prop_interp(feats_cl, idx, w, order=order)
