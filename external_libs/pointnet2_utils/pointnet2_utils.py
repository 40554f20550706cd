"""Drop-in for the reference's ``external_libs/pointnet2_utils/pointnet2_utils.py``.

Importers in the reference (unchanged): ``models/modules/pointnet_pp.py:3``, ``tsg_centroid_module.py:3``,
``tsg_seg_module.py:3``, ``tsegnet.py:8``, ``models/tsegnet_model.py:5``, ``models/tgn_loss.py:4``,
``models/tsg_loss.py:2``, ``ops_utils.py:5``.  Implementation: ``toothgroupnetwork_amd.pointnet2_utils``.
"""
from toothgroupnetwork_amd.pointnet2_utils import (  # noqa: F401
    PointNetFeaturePropagation,
    PointNetSetAbstraction,
    PointNetSetAbstractionMsg,
    farthest_point_sample,
    farthest_point_sample_np,
    index_points,
    pc_normalize,
    query_ball_point,
    sample_and_group,
    sample_and_group_all,
    square_distance,
    timeit,
)
