"""toothgroupnetwork_amd -- MI355X (gfx950) implementation of ToothGroupNetwork's point-cloud
sampling / grouping hot path (FPS, ball query / kNN, three_nn + interpolate, gather / group and the
Point-Transformer subtraction / aggregation operators).

    toothgroupnetwork_amd.pointops          mirror of external_libs/pointops/functions/pointops.py
    toothgroupnetwork_amd.pointnet2_utils   mirror of external_libs/pointnet2_utils/pointnet2_utils.py
    toothgroupnetwork_amd.resample          gen_utils.fps / resample_pcd (preprocess_data.py's FPS)
    toothgroupnetwork_amd.sharding          one-process-per-GPU mesh sharding + RCCL metric gather
    toothgroupnetwork_amd.csrc              HIP kernels behind the C ABI of include/tgn_pointops.h

The HIP library is the only compute path; nothing here falls back to the CPU.
"""
__version__ = "0.1.0"
