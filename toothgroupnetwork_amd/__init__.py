"""toothgroupnetwork_amd -- MI355X (gfx950) implementation of ToothGroupNetwork's point-cloud
sampling / grouping hot path (FPS, ball query / kNN, three_nn + interpolate, gather / group and the
Point-Transformer subtraction / aggregation operators).

    toothgroupnetwork_amd.pointops          mirror of external_libs/pointops/functions/pointops.py
    toothgroupnetwork_amd.pointnet2_utils   mirror of external_libs/pointnet2_utils/pointnet2_utils.py
    toothgroupnetwork_amd.resample          gen_utils.fps / resample_pcd (preprocess_data.py's FPS)
    toothgroupnetwork_amd.preprocess        preprocess_data.py / gen_utils.read_txt_obj_ls (OBJ -> 24 000-point .npy)
    toothgroupnetwork_amd.point_transformer mirrors of the cbl_point_transformer blocks (fused eval paths)
    toothgroupnetwork_amd.nets              whole-network mirrors (state_dict-compatible with the reference's modules)
    toothgroupnetwork_amd.sharding          one-process-per-GPU mesh sharding + the RCCL metric gather
    toothgroupnetwork_amd.launch            starting the ranks of a multi-GPU run; rank / device / backend records
    toothgroupnetwork_amd.eval_sharded      trainer.py's per-scan validation loop over ranks (LossMeter sums gathered once)
    toothgroupnetwork_amd.hotpath           the benchmarked launch plan (FPS -> ball query -> group over three HIP streams)
    toothgroupnetwork_amd.config            the Python-side switches, one object
    toothgroupnetwork_amd.csrc              HIP kernels behind the C ABI of include/tgn_pointops.h

The HIP library is the only compute path; nothing here falls back to the CPU.
"""
__version__ = "0.5.0"
