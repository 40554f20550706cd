#!/usr/bin/env python3
"""Round-2 golden fixtures (tests/golden/reference_cpu_r2.npz), produced like make_golden.py by running the
REFERENCE's own Python on CPU in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r2.py

  * farthest_point_sample_np with its RANDOM start (pointnet2_utils.py:103-118): torch.manual_seed(s) before the call
    fixes torch.randint, so the product -- which draws the start with the same call -- can be compared index for index,
    on clouds with exact distance ties and duplicated vertices too;
  * PointNetSetAbstraction / PointNetSetAbstractionMsg in eval mode at shapes where the fused set-abstraction path
    of the product is taken on its own (wide feature rows, single- and multi-layer shared MLPs);
  * gen_utils.read_txt_obj_ls's pure-python OBJ parsing (gen_utils.py:201-226) on a synthetic mesh file and the
    (24000, 7) float64 array preprocess_data.py:48-58 saves (vertex normals come from open3d in the reference, which
    is not installed here: the normals column block is produced by the oracle's restatement and marked unpinned).
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from make_golden import check, load_reference, ref_fps  # noqa: E402
from oracle import cpu as O  # noqa: E402
from toothgroupnetwork_amd import synth  # noqa: E402


def fps_random_start(R, out):
    clouds = {
        "arch": np.stack([synth.arch_cloud(2000, seed=s, with_normals=False) for s in (10, 11, 12)]),
        "lattice": np.stack([synth.lattice_cloud(9, dup=71, seed=s) for s in (4, 5)]),   # exact ties + duplicates
    }
    dup = synth.arch_cloud(1500, seed=13, with_normals=False)
    dup[700:900] = dup[100:300]                                                          # duplicated vertices
    clouds["dupverts"] = dup[None]
    for name, xyz in clouds.items():
        for seed, npoint in ((1, 256), (2, 64)):
            torch.manual_seed(seed)
            idx = R.farthest_point_sample_np(np.ascontiguousarray(xyz, dtype=np.float32), npoint)
            torch.manual_seed(seed)
            start = torch.randint(0, xyz.shape[1], (xyz.shape[0],), dtype=torch.long).numpy()
            assert np.array_equal(idx[:, 0], start)
            # the oracle restates it as: canonical FPS of [p_start, p_0, ..., p_{N-1}]
            padded = np.concatenate([xyz[np.arange(xyz.shape[0]), start][:, None], xyz], axis=1)
            ora = O.farthest_point_sample(padded, npoint) - 1
            ora[:, 0] = start
            ora = np.maximum(ora, 0)
            check(f"fps_np_{name}_seed{seed}", ora, idx)
            out[f"fpsnp_{name}_{seed}_idx"] = idx.astype(np.int32)
        out[f"fpsnp_{name}_xyz"] = xyz.astype(np.float32)
    # an exhausted cloud: 5 distinct points, 12 samples -> torch.max returns index 0 once every distance is 0
    few = np.repeat(synth.uniform_cloud(5, seed=3), 3, axis=0)[None]
    torch.manual_seed(7)
    out["fpsnp_few_idx"] = R.farthest_point_sample_np(few, 12).astype(np.int32)
    out["fpsnp_few_xyz"] = few


def main():
    torch.set_num_threads(1)
    R = load_reference()
    out = {}
    fps_random_start(R, out)
    try:
        from make_golden_r2_sa import sa_fixtures
        sa_fixtures(R, out)
    except ImportError:
        pass
    try:
        from make_golden_r2_pt import pt_fixtures
        pt_fixtures(out)
    except ImportError:
        pass
    try:
        from make_golden_r2_io import io_fixtures
        io_fixtures(out)
    except ImportError:
        pass
    np.savez_compressed(os.path.join(HERE, "reference_cpu_r2.npz"), **out)
    sz = os.path.getsize(os.path.join(HERE, "reference_cpu_r2.npz"))
    print(f"wrote tests/golden/reference_cpu_r2.npz ({sz / 1e6:.2f} MB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
