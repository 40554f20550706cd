"""Set-abstraction module fixtures of make_golden_r2.py: the reference's PointNetSetAbstraction[Msg] in eval mode at
shapes where the product's fused path is taken on its own (no monkey-patching in the tests):
  ssg_wide   PointNetSetAbstraction, single layer, 64 feature channels  -> per-point transform + gather-max
  ssg_narrow PointNetSetAbstraction, single layer, 6 feature channels   -> direct kernel (gather -> matrix cores -> max)
  msg        PointNetSetAbstractionMsg, one single-layer and one two-layer branch (the latter above the pay-off point)
Weights and BatchNorm statistics are seeded; the state_dicts go to module_weights_r2.pt."""
import os

import numpy as np
import torch

from make_golden import ref_fps
from toothgroupnetwork_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _randomise_bn(mod):
    for m_ in mod.modules():
        if isinstance(m_, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m_.running_mean.normal_(0, 0.2)
            m_.running_var.uniform_(0.5, 2.0)
            m_.weight.data.uniform_(0.5, 1.5)
            m_.bias.data.normal_(0, 0.1)


def sa_fixtures(R, out):
    torch.manual_seed(4321)
    R.farthest_point_sample = lambda x, n: torch.from_numpy(ref_fps(R, x.numpy(), n))
    B, N, D = 2, 1500, 64
    pts6 = np.stack([synth.arch_cloud(N, seed=s) for s in (30, 31)])                 # (B,N,6): xyz + normals
    xyz = np.ascontiguousarray(pts6[:, :, :3])
    feat = np.random.default_rng(5).normal(size=(B, N, D)).astype(np.float32)
    xyz_cf = torch.from_numpy(np.ascontiguousarray(xyz.transpose(0, 2, 1)))
    feat_cf = torch.from_numpy(np.ascontiguousarray(feat.transpose(0, 2, 1)))
    pts6_cf = torch.from_numpy(np.ascontiguousarray(pts6.transpose(0, 2, 1)))
    mods = {
        "ssg_wide": (R.PointNetSetAbstraction(128, 0.25, 32, 3 + D, [96], False), feat_cf),
        "ssg_narrow": (R.PointNetSetAbstraction(96, 0.3, 16, 3 + 6, [64], False), pts6_cf),
        "msg": (R.PointNetSetAbstractionMsg(128, [0.2, 0.3], [16, 32], D, [[128], [64, 96]]), feat_cf),
    }
    state = {}
    for name, (mod, f_cf) in mods.items():
        mod.eval()
        _randomise_bn(mod)
        with torch.no_grad():
            nx, nf = mod(xyz_cf, f_cf)
        out[f"sa_{name}_xyz"], out[f"sa_{name}_feat"] = nx.numpy(), nf.numpy()
        state[name] = mod.state_dict()
        print(f"  reference module {name}: new_xyz {tuple(nx.shape)} features {tuple(nf.shape)}")
    out["sa_in_xyz_cf"], out["sa_in_feat_cf"], out["sa_in_pts6_cf"] = xyz_cf.numpy(), feat_cf.numpy(), pts6_cf.numpy()
    torch.save(state, os.path.join(HERE, "module_weights_r2.pt"))
