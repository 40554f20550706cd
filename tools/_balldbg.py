import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import cpu as O
from toothgroupnetwork_amd import pointnet2_utils as U, synth
dev = torch.device("cuda")
xyz = np.stack([synth.uniform_cloud(5000, s) for s in (0, 1, 2)])
q = xyz[:, ::7][:, :300]
for radius, ns in [(0.05, 32), (0.1, 32), (0.2, 64), (2.5, 8)]:
    got = U.query_ball_point(radius, ns, torch.from_numpy(xyz).to(dev), torch.from_numpy(q).to(dev)).cpu().numpy()
    want = O.query_ball_point(radius, ns, xyz, q)
    bad = np.argwhere((got != want).any(-1))
    print(radius, ns, "bad rows", len(bad), bad[:6].tolist())
    for b, s in bad[:3]:
        print("  got ", got[b, s].tolist()); print("  want", want[b, s].tolist())
        d = ((xyz[b] - q[b, s]) ** 2).sum(-1); print("  hits", int((d <= radius ** 2).sum()))
