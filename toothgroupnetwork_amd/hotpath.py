"""The headline hot path as one pre-planned launch sequence: for a batch of scans, per set-abstraction
level, FPS -> ball query -> group (BASELINE.json: "24k-pt FPS+ball_query+group fwd").

This is what ``PointNetSetAbstraction[Msg].forward`` does before its shared MLP
(pointnet2_utils.py:276-285 / 160-169), with every buffer allocated once and every kernel launched
through the C ABI on one stream -- no host synchronisation, no allocation inside a step, so a step can
be replayed back-to-back (bench.py) or captured in a hipGraph.

Shape A of SURVEY.md section 8: N=24000, npoint=[4096,1024,256], nsample=32, radii=[0.05,0.1,0.2],
feature widths D=[6,128,512] (level l>0 groups synthetic features of the width the network would carry).
"""
import torch

from . import _lib
from ._lib import check, lib, ptr

SHAPE_A = dict(n=24000, npoint=[4096, 1024, 256], radius=[0.05, 0.1, 0.2], nsample=[32, 32, 32], d=[6, 128, 512])


def algorithmic_bytes(n, npoint, nsample, d, fused=False):
    """Compulsory HBM bytes per scan (inputs read once, outputs written once; fp32 data, int32 indices),
    per level: FPS = 12N + 4S ; ball = 12N + 12S + 4SK ; group = 4SK + 4N(3+D) + 12S + 4SK(3+D)
    (BASELINE.md section 4).  Returns (total, per_level list of dicts)."""
    levels = []
    N = n
    for S, K, D in zip(npoint, nsample, d):
        fps = 12 * N + 4 * S
        ball = 12 * N + 12 * S + 4 * S * K
        group = 4 * S * K + 4 * N * (3 + D) + 12 * S + (0 if fused else 4 * S * K * (3 + D))
        levels.append(dict(fps=fps, ball=ball, group=group, total=fps + ball + group))
        N = S
    return sum(l["total"] for l in levels), levels


class HotPath:
    """Pre-planned FPS -> ball query -> group over `levels` for a fixed batch of B scans."""

    def __init__(self, B, device, shape=SHAPE_A, xyz_first=True, index_dtype=torch.int32):
        self.B, self.device, self.shape = B, device, shape
        self.xyz_first = xyz_first
        self.L = lib()
        self.levels = []
        N = shape["n"]
        f32 = dict(dtype=torch.float32, device=device)
        for S, r, K, D in zip(shape["npoint"], shape["radius"], shape["nsample"], shape["d"]):
            lv = dict(N=N, S=S, K=K, D=D,
                      r2=float(torch.tensor(float(r) ** 2, dtype=torch.float32).item()),
                      fps_idx=torch.empty(B, S, dtype=torch.int32, device=device),
                      new_xyz=torch.empty(B, S, 3, **f32),
                      group_idx=torch.empty(B, S, K, dtype=index_dtype, device=device),
                      grouped=torch.empty(B, S, K, 3 + D, **f32))
            nbytes = int(self.L.tgn_ball_query_workspace_bytes(B, N, S))
            lv["ws_bytes"] = nbytes
            lv["ws"] = torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None
            self.levels.append(lv)
            N = S
        self.idx64 = int(index_dtype == torch.int64)
        self.events = None

    def enable_kernel_timing(self, steps):
        """HIP events on the launch stream around each kernel class (start/stop per step)."""
        names = [f"{k}_l{i + 1}" for i in range(len(self.levels)) for k in ("fps", "ball", "group")]
        self.events = {n: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                           for _ in range(steps)] for n in names}
        self._step = 0

    def kernel_times_ms(self):
        return {n: [a.elapsed_time(b) for a, b in evs[:self._step]] for n, evs in self.events.items()}

    def _timed(self, name, fn):
        if self.events is None or self._step >= len(self.events[name]):
            return fn()
        a, b = self.events[name][self._step]
        a.record()
        fn()
        b.record()

    def run(self, xyz, feats):
        """xyz: (B, N, 3) fp32 contiguous; feats: list of per-level feature tensors (B, N_l, D_l).
        Results land in self.levels[l]['grouped'] etc.  Asynchronous on the current stream."""
        L, st = self.L, _lib.stream()
        cur_xyz = xyz
        for i, lv in enumerate(self.levels):
            B, N, S, K, D = self.B, lv["N"], lv["S"], lv["K"], lv["D"]
            pts = feats[i]
            self._timed(f"fps_l{i + 1}", lambda: check(L.tgn_furthestsampling_dense(
                B, N, S, ptr(cur_xyz), None, ptr(lv["fps_idx"]), ptr(lv["new_xyz"]), _lib.FPS_LOCAL_INDEX, st), "fps"))
            self._timed(f"ball_l{i + 1}", lambda: check(L.tgn_ball_query(
                B, N, S, K, lv["r2"], ptr(cur_xyz), ptr(lv["new_xyz"]), ptr(lv["group_idx"]), self.idx64,
                ptr(lv["ws"]), lv["ws_bytes"], st), "ball_query"))
            self._timed(f"group_l{i + 1}", lambda: check(L.tgn_group_points(
                B, N, S, K, D, ptr(cur_xyz), ptr(lv["new_xyz"]), ptr(pts), ptr(lv["group_idx"]), self.idx64,
                int(self.xyz_first), ptr(lv["grouped"]), st), "group_points"))
            cur_xyz = lv["new_xyz"]
        if self.events is not None:
            self._step += 1
        return self.levels
