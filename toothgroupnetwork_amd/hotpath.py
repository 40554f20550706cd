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

SHAPE_A = dict(n=24000, npoint=[4096, 1024, 256], radius=[0.05, 0.1, 0.2], nsample=[32, 32, 32], d=[6, 128, 512],
               mlp=[[64, 128], [256, 512], [512, 1024]])   # fused mode only: the shared MLP of each level (one or two layers;
#                the reference networks use two everywhere, pointnet_pp.py:13-15); its last width is the next level's D
# Shape B of SURVEY.md section 8: what the reference instantiates (models/modules/pointnet_pp.py:13-15, scale 4) --
# multi-scale grouping, two radii per level, grouped layout [features, centred xyz] (pointnet2_utils.py:285)
SHAPE_B = dict(n=24000, npoint=[1024, 512, 256], radius=[[0.025, 0.05], [0.05, 0.1], [0.1, 0.2]],
               nsample=[[32, 64], [32, 64], [32, 64]], d=[6, 256, 1024], xyz_first=False,
               mlp=[[128, 128], [256, 512], [784, 1024]])   # fused mode: the reference's own two-layer MLPs, the same in both
#                branches of a level (pointnet_pp.py:13-15, scale 4); the branches' outputs sit side by side: D of the next level


# One set of side streams per (device, role) for the whole process: every HotPath launches on the same four.  Streams are not
# free -- the runtime maps them onto a handful of hardware queues (_lib.py: GPU_MAX_HW_QUEUES) and two streams on one queue
# serialise -- and two HotPath objects never need to overlap EACH OTHER.
_STREAMS = {}


def _side_stream(device, role, priority):
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), role)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device=device, priority=priority)
    return _STREAMS[key]


def _branch_mlps(mlp_level, nbranches):
    """shape['mlp'][level]: one list of widths for every branch, or one list per branch"""
    if isinstance(mlp_level[0], (list, tuple)):
        assert len(mlp_level) == nbranches
        return [list(m) for m in mlp_level]
    return [list(mlp_level)] * nbranches


def _branches(radius, nsample):
    """per-level (radius, nsample) pairs: scalars are one branch, lists are multi-scale grouping"""
    rs = list(radius) if isinstance(radius, (list, tuple)) else [radius]
    ks = list(nsample) if isinstance(nsample, (list, tuple)) else [nsample] * len(rs)
    return list(zip(rs, ks))


def algorithmic_bytes(n, npoint, nsample, d, fused=False, radius=None, mlp=None, **_):
    """Compulsory HBM bytes per scan (inputs read once, outputs written once; fp32 data, int32 indices),
    per level: FPS = 12N + 4S ; per (radius, K) branch: ball = 12N + 12S + 4SK ; group = 4SK + 4N(3+D) + 12S + 4SK(3+D)
    (Shape A 46 109 952 B, Shape B 165 863 680 B); fused: the grouped
    tensor is never written, the level's output is (S, C_out): group = 4SK + 4N(3+D) + 12S + 4*S*C_out (Shape A 12 588 288 B).
    Returns (total, per_level list of dicts)."""
    levels = []
    N = n
    for i, (S, D) in enumerate(zip(npoint, d)):
        ks = [k for _, k in _branches(radius[i] if radius is not None else 0.0, nsample[i])]
        fps = 12 * N + 4 * S
        ball = sum(12 * N + 12 * S + 4 * S * K for K in ks)
        outs = [m[-1] for m in _branch_mlps(mlp[i], len(ks))] if fused else [0] * len(ks)
        group = sum(4 * S * K + 4 * N * (3 + D) + 12 * S + (4 * S * co if fused else 4 * S * K * (3 + D)) for K, co in zip(ks, outs))
        levels.append(dict(fps=fps, ball=ball, group=group, total=fps + ball + group))
        N = S
    return sum(l["total"] for l in levels), levels


def plan_schedule(fps_l1_ms, group_ms, setup_ms, n_levels, grid_ms=0.0):
    """The two decisions of the phased schedule, from three launches timed when a HotPath is first run:
      spacer_us         how long the gated groupings are held back behind the start of FPS level 1: most of that kernel's
                        set-up (it streams its cloud three times with latency-bound loads; a grouping launch saturating HBM
                        at that moment stretches it) -- 0.6 x the time of an FPS launch that only sets up;
      last_query_early  whether the LAST level's ball query (a 22-VGPR scan: two of its waves fit per SIMD beside an FPS
                        level-1 workgroup) moves in front of the groupings, beside the next step's FPS level 1.  It pays when
                        the groupings leave room there (Shape A: 3.05 of 3.33 ms -- with the 0.15 ms the query itself takes there it just fits;
                        measured 4.69 -> 4.62 ms per step) and costs
                        when they do not (Shape B, whose groupings take ten times the FPS launch: 2 % slower)."""
    # (the groupings are gated behind the END of phase 2 -- its last query and its last FPS level; grid_ms is what still lies
    #  between that point and the start of FPS level 1 on the FPS stream: the next step's level-1 grid build, when it is built there)
    spacer_us = int(max(0.0, min(600.0, 600.0 * setup_ms + 1000.0 * grid_ms)))
    return dict(spacer_us=spacer_us, last_query_early=bool(n_levels >= 2 and group_ms + 0.15 <= fps_l1_ms),
                fps_l1_ms=float(fps_l1_ms), group_ms=float(group_ms), setup_ms=float(setup_ms), grid_ms=float(grid_ms))


class HotPath:
    """Pre-planned FPS -> ball query -> group over the levels of `shape` for a fixed batch of B scans.

    pipeline=True software-pipelines consecutive steps over three HIP streams in two phases per step:

      phase 1   stream F: FPS level 1 of step k (latency bound: one workgroup per CU, 2 x 232 of a SIMD's 512 VGPRs, 63 KiB
                of LDS, almost no issue slots or bandwidth)  ||  stream G: the groupings of step k-1 (HBM bound) in what the
                FPS workgroups leave free -- one wave of <= 48 VGPRs per SIMD and ~95 KiB of LDS per CU, which is what the
                row-piece / pairs grouping kernels are built for (4 waves per CU each; their grids are bounded to exactly
                that, so an FPS workgroup never waits for a CU to drain);
      phase 2   stream F: FPS levels 2, 3 (small workgroups, on the bucket-skipping kernel: a tenth of the vector
                instructions of the plain one, and the queries beside it are bound by vector-ALU issue)  ||  stream H: the
                ball queries (45 VGPRs at full occupancy: stream F does not start the next step's level 1 before they are
                through) -- and, on F behind FPS level 3, the level-1 ball-query grid of step k+1, which depends on the
                input cloud only.

    Buffers are double-buffered by step parity; HIP events order FPS level l before the ball query of level l, the last
    ball query of a step before its groupings and before the next step's FPS level 1, and step k-2's groupings before step
    k's producers.  Two numbers of this plan depend on the shape and the chip -- how long the groupings wait behind the start
    of FPS level 1, and whether the last level's query moves in front of them -- and are measured, not assumed: the first
    run times three launches (plan_schedule).  `plan=dict(spacer_us=.., last_query_early=..)` fixes them instead.

    fused=True: every level is a whole set-abstraction level (shape['mlp']: one- or two-layer shared MLP per (radius,
    nsample) branch, eval-mode BatchNorm folded, synthetic seeded weights): FPS -> ball query -> tgn_sa_mlp2_max /
    tgn_sa_direct_max / transform + gather-max; the branches of a multi-scale level write side by side into one tensor.  No grouped tensor and no (B,S,K,.) layer output is written; level l's (B,S,C_out) output is level l+1's
    feature input.  These kernels are bound by the fp32 matrix cores and do not fit beside an FPS level-1 workgroup (120 VGPRs,
    62 KiB of LDS): the pipelined fused schedule is FPS + ball queries of step k+1 on stream F over the set-abstraction
    kernels of step k on stream G."""

    def __init__(self, B, device, shape=SHAPE_A, index_dtype=torch.int32, pipeline=False, fps_prefix=False, fused=False,
                 plan=None, group_max_blocks=None, grid_stream="own", fps_low_valu=True, fps_ties="first"):
        self.B, self.device, self.shape = B, device, shape
        self.fused = bool(fused)
        # fps_prefix: hand every FPS level the certificate of the level that produced its input (FPS of an FPS result
        # is the identity, include/tgn_pointops.h): levels > 0 then return 0..S-1 without iterating, decided per cloud
        # on the device.  Off by default: the headline benchmark runs every level's sampling for real.
        self.fps_prefix = bool(fps_prefix)
        self.xyz_first = shape.get("xyz_first", True)
        self.L = lib()
        self.pipeline = bool(pipeline)
        self.phased = self.pipeline and not self.fused
        # phase-2 placement of the NEXT step's level-1 grid build: "own" (default) on a stream of its own, released when the previous
        # step's phase 2 starts -- it then runs beside FPS level 2 and the level-1 query instead of lengthening the FPS chain (4.55
        # against 4.74 ms per step once the queries got cheap, profiles/r06_phase2_experiments.txt); "F" behind FPS level 3 on the FPS
        # stream (rounds 2-5); "H" behind the last phase-2 ball query on the query stream.  fps_low_valu: FPS levels 2-3 on the bucket-skipping
        # kernel (few vector instructions) or on the plain register-resident one (a shorter chain that issues ten times as many)
        self.grid_stream = str(grid_stream)
        self.fps_low_valu = bool(fps_low_valu)
        # "first": first index wins a distance tie (pointnet2_utils.py:103-118, the default everywhere); "tree": the order of the
        # reference CUDA kernel's shared-memory reduction tree (sampling_cuda_kernel.cu:5-10,64-123), pinned against oracle/_ref
        if fps_ties not in ("first", "tree"):
            raise ValueError("fps_ties must be 'first' or 'tree'")
        self.fps_ties = fps_ties
        # the gated groupings run beside the FPS level-1 workgroups: 256 "blocks" = one wave per SIMD
        # (group_max_blocks: force the bound also on one stream -- counter passes that want the grouping's HBM traffic at the grid it
        # has in the phased schedule, tools/gpu_pmc.sh)
        self.group_max_blocks = int(group_max_blocks) if group_max_blocks is not None else (256 if self.phased else 0)
        self.sets = [self._alloc(B, device, shape, index_dtype) for _ in range(2 if pipeline else 1)]
        self.levels = self.sets[0]
        self.idx64 = int(index_dtype == torch.int64)
        self.events = None
        self.step_no = 0
        self.plan = dict(plan) if plan is not None else None
        if self.pipeline:
            nl = len(shape["npoint"])
            self.s_fps = _side_stream(device, "fps", -1)
            self.s_rest = _side_stream(device, "rest", 0)
            self.s_ball = _side_stream(device, "ball", -1) if self.phased else None
            self.s_grid = _side_stream(device, "grid", 0) if (self.phased and self.grid_stream == "own") else None
            self.ev_ball = [[torch.cuda.Event() for _ in range(nl)] for _ in range(2)]
            self.ev_fps = [[torch.cuda.Event() for _ in range(nl)] for _ in range(2)]
            self.ev_grid = [torch.cuda.Event() for _ in range(2)]
            self.ev_done = [torch.cuda.Event() for _ in range(2)]
            self.ev_start = torch.cuda.Event()

    def describe_schedule(self):
        if not self.pipeline:
            return "1 stream"
        if not self.phased:
            return "2 HIP streams, steps software-pipelined (FPS + ball queries of step k+1 over the set-abstraction kernels of step k)"
        pl = self.plan or {}
        nl = len(self.shape["npoint"])
        early = pl.get("last_query_early")
        return (("4" if self.s_grid is not None else "3") + " HIP streams, 2 phases per step: FPS level 1 of step k beside the groupings of step k-1" +
                (f" (held back {pl['spacer_us']} us)" if "spacer_us" in pl else "") +
                (f", then FPS levels 2-{nl} beside the ball queries of levels 1-{nl - 1} (level {nl}: in front of the groupings)"
                 if early else f", then FPS levels 2-{nl} beside the {nl} ball queries") +
                " and the next step's level-1 ball-query grid" + (" (on a stream of its own)" if self.s_grid is not None else "") +
                ("" if pl else " [plan measured on the first run]") +
                (f"; calibration: FPS level 1 {pl['fps_l1_ms']:.2f} ms, its set-up {pl['setup_ms']:.3f} ms, groupings {pl['group_ms']:.2f} ms"
                 if "fps_l1_ms" in pl else ""))

    def _alloc(self, B, device, shape, index_dtype):
        levels = []
        N = shape["n"]
        f32 = dict(dtype=torch.float32, device=device)
        for li, (S, r, K, D) in enumerate(zip(shape["npoint"], shape["radius"], shape["nsample"], shape["d"])):
            lv = dict(N=N, S=S, D=D,
                      fps_idx=torch.empty(B, S, dtype=torch.int32, device=device),
                      new_xyz=torch.empty(B, S, 3, **f32),
                      cert=torch.empty(B, dtype=torch.int32, device=device), branches=[])
            nbytes = int(self.L.tgn_ball_query_workspace_bytes(B, N, S))
            for rb, kb in _branches(r, K):   # one ball query + grouping per radius (multi-scale grouping: several)
                lv["branches"].append(dict(
                    K=kb, r2=float(torch.tensor(float(rb) ** 2, dtype=torch.float32).item()),
                    group_idx=torch.empty(B, S, kb, dtype=index_dtype, device=device),
                    grouped=None if self.fused else torch.empty(B, S, kb, 3 + D, **f32), ws_bytes=nbytes,
                    ws=torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None))
            if self.fused:
                mlps = _branch_mlps(shape["mlp"][li], len(lv["branches"]))
                if len(mlps) > 1 and any(len(m) != 2 for m in mlps):
                    raise ValueError("HotPath(fused=True): multi-scale levels need two-layer shared MLPs (the chained kernel writes "
                                     "its columns of the concatenated output)")
                lv["out"] = torch.empty(B, S, sum(m[-1] for m in mlps), **f32)      # the branches side by side (pointnet2_utils.py:296-298)
                col = 0
                for bi, (br, widths) in enumerate(zip(lv["branches"], mlps)):
                    br.update(self._fused_operands(10 * li + bi, N, S, br["K"], D, widths, device))
                    br["out"] = lv["out"][:, :, col:col + widths[-1]]
                    col += widths[-1]
                lv.update({k: lv["branches"][0][k] for k in ("layers", "C1", "C_out")})
            lv.update({k: lv["branches"][0][k] for k in ("K", "r2", "group_idx", "grouped", "ws", "ws_bytes")})
            levels.append(lv)
            N = S
        return levels

    def _fused_operands(self, li, N, S, K, D, widths, device):
        """Folded operands of level li's shared MLP (synthetic, seeded; BatchNorm = identity folded in): `layers` keeps the
        plain (C_out, C_in) matrices in the [x, y, z, features...] column order and the biases for the parity tests."""
        f32 = dict(dtype=torch.float32, device=device)
        g = torch.Generator(device="cpu").manual_seed(1000 + li)
        widths = list(widths)
        if len(widths) not in (1, 2):
            raise ValueError("HotPath(fused=True): one- or two-layer shared MLPs")
        C1 = widths[0]
        C1p = (C1 + 15) // 16 * 16 if len(widths) == 2 else C1
        W1 = torch.randn(C1, 3 + D, generator=g) / float(D + 3) ** 0.5           # columns [x, y, z, features...]
        b1 = 0.1 * torch.randn(C1, generator=g)
        layers = [(W1.numpy().copy(), b1.numpy().copy())]
        Wt = torch.zeros(D + 3, C1p)
        Wt[:D, :C1], Wt[D:, :C1] = W1[:, 3:].t(), W1[:, :3].t()                   # rows [features..., x, y, z]
        Wd = torch.zeros(16, C1p)
        if D + 3 <= 16:
            Wd[:3], Wd[3:3 + D] = Wt[D:], Wt[:D]
        b1p = torch.zeros(C1p)
        b1p[:C1] = b1
        out = dict(C1=C1, C1p=C1p, C_out=widths[-1], nlayers=len(widths), Wt=Wt.to(device).contiguous(),
                   Wxs=Wt[D:].to(device).contiguous(), Wd=Wd.to(device), b1=b1p.to(device))
        if len(widths) == 1:
            out["direct"] = bool(self.L.tgn_sa_direct_supported(K, D, C1))
        else:
            C2 = widths[1]
            W2 = torch.randn(C2, C1, generator=g) / float(C1) ** 0.5
            b2 = 0.1 * torch.randn(C2, generator=g)
            layers.append((W2.numpy().copy(), b2.numpy().copy()))
            W2p = torch.zeros(C2, C1p)
            W2p[:, :C1] = W2
            out.update(W2f=W2p.view(C2, C1p // 8, 8).permute(1, 0, 2).contiguous().to(device), b2=b2.to(device),
                       direct=bool(self.L.tgn_sa_mlp2_direct_supported(K, D)))
            from . import pointnet2_utils as U
            # second layer on the bf16 matrix cores at fp32 accuracy (tgn_sa_mlp2_max_bf16x3) unless TGN_SA_BF16X3=0
            out["W2s"] = U.split_second_layer(out["W2f"]) if U.SA_BF16X3 else None
            out["Wts"] = U.split_point_transform(out["Wt"]) if (U.SA_BF16X3 and not out["direct"]) else None
        out["layers"] = layers
        out["A"] = None if out["direct"] else torch.empty(self.B, N, C1p, **f32)
        return out

    def _ball(self, lv, br, cur_xyz, st, prebuilt=False):
        fn = self.L.tgn_ball_query_prebuilt if prebuilt else self.L.tgn_ball_query
        return check(fn(self.B, lv["N"], lv["S"], br["K"], br["r2"], ptr(cur_xyz), ptr(lv["new_xyz"]),
                        ptr(br["group_idx"]), self.idx64, ptr(br["ws"]), br["ws_bytes"], st), "ball_query")

    def _ball_build(self, lv, br, cur_xyz, st):
        return check(self.L.tgn_ball_query_build(self.B, lv["N"], lv["S"], br["K"], br["r2"], ptr(cur_xyz), ptr(br["ws"]),
                                                 br["ws_bytes"], st), "ball_query_build")

    def _group(self, lv, br, cur_xyz, pts, st):
        return check(self.L.tgn_group_points_ex(self.B, lv["N"], lv["S"], br["K"], lv["D"], ptr(cur_xyz), ptr(lv["new_xyz"]),
                                                ptr(pts), ptr(br["group_idx"]), self.idx64, int(self.xyz_first),
                                                ptr(br["grouped"]), 0, -1, self.group_max_blocks, st), "group_points")

    def _sa(self, lv, br, cur_xyz, pts, st):
        """one fused (radius, nsample) branch of a set-abstraction level on stream st"""
        L, B = self.L, self.B
        if not br["direct"] and br.get("Wts") is not None and B * lv["N"] <= 65535 * 128:
            check(L.tgn_sa_point_transform_bf16x3(B * lv["N"], lv["D"], br["Wts"][1], br["C1p"], ptr(cur_xyz), ptr(pts), ptr(br["Wts"][0]),
                                                  ptr(br["A"]), st), "sa_point_transform_bf16x3")
        elif not br["direct"]:
            check(L.tgn_sa_point_transform(B * lv["N"], lv["D"], br["C1p"], ptr(cur_xyz), ptr(pts), ptr(br["Wt"]), ptr(br["A"]), st),
                  "sa_point_transform")
        out = br["out"]
        if br["nlayers"] == 2 and br["W2s"] is not None:
            return check(L.tgn_sa_mlp2_max_bf16x3(B, lv["N"], lv["S"], br["K"], lv["D"], br["C1p"], br["C_out"], ptr(br["A"]), ptr(cur_xyz),
                                                  ptr(pts), ptr(lv["new_xyz"]), ptr(br["Wd"] if br["direct"] else br["Wxs"]), ptr(br["b1"]),
                                                  ptr(br["group_idx"]), self.idx64, ptr(br["W2s"]), ptr(br["b2"]), ptr(out),
                                                  out.stride(1), st), "sa_mlp2_max_bf16x3")
        if br["nlayers"] == 2:
            return check(L.tgn_sa_mlp2_max(B, lv["N"], lv["S"], br["K"], lv["D"], br["C1p"], br["C_out"], ptr(br["A"]), ptr(cur_xyz),
                                           ptr(pts), ptr(lv["new_xyz"]), ptr(br["Wd"] if br["direct"] else br["Wxs"]), ptr(br["b1"]),
                                           ptr(br["group_idx"]), self.idx64, ptr(br["W2f"]), ptr(br["b2"]), ptr(out), out.stride(1), st),
                         "sa_mlp2_max")
        if br["direct"]:
            return check(L.tgn_sa_direct_max(B, lv["N"], lv["S"], br["K"], lv["D"], br["C1"], ptr(cur_xyz), ptr(lv["new_xyz"]),
                                             ptr(pts), ptr(br["Wd"]), ptr(br["b1"]), ptr(br["group_idx"]), self.idx64, 1,
                                             ptr(out), st), "sa_direct_max")
        return check(L.tgn_sa_gather_max(B, lv["N"], lv["S"], br["K"], br["C1"], ptr(br["A"]), ptr(lv["new_xyz"]), ptr(br["Wxs"]),
                                         ptr(br["b1"]), ptr(br["group_idx"]), self.idx64, 1, ptr(out), st), "sa_gather_max")

    def _consume(self, i, lv, cur_xyz, feats, levels, st):
        """what follows the ball query of level i: the grouping (materialised) or the fused level"""
        if self.fused:
            pts = feats[0] if i == 0 else levels[i - 1]["out"]     # level l consumes level l-1's output features
            return [self._sa(lv, br, cur_xyz, pts, st) for br in lv["branches"]]
        return [self._group(lv, br, cur_xyz, feats[i], st) for br in lv["branches"]]

    def phase2_event(self):
        """(pipelined, phased mode) the event recorded behind FPS level 1 of the most recently issued step -- the start of its phase
        2, the one stretch of a step in which the whole chip takes short kernels (beside FPS level 1 anything larger than 48 VGPRs
        keeps a CU from starting its FPS workgroup and anything smaller shares ONE wave slot per SIMD with the groupings).  A
        caller that prepares the next inputs on a stream of its own -- tools/secondary_bench.py: hot_path_with_h2d splits the
        coordinate block off behind it -- waits for this event first.  None before the first step."""
        if not self.phased or self.step_no < 1:
            return None
        return self.ev_fps[(self.step_no - 1) & 1][0]

    def take_index_error(self):
        """HotPath launches unchecked on streams of its own (the ball queries it runs produce valid indices by construction;
        a caller-supplied index tensor would not): True if any of its kernels latched an out-of-range gather index since the
        flags were last read.  Synchronises the device (every stream's flag is read: _lib.take_index_error_device)."""
        return _lib.take_index_error_device()

    def enable_kernel_timing(self, steps, stride=1, only=None):
        """HIP events on the launch stream around each kernel class (start/stop), on every `stride`-th step: a timing
        event is a barrier packet in its queue, and a dozen of them per step cost the pipelined schedule ~4 % even at
        stride 4 (4.55 against 4.37 ms per step, profiles/r06_timeline.txt).  only=("fps_l1",) times that kernel class alone
        (two events per timed step between launches that are serialised anyway)."""
        names = [f"{k}_l{i + 1}" for i in range(len(self.levels)) for k in ("fps", "ball", "group")]
        if only is not None:
            names = [n for n in names if n in only]
        self.events = {n: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                           if s_ % stride == 0 else None for s_ in range(steps)] for n in names}
        self._step = 0

    def kernel_times_ms(self):
        return {n: [e[0].elapsed_time(e[1]) for e in evs[:self._step] if e is not None] for n, evs in self.events.items()}

    def _timed(self, name, fn, stream=None):
        if self.events is None or name not in self.events or self._step >= len(self.events[name]) or self.events[name][self._step] is None:
            return fn()
        a, b = self.events[name][self._step]
        a.record(stream)
        fn()
        b.record(stream)

    def run(self, xyz, feats, inputs_on_current_stream=True, input_event=None, more=True):
        """xyz: (B, N, 3) fp32 contiguous; feats: list of per-level feature tensors (B, N_l, D_l).
        Results land in self.levels[l]['grouped'] etc. (pipelined: self.sets[step parity]).  Asynchronous.
        Pipelined mode: the results of a call live in one of two buffer sets and are overwritten by the call after
        next.  By default every call first waits for the caller's stream, which orders it after whatever the caller
        has enqueued there -- producing the inputs, reading earlier results.  inputs_on_current_stream=False drops
        that wait (the inputs were complete long ago, e.g. a resident dataset), so that this step does not wait for
        the previous step's results and consecutive steps overlap; the caller then has to make sure on its own that
        its reads of step k's results are done before it issues call k+2 (tools/pipeline_stress.py).
        more=False (pipelined mode): no further call follows this one.  The groupings of a step normally run beside the NEXT step's
        FPS level 1, their grids bounded to the one wave per SIMD it leaves free; the last step of a batch has nothing beside it
        and lets them use the whole chip (1.8 instead of 3.0 ms of tail).
        input_event (pipelined mode): a torch.cuda.Event behind which this call's inputs are complete -- e.g. recorded on a copy
        stream behind the host-to-device copy of this step's scans; the step's streams wait for it and for nothing else of the
        caller's, so the copy of step k+1 overlaps step k."""
        if self.pipeline:
            if self.phased and self.plan is None:
                self.plan = self._calibrate(xyz, feats)
            return self._run_pipelined(xyz, feats, inputs_on_current_stream, input_event, more)
        return self._run_one_stream(xyz, feats, self.levels, timed=True)

    def _run_one_stream(self, xyz, feats, levels, timed=False):
        st = _lib.stream()
        cur_xyz = xyz
        run = self._timed if timed else (lambda name, fn: fn())
        for i, lv in enumerate(levels):
            run(f"fps_l{i + 1}", lambda: self._fps(i, lv, cur_xyz, levels, st))
            run(f"ball_l{i + 1}", lambda: [self._ball(lv, br, cur_xyz, st) for br in lv["branches"]])
            run(f"group_l{i + 1}", lambda: self._consume(i, lv, cur_xyz, feats, levels, st))
            cur_xyz = lv["new_xyz"]
        if timed and self.events is not None:
            self._step += 1
        return levels

    def _calibrate(self, xyz, feats):
        """Three timed launches on the caller's stream before the first pipelined step, behind one untimed pass (the results
        they leave in buffer set 0 are the step's own, computed once more by the step itself): FPS level 1; an FPS launch
        of 2 samples (= its set-up); the groupings at the grid they get beside FPS.  One synchronisation, once per HotPath."""
        levels = self.sets[0]
        st = _lib.stream()
        self._run_one_stream(xyz, feats, levels)      # untimed: code objects loaded, caches and clocks warm
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        lv0 = levels[0]
        ev[0].record()
        self._fps(0, lv0, xyz, levels, st)
        ev[1].record()
        scratch_idx = torch.empty(self.B, 2, dtype=torch.int32, device=self.device)
        ev[2].record()
        check(self.L.tgn_furthestsampling_dense(self.B, lv0["N"], 2, ptr(xyz), None, ptr(scratch_idx), None, _lib.FPS_LOCAL_INDEX, st),
              "fps set-up probe")
        ev[3].record()
        cur = xyz
        for i, lv in enumerate(levels):      # everything the groupings need
            if i > 0:
                self._fps(i, lv, cur, levels, st)
            for br in lv["branches"]:
                self._ball(lv, br, cur, st)
            cur = lv["new_xyz"]
        ev[4].record()
        cur = xyz
        for i, lv in enumerate(levels):
            self._consume(i, lv, cur, feats, levels, st)
            cur = lv["new_xyz"]
        ev[5].record()
        ev[6].record()
        for br in lv0["branches"]:
            self._ball_build(lv0, br, xyz, st)
        ev[7].record()
        ev[7].synchronize()
        return plan_schedule(ev[0].elapsed_time(ev[1]), ev[4].elapsed_time(ev[5]), ev[2].elapsed_time(ev[3]), len(levels),
                             grid_ms=ev[6].elapsed_time(ev[7]) if self.grid_stream == "F" else 0.0)

    def _fps(self, i, lv, cur_xyz, levels, st):
        L = self.L
        # phased schedule: FPS levels 2-3 run beside the ball queries, which are bound by vector-ALU issue -- the bucket-skipping
        # kernel issues a tenth of the plain kernel's vector instructions (0.86 vs 0.75 ms for level 2 itself, but the level-1
        # ball query beside it 1.10 instead of 1.28 ms)
        flags = (_lib.FPS_LOCAL_INDEX | (_lib.FPS_LOW_VALU if (self.phased and i > 0 and self.fps_low_valu) else 0)
                 | (_lib.FPS_TREE_TIES if self.fps_ties == "tree" else 0))
        if not self.fps_prefix:
            return check(L.tgn_furthestsampling_dense(self.B, lv["N"], lv["S"], ptr(cur_xyz), None, ptr(lv["fps_idx"]),
                                                      ptr(lv["new_xyz"]), flags, st), "fps")
        cert_in = levels[i - 1]["cert"] if i > 0 else None   # level i samples level i-1's new_xyz
        return check(L.tgn_furthestsampling_dense_prefix(self.B, lv["N"], lv["S"], ptr(cur_xyz), None, 0, ptr(lv["fps_idx"]),
                                                         ptr(lv["new_xyz"]), ptr(cert_in), None, ptr(lv["cert"]),
                                                         flags, st), "fps")

    def _run_pipelined(self, xyz, feats, inputs_on_current_stream=True, input_event=None, more=True):
        p = self.step_no & 1
        levels = self.sets[p]
        nl = len(levels)
        sf, sg, sb = self.s_fps, self.s_rest, self.s_ball
        pf, pg = _lib.c_void_p(sf.cuda_stream), _lib.c_void_p(sg.cuda_stream)
        pb = _lib.c_void_p(sb.cuda_stream) if sb is not None else None
        cur = torch.cuda.current_stream()
        if input_event is not None:
            for s_ in (sf, sg, sb, getattr(self, "s_grid", None)):
                if s_ is not None:
                    s_.wait_event(input_event)
        elif inputs_on_current_stream or self.step_no == 0:
            self.ev_start.record(cur)      # inputs produced on the caller's stream
            for s_ in (sf, sg, sb, getattr(self, "s_grid", None)):
                if s_ is not None:
                    s_.wait_event(self.ev_start)
        if self.step_no >= 2:
            sf.wait_event(self.ev_done[p])  # buffer set p is free again once step k-2's consumers are through
        clouds = [xyz] + [lv["new_xyz"] for lv in levels[:-1]]      # the cloud level i samples / queries / groups from
        if self.phased:
            # levels >= early are queried on stream G, in front of the groupings (beside the NEXT step's FPS level 1)
            early = nl - 1 if self.plan.get("last_query_early") else nl
            # the level-1 grid depends on the input cloud only: it goes onto stream F BEFORE the fence below, i.e. behind
            # the previous step's FPS level 3, where stream F would otherwise idle until that step's ball queries are done
            if self.grid_stream == "F" or self.step_no == 0:
                sx, px = sf, pf
            elif self.grid_stream == "H":
                sx, px = sb, pb                                  # behind the previous step's phase-2 queries, already enqueued there
            else:
                sx, px = self.s_grid, _lib.c_void_p(self.s_grid.cuda_stream)
                sx.wait_event(self.ev_fps[1 - p][0])             # released when the previous step's phase 2 starts
            for br in levels[0]["branches"]:
                self._ball_build(levels[0], br, xyz, px)
            self.ev_grid[p].record(sx)
            if sx is not sf:
                sf.wait_event(self.ev_grid[p])                   # (a grid build cannot share a CU with an FPS level-1 workgroup)
            if self.step_no >= 1:
                sf.wait_event(self.ev_ball[1 - p][early - 1])   # the previous step's phase-2 queries are through
            for i, lv in enumerate(levels):
                self._timed(f"fps_l{i + 1}", lambda: self._fps(i, lv, clouds[i], levels, pf), sf)
                self.ev_fps[p][i].record(sf)
                if i >= early:
                    continue
                sb.wait_event(self.ev_fps[p][i])
                if i == 0:
                    sb.wait_event(self.ev_grid[p])
                # (levels > 0: the grid was built on this stream behind the previous level's query -- its cloud, the previous level's
                #  samples, was complete then; only the queries wait for this level's FPS)
                self._timed(f"ball_l{i + 1}", lambda: [self._ball(lv, br, clouds[i], pb, prebuilt=True) for br in lv["branches"]], sb)
                self.ev_ball[p][i].record(sb)
                if i + 1 < early:
                    for br in levels[i + 1]["branches"]:
                        self._ball_build(levels[i + 1], br, clouds[i + 1], pb)
            bound = self.group_max_blocks
            if more:
                # all of this step's groupings run beside the NEXT step's FPS level 1: released by the last phase-2 query
                sg.wait_event(self.ev_ball[p][early - 1])
                sg.wait_event(self.ev_fps[p][nl - 1])     # ... and by the last FPS level: the end of phase 2, whichever chain is longer
                if self.plan.get("spacer_us"):
                    check(self.L.tgn_stream_delay(int(self.plan["spacer_us"]), pg), "stream_delay")
                for j in range(early, nl):
                    lj = levels[j]
                    self._timed(f"ball_l{j + 1}", lambda: [self._ball(lj, br, clouds[j], pg) for br in lj["branches"]], sg)
                    self.ev_ball[p][j].record(sg)
                for i, lv in enumerate(levels):
                    self._timed(f"group_l{i + 1}", lambda: self._consume(i, lv, clouds[i], feats, levels, pg), sg)
            else:
                # the last step of a batch: nothing follows, nothing to keep clear of -- each grouping starts when its own query is
                # through, on the whole chip (the queries of levels >= early run here, each behind its FPS level)
                self.group_max_blocks = 0
                for i, lv in enumerate(levels):
                    if i < early:
                        sg.wait_event(self.ev_ball[p][i])
                    else:
                        sg.wait_event(self.ev_fps[p][i])
                        self._timed(f"ball_l{i + 1}", lambda: [self._ball(lv, br, clouds[i], pg) for br in lv["branches"]], sg)
                        self.ev_ball[p][i].record(sg)
                    self._timed(f"group_l{i + 1}", lambda: self._consume(i, lv, clouds[i], feats, levels, pg), sg)
            self.group_max_blocks = bound
        else:
            for i, lv in enumerate(levels):
                self._timed(f"fps_l{i + 1}", lambda: self._fps(i, lv, clouds[i], levels, pf), sf)
                self._timed(f"ball_l{i + 1}", lambda: [self._ball(lv, br, clouds[i], pf) for br in lv["branches"]], sf)
                self.ev_fps[p][i].record(sf)
            for i, lv in enumerate(levels):
                sg.wait_event(self.ev_fps[p][i])
                self._timed(f"group_l{i + 1}", lambda: self._consume(i, lv, clouds[i], feats, levels, pg), sg)
        self.ev_done[p].record(sg)
        cur.wait_event(self.ev_done[p])     # the caller's stream sees this step's results
        self.step_no += 1
        if self.events is not None:
            self._step += 1
        return levels
