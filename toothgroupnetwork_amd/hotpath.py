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
               c_out=[128, 512, 1024])   # c_out: width of the (single-layer) shared MLP of each level, fused mode only
# Shape B of SURVEY.md section 8: what the reference instantiates (models/modules/pointnet_pp.py:13-15, scale 4) --
# multi-scale grouping, two radii per level, grouped layout [features, centred xyz] (pointnet2_utils.py:285)
SHAPE_B = dict(n=24000, npoint=[1024, 512, 256], radius=[[0.025, 0.05], [0.05, 0.1], [0.1, 0.2]],
               nsample=[[32, 64], [32, 64], [32, 64]], d=[6, 256, 1024], xyz_first=False)


def _branches(radius, nsample):
    """per-level (radius, nsample) pairs: scalars are one branch, lists are multi-scale grouping"""
    rs = list(radius) if isinstance(radius, (list, tuple)) else [radius]
    ks = list(nsample) if isinstance(nsample, (list, tuple)) else [nsample] * len(rs)
    return list(zip(rs, ks))


def algorithmic_bytes(n, npoint, nsample, d, fused=False, radius=None, c_out=None, **_):
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
        group = sum(4 * S * K + 4 * N * (3 + D) + 12 * S + (4 * S * c_out[i] if fused else 4 * S * K * (3 + D)) for K in ks)
        levels.append(dict(fps=fps, ball=ball, group=group, total=fps + ball + group))
        N = S
    return sum(l["total"] for l in levels), levels


def default_ball_split(shape, B):
    """Where the last level's ball query goes in the phased schedule (HotPath, ball_split): 4 = in front of the groupings,
    beside the next step's FPS level 1 -- which pays when the groupings leave room there (Shape A: 3.0 of 3.3 ms; Shape B's
    groupings take ten times the FPS launch and the same move costs 2 %) -- else 0.  Estimate: grouping bytes at the ~3.8 TB/s
    the kernels reach beside FPS against (S1 - 1) iterations of ~0.81 us."""
    if len(shape["npoint"]) < 2:
        return 0
    _, per = algorithmic_bytes(**shape)
    group_ms = sum(l["group"] for l in per) * B / 3.8e9
    fps_ms = (shape["npoint"][0] - 1) * 0.81e-3
    return 4 if group_ms + 0.25 <= fps_ms else 0


class HotPath:
    """Pre-planned FPS -> ball query -> group over `levels` for a fixed batch of B scans.

    pipeline=True software-pipelines consecutive steps over three HIP streams in two phases per step:

      phase 1   stream F: FPS level 1 of step k (latency bound: one workgroup per CU, 2 x 232 of a SIMD's 512 VGPRs, 63 KiB
                of LDS, almost no issue slots or bandwidth)  ||  stream G: the groupings of step k-1 (HBM bound) in what the
                FPS workgroups leave free -- one wave of <= 48 VGPRs per SIMD and ~95 KiB of LDS per CU, which is what the
                row-piece / pairs grouping kernels are built for (4 waves per CU each);
      phase 2   stream F: FPS levels 2, 3 (small workgroups, VALU bound)  ||  stream H: the three ball queries (VALU bound,
                45 VGPRs at full occupancy: stream F does not start the next step's level 1 before the level-2 query is
                through; the last level's query -- a 22-VGPR scan -- moves in front of the groupings of phase 1 when those leave
                room) -- and, on F behind FPS level 3, the level-1 ball-query grid of step k+1.

    Buffers are double-buffered by step parity; HIP events order FPS level l before the ball query of level l, the last ball
    query of a step before its groupings (group_gate) and before the next step's FPS level 1 (ball_stream = 2), and step
    k-2's groupings before step k's producers.  ball_stream = 0 / group_gate = False give the round-1 two-stream schedule
    (ball queries in line on stream F, groupings released level by level)."""

    def __init__(self, B, device, shape=SHAPE_A, xyz_first=True, index_dtype=torch.int32, pipeline=False,
                 fps_prefix=False, group_impl=0, group_policy=-1, group_max_blocks=None, fused=False, ball_stream=None,
                 group_gate=None, early_grid=None, ball_split=None, grid_stream=False, low_valu=True, group_order=None, group_delay_us=None):
        self.B, self.device, self.shape = B, device, shape
        # launch knobs of the grouping kernel (include/tgn_pointops.h, tgn_group_points_ex).  In the pipelined schedule its
        # grid is bounded to what fits beside the FPS level-1 workgroups, so that those never wait for a CU to drain:
        # 256 "blocks" = one wave per SIMD (4 single-wave workgroups of the row-piece kernel per CU).
        nl = len(shape["npoint"])
        per_level = lambda v, d: [int(x) for x in v] if isinstance(v, (list, tuple)) else [int(d if v is None else v)] * nl
        self.group_impl, self.group_policy = per_level(group_impl, 0), per_level(group_policy, -1)   # scalar or one value per level
        self.group_max_blocks = per_level(group_max_blocks, 256 if pipeline else 0)
        # fused: every level is a whole set-abstraction level with a single-layer shared MLP (eval-mode BatchNorm folded):
        # FPS -> ball query -> [per-point transform on the fp32 matrix cores + gather-max | direct kernel]; the grouped
        # tensor is never written and level l's (B,S,C_out) output is level l+1's feature input.
        self.fused = bool(fused)
        # fps_prefix: hand every FPS level the certificate of the level that produced its input (FPS of an FPS result
        # is the identity, include/tgn_pointops.h): levels > 0 then return 0..S-1 without iterating, decided per cloud
        # on the device.  Off by default: the headline benchmark runs every level's sampling for real.
        self.fps_prefix = bool(fps_prefix)
        self.xyz_first = shape.get("xyz_first", xyz_first)
        self.L = lib()
        self.pipeline = pipeline
        self.sets = [self._alloc(B, device, shape, index_dtype) for _ in range(2 if pipeline else 1)]
        self.levels = self.sets[0]
        self.idx64 = int(index_dtype == torch.int64)
        self.events = None
        self.step_no = 0
        # schedule knobs (defaults = the phased schedule above; the fused mode keeps the two-stream one: its
        # set-abstraction kernels are matrix-core bound and do not fit the leftovers of an FPS workgroup)
        dflt = pipeline and not self.fused
        self.ball_stream = (2 if dflt else 0) if ball_stream is None else (int(ball_stream) if pipeline else 0)
        self.group_gate = dflt if group_gate is None else (bool(group_gate) and pipeline)
        self.early_grid = (self.ball_stream == 2) if early_grid is None else (bool(early_grid) and self.ball_stream == 2)
        self.low_valu = bool(low_valu)
        # The first ~0.17 ms of an FPS level-1 workgroup is its set-up: it streams its cloud three times (latency-bound loads),
        # and a grouping launch that saturates HBM at the same moment stretches it (3.59 instead of 3.46 ms for the launch).
        # The gated groupings are therefore held back by a one-wave spacer kernel for about as long as the set-up takes
        # (~4 ns per point of the level-1 cloud: 100 us for 24 000 points; 0: 3.43 instead of 3.32 ms for the launch; 150: the
        # work beside FPS then ends after it and runs into phase 2).
        self.group_delay_us = (min(300, shape["n"] // 240) if (self.group_gate and not self.fused) else 0) \
            if group_delay_us is None else int(group_delay_us)
        self.group_order = list(group_order) if group_order else None   # gated schedule: order of the grouping launches
        if pipeline:
            self.s_fps = torch.cuda.Stream(device=device, priority=-1)
            self.s_rest = torch.cuda.Stream(device=device, priority=0)
            self.s_ball = torch.cuda.Stream(device=device, priority=-1) if self.ball_stream else None
            # ball_split: where the queries of the levels after the first run.
            #   0  behind the level-1 query on stream H (phase 2);
            #   4  (default where it pays, below) the LAST level's query -- a 22-VGPR scan over 1024-point clouds -- in front of
            #      this step's groupings on stream G, i.e. beside the NEXT step's FPS level 1, and the fence in front of that
            #      FPS launch waits for the level-2 query only: 0.20 instead of 0.07 ms for the query, but the groupings leave
            #      0.4 of the 3.4 ms free and phase 2 ends 0.07 ms earlier: 4.69 -> 4.62 ms per step;
            #   5  the same for every level after the first: the level-2 query (45 VGPRs: one workgroup per CU there) needs
            #      1.2-4 ms beside FPS: 5.2-7.6 ms per step;
            #   1, 2  (experiments of the first half of round 2) levels 2-3 / the last level on a second query
            #      stream in phase 2: 5.44 / 5.195 ms against 5.18 then -- the small kernels at the tail of phase 2 slow each
            #      other down by what the overlap gains.
            if ball_split is None:
                ball_split = default_ball_split(shape, B) if self.group_gate else 0
            self.ball_split = int(ball_split) if self.ball_stream == 2 else 0
            self.s_ball2 = torch.cuda.Stream(device=device, priority=-1) if self.ball_split in (1, 2, 3) else None
            self.shadow_from = {4: nl - 1, 5: 1}.get(self.ball_split, nl) if self.group_gate else nl
            self.ev_lgrid = [[torch.cuda.Event() for _ in shape["npoint"]] for _ in range(2)]
            self.ev_ball = [[torch.cuda.Event() for _ in shape["npoint"]] for _ in range(2)]
            self.ev_fps = [[torch.cuda.Event() for _ in shape["npoint"]] for _ in range(2)]
            self.ev_grid = [torch.cuda.Event() for _ in range(2)]
            # the early level-1 grid on a stream of its own (released when the previous step's FPS level 1 is done, i.e. at
            # the start of phase 2) instead of behind FPS level 3 on stream F
            self.s_grid = torch.cuda.Stream(device=device, priority=-1) if (self.early_grid and grid_stream) else None
            self.ev_done = [torch.cuda.Event() for _ in range(2)]
            self.ev_start = torch.cuda.Event()

    def describe_schedule(self):
        if not self.pipeline:
            return "1 stream"
        if self.ball_stream == 2:
            return ("3 HIP streams, 2 phases per step: FPS level 1 of step k beside the groupings of step k-1" +
                    ("" if self.group_gate else " (released level by level)") +
                    (", then FPS levels 2-3 beside the ball queries of levels 1-%d (level %d: in front of the groupings)"
                     % (self.shadow_from, self.shadow_from + 1) if getattr(self, "shadow_from", 99) < len(self.shape["npoint"])
                     else ", then FPS levels 2-3 beside the three ball queries") +
                    (" and the next step's level-1 ball-query grid" if self.early_grid else ""))
        if self.ball_stream == 1:
            return "3 HIP streams, free-running (FPS chain | ball queries | groupings)"
        return "2 HIP streams, steps software-pipelined (FPS of step k+1 over ball query + group of step k)"

    def _alloc(self, B, device, shape, index_dtype):
        levels = []
        N = shape["n"]
        f32 = dict(dtype=torch.float32, device=device)
        for S, r, K, D in zip(shape["npoint"], shape["radius"], shape["nsample"], shape["d"]):
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
                li = len(levels)
                C1 = shape["c_out"][li]
                g = torch.Generator(device="cpu").manual_seed(1000 + li)
                # folded weights of the level's Conv2d(3+D -> C1) + BatchNorm (synthetic, seeded): Wt rows [features.., x,y,z]
                Wt = (torch.randn(D + 3, C1, generator=g) / float(D + 3) ** 0.5).to(device)
                Wd = torch.zeros(16, C1, device=device)
                if D + 3 <= 16:
                    Wd[:3], Wd[3:3 + D] = Wt[D:], Wt[:D]
                lv.update(C1=C1, Wt=Wt.contiguous(), Wxs=Wt[D:].contiguous(), Wd=Wd, b2=(0.1 * torch.randn(C1, generator=g)).to(device),
                          out=torch.empty(B, S, C1, **f32),
                          direct=bool(self.L.tgn_sa_direct_supported(lv["branches"][0]["K"], D, C1)))
                lv["A"] = None if lv["direct"] else torch.empty(B, N, C1, **f32)
            lv.update({k: lv["branches"][0][k] for k in ("K", "r2", "group_idx", "grouped", "ws", "ws_bytes")})
            levels.append(lv)
            N = S
        return levels

    def _ball(self, lv, br, cur_xyz, st, prebuilt=False):
        fn = self.L.tgn_ball_query_prebuilt if prebuilt else self.L.tgn_ball_query
        return check(fn(self.B, lv["N"], lv["S"], br["K"], br["r2"], ptr(cur_xyz), ptr(lv["new_xyz"]),
                        ptr(br["group_idx"]), self.idx64, ptr(br["ws"]), br["ws_bytes"], st), "ball_query")

    def _ball_build(self, lv, br, cur_xyz, st):
        return check(self.L.tgn_ball_query_build(self.B, lv["N"], lv["S"], br["K"], br["r2"], ptr(cur_xyz), ptr(br["ws"]),
                                                 br["ws_bytes"], st), "ball_query_build")

    def _group(self, lv, br, cur_xyz, pts, st, i=0):
        return check(self.L.tgn_group_points_ex(self.B, lv["N"], lv["S"], br["K"], lv["D"], ptr(cur_xyz), ptr(lv["new_xyz"]),
                                                ptr(pts), ptr(br["group_idx"]), self.idx64, int(self.xyz_first),
                                                ptr(br["grouped"]), self.group_impl[i], self.group_policy[i],
                                                self.group_max_blocks[i], st), "group_points")

    def _sa(self, lv, br, cur_xyz, pts, st):
        """one fused set-abstraction level on stream st (tgn_sa_direct_max, or tgn_sa_point_transform + tgn_sa_gather_max)"""
        L, B = self.L, self.B
        if lv["direct"]:
            return check(L.tgn_sa_direct_max(B, lv["N"], lv["S"], br["K"], lv["D"], lv["C1"], ptr(cur_xyz), ptr(lv["new_xyz"]),
                                             ptr(pts), ptr(lv["Wd"]), ptr(lv["b2"]), ptr(br["group_idx"]), self.idx64, 1,
                                             ptr(lv["out"]), st), "sa_direct_max")
        check(L.tgn_sa_point_transform(B * lv["N"], lv["D"], lv["C1"], ptr(cur_xyz), ptr(pts), ptr(lv["Wt"]), ptr(lv["A"]), st),
              "sa_point_transform")
        return check(L.tgn_sa_gather_max(B, lv["N"], lv["S"], br["K"], lv["C1"], ptr(lv["A"]), ptr(lv["new_xyz"]), ptr(lv["Wxs"]),
                                         ptr(lv["b2"]), ptr(br["group_idx"]), self.idx64, 1, ptr(lv["out"]), st), "sa_gather_max")

    def _consume(self, i, lv, cur_xyz, feats, levels, st):
        """what follows the ball query of level i: the grouping (materialised) or the fused level"""
        if self.fused:
            pts = feats[0] if i == 0 else levels[i - 1]["out"]     # level l consumes level l-1's output features
            return [self._sa(lv, br, cur_xyz, pts, st) for br in lv["branches"]]
        return [self._group(lv, br, cur_xyz, feats[i], st, i) for br in lv["branches"]]

    def enable_kernel_timing(self, steps, stride=1):
        """HIP events on the launch stream around each kernel class (start/stop), on every `stride`-th step: a timing
        event is a barrier packet in its queue, and a dozen of them per step cost the pipelined schedule ~5 %."""
        names = [f"{k}_l{i + 1}" for i in range(len(self.levels)) for k in ("fps", "ball", "group")]
        self.events = {n: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                           if s_ % stride == 0 else None for s_ in range(steps)] for n in names}
        self._step = 0

    def kernel_times_ms(self):
        return {n: [e[0].elapsed_time(e[1]) for e in evs[:self._step] if e is not None] for n, evs in self.events.items()}

    def _timed(self, name, fn, stream=None):
        if self.events is None or self._step >= len(self.events[name]) or self.events[name][self._step] is None:
            return fn()
        a, b = self.events[name][self._step]
        a.record(stream)
        fn()
        b.record(stream)

    def run(self, xyz, feats, inputs_on_current_stream=True):
        """xyz: (B, N, 3) fp32 contiguous; feats: list of per-level feature tensors (B, N_l, D_l).
        Results land in self.levels[l]['grouped'] etc. (pipelined: self.sets[step parity]).  Asynchronous.
        Pipelined mode: the results of a call live in one of two buffer sets and are overwritten by the call after
        next.  By default every call first waits for the caller's stream, which orders it after whatever the caller
        has enqueued there -- producing the inputs, reading earlier results.  inputs_on_current_stream=False drops
        that wait (the inputs were complete long ago, e.g. a resident dataset), so that this step does not wait for
        the previous step's results and consecutive steps overlap; the caller then has to make sure on its own that
        its reads of step k's results are done before it issues call k+2 (tools/pipeline_stress.py)."""
        if self.pipeline:
            return self._run_pipelined(xyz, feats, inputs_on_current_stream)
        L, st = self.L, _lib.stream()
        cur_xyz = xyz
        for i, lv in enumerate(self.levels):
            B, N, S, K, D = self.B, lv["N"], lv["S"], lv["K"], lv["D"]
            self._timed(f"fps_l{i + 1}", lambda: self._fps(i, lv, cur_xyz, self.levels, st))
            self._timed(f"ball_l{i + 1}", lambda: [self._ball(lv, br, cur_xyz, st) for br in lv["branches"]])
            self._timed(f"group_l{i + 1}", lambda: self._consume(i, lv, cur_xyz, feats, self.levels, st))
            cur_xyz = lv["new_xyz"]
        if self.events is not None:
            self._step += 1
        return self.levels

    def _fps(self, i, lv, cur_xyz, levels, st):
        L = self.L
        # phased schedule: FPS levels 2-3 run beside the ball queries, which are bound by vector-ALU issue -- the bucket-skipping
        # kernel issues a tenth of the plain kernel's vector instructions (0.86 vs 0.75 ms for level 2 itself, but the level-1
        # ball query beside it 1.10 instead of 1.28 ms)
        flags = _lib.FPS_LOCAL_INDEX | (_lib.FPS_LOW_VALU if (self.low_valu and self.pipeline and self.ball_stream == 2 and i > 0) else 0)
        if not self.fps_prefix:
            return check(L.tgn_furthestsampling_dense(self.B, lv["N"], lv["S"], ptr(cur_xyz), None, ptr(lv["fps_idx"]),
                                                      ptr(lv["new_xyz"]), flags, st), "fps")
        cert_in = levels[i - 1]["cert"] if i > 0 else None   # level i samples level i-1's new_xyz
        return check(L.tgn_furthestsampling_dense_prefix(self.B, lv["N"], lv["S"], ptr(cur_xyz), None, 0, ptr(lv["fps_idx"]),
                                                         ptr(lv["new_xyz"]), ptr(cert_in), None, ptr(lv["cert"]),
                                                         flags, st), "fps")

    def _run_pipelined(self, xyz, feats, inputs_on_current_stream=True):
        p = self.step_no & 1
        levels = self.sets[p]
        sf, sg, sb, sb2 = self.s_fps, self.s_rest, self.s_ball, self.s_ball2
        pf, pg = _lib.c_void_p(sf.cuda_stream), _lib.c_void_p(sg.cuda_stream)
        pb = _lib.c_void_p(sb.cuda_stream) if sb is not None else None
        pb2 = _lib.c_void_p(sb2.cuda_stream) if sb2 is not None else None
        cur = torch.cuda.current_stream()
        if inputs_on_current_stream or self.step_no == 0:
            self.ev_start.record(cur)      # inputs produced on the caller's stream
            for s_ in (sf, sg, sb, sb2):
                if s_ is not None:
                    s_.wait_event(self.ev_start)
        if self.step_no >= 2:
            sf.wait_event(self.ev_done[p])  # buffer set p is free again once step k-2's consumers are through
        if self.early_grid:
            # the level-1 grid depends on the input cloud only: it goes onto stream F BEFORE the fence below, i.e. behind
            # the previous step's FPS level 3, where stream F would otherwise idle until that step's ball queries are done
            sq = self.s_grid if self.s_grid is not None else sf
            if self.s_grid is not None:
                if inputs_on_current_stream or self.step_no == 0:
                    sq.wait_event(self.ev_start)
                if self.step_no >= 1:
                    sq.wait_event(self.ev_fps[1 - p][0])     # not beside an FPS level-1 workgroup: phase 2 of the previous step
                if self.step_no >= 2:
                    sq.wait_event(self.ev_done[p])
            for br in levels[0]["branches"]:
                self._ball_build(levels[0], br, xyz, _lib.c_void_p(sq.cuda_stream))
            self.ev_grid[p].record(sq)
            if self.s_grid is not None:
                sf.wait_event(self.ev_grid[p])               # ... and out of the way before this step's level 1 starts
        shadow_from = getattr(self, "shadow_from", len(levels))
        if self.ball_stream == 2 and self.step_no >= 1:
            last = min(shadow_from, len(levels)) - 1
            for ev in (self.ev_ball[1 - p] if sb2 is not None else self.ev_ball[1 - p][last:last + 1]):
                sf.wait_event(ev)                    # phased: the previous step's ball queries (of phase 2) are through
        cur_xyz = xyz
        nl = len(levels)
        split = self.ball_split if sb2 is not None else 0
        for i, lv in enumerate(levels):
            self._timed(f"fps_l{i + 1}", lambda: self._fps(i, lv, cur_xyz, levels, pf), sf)
            if sb is None:
                self._timed(f"ball_l{i + 1}", lambda: [self._ball(lv, br, cur_xyz, pf) for br in lv["branches"]], sf)
            self.ev_fps[p][i].record(sf)
            if sb is not None and i >= shadow_from:
                pass                                 # queried on stream G, in front of the groupings (below)
            elif sb is not None:
                if split == 2 and i + 1 < nl:
                    # the next level's grid: its cloud (this level's samples) exists now
                    sb2.wait_event(self.ev_fps[p][i])
                    for br in levels[i + 1]["branches"]:
                        self._ball_build(levels[i + 1], br, lv["new_xyz"], pb2)
                    self.ev_lgrid[p][i + 1].record(sb2)
                on2 = (split == 1 and i > 0) or (split == 2 and i == nl - 1 and i > 0)
                sq, pq = (sb2, pb2) if on2 else (sb, pb)
                sq.wait_event(self.ev_fps[p][i])
                pre = (self.early_grid and i == 0) or (split == 2 and i > 0)
                if self.early_grid and i == 0:
                    sq.wait_event(self.ev_grid[p])
                if split == 2 and i > 0:
                    sq.wait_event(self.ev_lgrid[p][i])
                self._timed(f"ball_l{i + 1}", lambda: [self._ball(lv, br, cur_xyz, pq, prebuilt=pre) for br in lv["branches"]], sq)
                self.ev_ball[p][i].record(sq)
            cur_xyz = lv["new_xyz"]
        cur_xyz = xyz
        ev_q = self.ev_ball if sb is not None else self.ev_fps    # "the queries of level i are done"
        clouds = [xyz] + [lv["new_xyz"] for lv in levels[:-1]]      # the cloud level i groups from
        order = self.group_order if (self.group_gate and self.group_order) else list(range(len(levels)))
        for n_, i in enumerate(order):
            lv, cur_xyz = levels[i], clouds[i]
            if self.group_gate:
                if n_ == 0:
                    if sb is not None and shadow_from < nl:
                        sg.wait_event(ev_q[p][shadow_from - 1])
                        sg.wait_event(self.ev_fps[p][nl - 1])
                    else:
                        for ev in (ev_q[p] if sb2 is not None else ev_q[p][-1:]):
                            sg.wait_event(ev)        # all of this step's groupings run beside the NEXT step's FPS level 1
                    if self.group_delay_us:
                        check(self.L.tgn_stream_delay(self.group_delay_us, pg), "stream_delay")
                    if sb is not None:
                        for j in range(shadow_from, nl):
                            lj, cj = levels[j], clouds[j]
                            self._timed(f"ball_l{j + 1}", lambda: [self._ball(lj, br, cj, pg) for br in lj["branches"]], sg)
                            self.ev_ball[p][j].record(sg)
            else:
                sg.wait_event(ev_q[p][i])
            self._timed(f"group_l{i + 1}", lambda: self._consume(i, lv, cur_xyz, feats, levels, pg), sg)
        self.ev_done[p].record(sg)
        cur.wait_event(self.ev_done[p])     # the caller's stream sees this step's results
        self.step_no += 1
        if self.events is not None:
            self._step += 1
        return levels
