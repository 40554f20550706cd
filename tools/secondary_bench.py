#!/usr/bin/env python3
"""The non-headline configurations of BASELINE.json, measured in the SAME process as bench.py's headline (after it) so that
the driver's bench run witnesses them: bench.py attaches `secondary` = measure_all(...) to its JSON line at N = 1.

Every entry: {"value", "unit", "ms", "config", "roofline": {"bound", "achieved", "peak", "unit", "frac"}} -- the roofline of
the kernel family that bounds the entry, priced like the headline's (algorithmic bytes or flops per launch / HIP-event time).
An entry that fails reports {"error": ...} and never takes the headline down.

  shape_B_materialised   config 2's operators at the widths the reference net instantiates (pointnet_pp.py:13-15), grouped tensors written
  fused_shape_A / _B     the same levels with the shared MLP fused in; _B is the reference net's own SA stack
  gather_family          grouping / subtraction / aggregation / interpolation forward + backward (pointops_api.cpp:12-23) at config 4's first stage
  knn_24000_k36          pointops.knnquery at config 4's first stage (blocks.py:34)
  fps_100k_to_24k        preprocess_data.py:55-56 / gen_utils.py:124-140, one scan and 64 scans per launch (config 5's kernel)
  pnpp_forward_8x24000   config 2's whole network (pointnet_pp.py get_model) on 8 scans, eval
  pt_forward_24000       config 4: PointTransformerSeg encoder / decoder forward on one 24 000-point scan, eager and as a HIP graph
  train_step_graph       config 3's first-stage step (forward, loss terms, backward, Adam) as one HIP graph, fp32

    python tools/secondary_bench.py            (stand-alone: prints the dict)"""
import importlib.util
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from toothgroupnetwork_amd import _lib, hotpath, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0
MFMA_FP32_PEAK = 157.3      # TFLOP/s, v_mfma_f32_32x32x2_f32 (MI355X_MICROARCH.md)
MFMA_BF16_PEAK = 2500.0     # TFLOP/s dense, v_mfma_f32_32x32x16_bf16 (same guide)


def _sa_peak():
    """(peak, what it is) for the fused set-abstraction second layer: an fp32 product is SIX bf16 MFMA products in the default
    bf16x3 form, so the ceiling in fp32-equivalent flops is the bf16 peak / 6; the exact-fp32 form is priced against
    the fp32 MFMA peak."""
    from toothgroupnetwork_amd import pointnet2_utils as U
    if U.SA_BF16X3:
        return MFMA_BF16_PEAK / 6.0, "fp32-equivalent flops; peak = bf16 dense MFMA peak / 6 (bf16x3: six bf16 products per fp32 product)"
    return MFMA_FP32_PEAK, "exact fp32 MFMA (TGN_SA_BF16X3=0)"


def _roof(bound, achieved, peak, unit, **more):
    return dict(bound=bound, achieved=achieved, peak=peak, unit=unit, frac=(achieved / peak) if peak else None, **more)


def _events(fn, reps, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def fused_flops(shape):
    """fp32 flops per scan of the fused levels: first layers per POINT where the transform commutes with the gather (3+D > 16), per
    gathered row in the direct form; second layers per gathered row."""
    fl, Nl = 0, shape["n"]
    for S, r, K, D, m in zip(shape["npoint"], shape["radius"], shape["nsample"], shape["d"], shape["mlp"]):
        brs = hotpath._branches(r, K)
        for (_, kb), widths in zip(brs, hotpath._branch_mlps(m, len(brs))):
            direct = (3 + D) <= 16
            fl += 2 * (S * kb if direct else Nl) * (3 + D) * widths[0]
            if len(widths) == 2:
                fl += 2 * S * kb * widths[0] * widths[1]
        Nl = S
    return fl


def hot_path(make_inputs, device, shape_name, fused, B=256, steps=6, warmup=2):
    shape = hotpath.SHAPE_A if shape_name == "A" else hotpath.SHAPE_B
    xyz, feats, _ = make_inputs(B, device, 100, shape)
    hp = hotpath.HotPath(B, device, shape=shape, pipeline=True, fused=fused)
    for _ in range(warmup):
        hp.run(xyz, feats, inputs_on_current_stream=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        hp.run(xyz, feats, inputs_on_current_stream=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    value = B * steps / dt
    nbytes, _ = hotpath.algorithmic_bytes(fused=fused, **shape)
    out = dict(value=value, unit="meshes/s", ms=1e3 * dt / steps, steps=steps,
               config=f"shape_{shape_name}{' fused' if fused else ' materialised'}, {B} scans per step, phased schedule")
    if fused:
        fl = fused_flops(shape)
        tf = fl * value / 1e12
        peak, what = _sa_peak()
        out["roofline"] = _roof("mfma", tf, peak, "TFLOP/s", flops_per_scan=fl, fp32_mfma_peak=MFMA_FP32_PEAK,
                                note="whole step (FPS and ball queries hide under the set-abstraction kernels); " + what)
    else:
        out["roofline"] = _roof("hbm", nbytes * value / 1e9, HBM_PEAK_GBS, "GB/s", algorithmic_bytes_per_scan=nbytes,
                                note="whole path; the grouping stores are the bulk")
    del hp
    torch.cuda.empty_cache()
    return out


def hot_path_tree_ties(make_inputs, device, B=256, steps=8, warmup=3):
    """The headline step with the FPS tie order of the reference's CUDA kernel (sampling_cuda_kernel.cu:5-10,64-123: the winner of a
    distance tie is whoever the shared-memory reduction tree keeps) instead of first-index-wins -- the one FPS mode that is pinned
    against the reference's own kernel (oracle/_ref), next to the default schedule measured the same way."""
    shape = hotpath.SHAPE_A
    xyz, feats, _ = make_inputs(B, device, 100, shape)
    res = {}
    for ties in ("first", "tree"):
        hp = hotpath.HotPath(B, device, shape=shape, pipeline=True, fps_ties=ties)
        for _ in range(warmup):
            hp.run(xyz, feats, inputs_on_current_stream=False)
        torch.cuda.synchronize()
        hp.enable_kernel_timing(steps, stride=2)
        t0 = time.perf_counter()
        for _ in range(steps):
            hp.run(xyz, feats, inputs_on_current_stream=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        kt = {k: float(np.mean(v)) for k, v in hp.kernel_times_ms().items() if v}
        res[ties] = dict(value=B * steps / dt, ms=1e3 * dt / steps, fps_l1_ms=kt.get("fps_l1"),
                         us_per_fps_iteration=1e3 * kt["fps_l1"] / (shape["npoint"][0] - 1) if "fps_l1" in kt else None,
                         kernel_ms={k: round(v, 4) for k, v in sorted(kt.items())})
        del hp
        torch.cuda.empty_cache()
    t, f = res["tree"], res["first"]
    return dict(value=t["value"], unit="meshes/s", ms=t["ms"], steps=steps, fps_l1_ms=t["fps_l1_ms"], us_per_fps_iteration=t["us_per_fps_iteration"],
                kernel_ms=t["kernel_ms"], first_index_ties=f, slowdown_vs_first_index=t["ms"] / f["ms"],
                config=f"shape_A materialised, {B} scans per step, phased schedule, FPS with TGN_FPS_TREE_TIES at every level",
                note="the tie key of the tree order is (bit-reversed reference thread, position within it) instead of the point index: one more "
                     "compare per candidate where distances tie exactly, nothing else at levels 1-2; at level 3 the lean kernel keeps a tree "
                     "key per point and pays two more instructions per selection step; the prefix-identity shortcut is off in this mode "
                     "(include/tgn_pointops.h) and is not part of the headline either")


def hot_path_with_h2d(make_inputs, device, B=256, steps=8, warmup=3):
    """The headline step fed from HOST memory (BASELINE.md section 3 excludes the copy from the headline and asks for it separately;
    the reference moves every scan host -> device, gen_utils.py:138, pointnet_pp_model.py:16-20): each step's 256 scans (xyz + normals,
    147 MB) are copied from PINNED memory on a copy stream into one of two device buffers while the previous step computes; the
    step's streams wait for the copy's event only.  Reported: meshes/s and GB/s over PCIe of that pipeline, the pipelined rate with
    resident inputs measured the same way, and -- once -- what the same copy costs from pageable memory."""
    shape = hotpath.SHAPE_A
    xyz_d, feats_d, scans = make_inputs(B, device, 100, shape)
    host = torch.from_numpy(scans).pin_memory()                         # (B, N, 6) fp32
    nbytes = host.numel() * 4
    hp = hotpath.HotPath(B, device, shape=shape, pipeline=True)
    s_copy = torch.cuda.Stream(device=device)
    # FOUR input buffers: step k's scans are copied while step k-1 samples its own (FPS level 1 reads them for 3.3 ms), step
    # k-2's groupings still gather from theirs, and the copy has to be through one phase earlier than the step needs it (the
    # split waits for a phase-2 start) -- with two, the copy could only start when those groupings are through, one
    # millisecond before its step begins (measured: 10.4 ms per step, the copy fully exposed)
    NB = 4
    pts = [torch.empty_like(feats_d[0]) for _ in range(NB)]
    xyzs = [torch.empty_like(xyz_d) for _ in range(NB)]
    ev_in = [torch.cuda.Event() for _ in range(NB)]
    ev_after = [torch.cuda.Event() for _ in range(NB)]       # on the caller's stream behind run(): that step's results are complete
    marks = {}

    s_split = torch.cuda.Stream(device=device, priority=-1)  # the split has a stream of its own: a gated split must not hold back the next copy
    ev_copied = [torch.cuda.Event() for _ in range(NB)]

    def step(k, mark=False):
        p = k % NB
        with torch.cuda.stream(s_copy):
            if k >= NB:
                s_copy.wait_event(ev_after[p])                         # buffer p's last readers: step k-4's groupings
            pts[p].copy_(host, non_blocking=True)
            ev_copied[p].record(s_copy)
        with torch.cuda.stream(s_split):
            s_split.wait_event(ev_copied[p])
            # the (B, N, 3) coordinate block the samplers read: split off when the running step's phase 2 starts (beside FPS level 1
            # the split shares one wave slot per SIMD with the groupings -- 2-3 ms, torch's strided copy_ or a row kernel alike --
            # and a wave of more than 48 VGPRs keeps a CU from starting its FPS workgroup: profiles/r06_h2d.txt)
            gate = hp.phase2_event()
            if gate is not None:
                s_split.wait_event(gate)
            _lib.check(_lib.lib().tgn_slice_columns(B * shape["n"], 6, 0, 3, _lib.ptr(pts[p]), _lib.ptr(xyzs[p]),
                                                    _lib.c_void_p(s_split.cuda_stream)), "slice_columns")
            ev_in[p].record(s_split)
        hp.run(xyzs[p], [pts[p]] + feats_d[1:], inputs_on_current_stream=False, input_event=ev_in[p])
        ev_after[p].record(torch.cuda.current_stream())
        if mark:
            marks[k] = torch.cuda.Event(enable_timing=True)
            marks[k].record(torch.cuda.current_stream())

    # steady state: `steps` consecutive steps in the middle of a longer run, between two events on the caller's stream (each
    # is complete when its step's results are); no synchronisation in between -- the copy of step k+1 runs under step k
    lead = 6
    for k in range(lead + steps + 2):
        step(k, mark=k in (lead - 1, lead + steps - 1))
    torch.cuda.synchronize()
    dt = 1e-3 * marks[lead - 1].elapsed_time(marks[lead + steps - 1])

    hp2 = hotpath.HotPath(B, device, shape=shape, pipeline=True)
    m2 = {}
    for k in range(lead + steps + 2):                                   # the same measurement with resident inputs
        hp2.run(xyz_d, feats_d, inputs_on_current_stream=False)
        if k in (lead - 1, lead + steps - 1):
            m2[k] = torch.cuda.Event(enable_timing=True)
            m2[k].record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    dt_res = 1e-3 * m2[lead - 1].elapsed_time(m2[lead + steps - 1])
    # the copy alone: pinned and pageable
    copy_ms = _events(lambda: pts[0].copy_(host, non_blocking=True), 5, 1)
    pageable = torch.from_numpy(scans.copy())
    page_ms = _events(lambda: pts[0].copy_(pageable), 3, 1)
    value = B * steps / dt
    return dict(value=value, unit="meshes/s", ms=1e3 * dt / steps, steps=steps, pcie_GBs=nbytes * steps / dt / 1e9,
                bytes_per_step=nbytes, resident_inputs={"value": B * steps / dt_res, "ms": 1e3 * dt_res / steps},
                slowdown_vs_resident=dt / dt_res,
                copy_alone={"pinned_ms": copy_ms, "pinned_GBs": nbytes / copy_ms / 1e6, "pageable_ms": page_ms, "pageable_GBs": nbytes / page_ms / 1e6},
                config=f"shape_A materialised, {B} scans per step staged from pinned host memory on a copy stream (four input buffers), "
                       f"overlapped with the previous step; {steps} consecutive steps in steady state between two events",
                roofline=_roof("pcie", nbytes * steps / dt / 1e9, 64.0, "GB/s",
                               note="PCIe Gen5 x16 = 64 GB/s per direction (raw); the step needs bytes_per_step / ms_resident"))


def _valu_counts():
    """committed instruction counts of the issue-bound scans and the measured issue ceiling (profiles/r06_valu_counts.json:
    rocprofv3 --pmc SQ_INSTS_VALU passes and tools/valu_bench.hip)"""
    try:
        return json.load(open(os.path.join(REPO, "profiles", "r06_valu_counts.json")))
    except Exception:
        return None


def _valu_roof(entry, ms, note):
    """fraction of the vector-issue ceiling: (wave-instructions per launch / SIMDs) x ceiling ns per instruction / measured time"""
    vc = _valu_counts()
    if not vc or entry not in vc:
        return _roof("valu", None, None, "ms", note="profiles/r06_valu_counts.json is missing: no instruction count to price against")
    c = vc["issue_ceiling"]
    floor_ms = vc[entry]["sq_insts_valu"] / c["simds"] * c["ns_per_wave_instruction_per_simd"] * 1e-6
    return _roof("valu", ms, floor_ms, "ms per launch (lower is better; frac = time at the vector-issue ceiling / achieved)",
                 wave_instructions_per_launch=vc[entry]["sq_insts_valu"], valu_per_query=vc[entry]["sq_insts_valu"] / vc[entry]["queries"],
                 ceiling_ns_per_wave_instruction_per_simd=c["ns_per_wave_instruction_per_simd"], simds=c["simds"],
                 counts_source="profiles/r06_valu_counts.json (committed rocprofv3 --pmc SQ_INSTS_VALU pass of the same workload, not this run)",
                 note=note) | {"frac": floor_ms / ms}


def knn(device):
    from toothgroupnetwork_amd import pointops as P
    xyz = torch.from_numpy(synth.arch_cloud(24000, 1, False)).to(device)
    off = torch.tensor([24000], dtype=torch.int32, device=device)

    def run():
        P.knn_cache_clear()
        P.knnquery(36, xyz, xyz, off, off)
    ms = _events(run, 10, 2)
    nbytes = 12 * 24000 * 2 + 8 * 24000 * 36
    return dict(value=1e3 / ms, unit="calls/s", ms=ms, config="pointops.knnquery(36, xyz, xyz) on one 24 000-point scan (grid kernel)",
                roofline=_valu_roof("knn_24000_k36", ms,
                                    "one call = grid build + query + heap replay launches on one scan; the query kernel keeps 36-entry sorted "
                                    "lists per lane and waits on dependent record loads for 46 % of its wave cycles (SQ_WAIT_INST_ANY / "
                                    "SQ_WAVE_CYCLES): latency of a batch-1 call, not issue -- which is what a fraction of 0.15 says"),
                hbm_view=dict(algorithmic_bytes=nbytes, GBs=nbytes / ms / 1e6, pair_evaluations_brute_force=24000 * 24000))


def ball_l1(make_inputs, device, B=256):
    """The level-1 ball query of the headline alone (1 048 576 queries over 256 grids, K = 32): vector instructions per launch from the
    committed counter pass, against the measured vector-issue ceiling."""
    shape = hotpath.SHAPE_A
    xyz, feats, _ = make_inputs(B, device, 100, shape)
    hp = hotpath.HotPath(B, device, shape=shape, pipeline=False)
    hp.run(xyz, feats)
    torch.cuda.synchronize()
    lv = hp.levels[0]
    st = _lib.stream()
    ms = _events(lambda: [hp._ball(lv, br, xyz, st, prebuilt=True) for br in lv["branches"]], 10, 2)
    ms_build = _events(lambda: [hp._ball_build(lv, br, xyz, st) for br in lv["branches"]], 10, 2)
    q = B * shape["npoint"][0]
    out = dict(value=q / ms / 1e3, unit="M queries/s", ms=ms, grid_build_ms=ms_build,
               config=f"query_ball_point(0.05, 32) of {B} x 24 000-point scans, 4096 queries each, grid prebuilt (chunked bitmap kernel)",
               roofline=_valu_roof("ball_l1", ms, "bound by vector-ALU issue and the latency of its dependent LDS / L2 reads; 185 vector "
                                                  "instructions per query (round 2's kernel: 367), profiles/r06_ball_sq_by_stage.txt"))
    del hp
    torch.cuda.empty_cache()
    return out


def fps_large(device):
    out = {}
    for nscan in (1, 64):
        pts = np.stack([synth.arch_cloud(100000, 200 + (i % 4), False) for i in range(min(nscan, 4))])
        pts = np.concatenate([pts] * ((nscan + 3) // 4))[:nscan]
        xyz = torch.from_numpy(pts).to(device).contiguous()
        idx = torch.empty(nscan, 24000, dtype=torch.int32, device=device)
        from toothgroupnetwork_amd.pointops import fps_workspace
        ws, nbytes = fps_workspace(nscan, 100000, nscan * 100000, device)
        L = _lib.lib()

        def run():
            _lib.check(L.tgn_furthestsampling_dense_ws(nscan, 100000, 24000, _lib.ptr(xyz), _lib.ptr(ws), nbytes, _lib.ptr(idx), None,
                                                       _lib.FPS_LOCAL_INDEX, _lib.stream()), "fps")
        ms = _events(run, 3, 1)
        out[f"{nscan}_scan{'s' if nscan > 1 else ''}"] = dict(
            value=nscan * 1e3 / ms, unit="scans/s", ms=ms,
            roofline=dict(_roof("latency", 1e3 * ms / 23999, fps_floor_us(), "us per FPS iteration (frac = floor / achieved)",
                                hbm_algorithmic_GBs=(12 * 100000 + 96000) * nscan / ms / 1e6,
                                note="serial chain of 23 999 block-wide arg-maxes per scan, one workgroup per scan out of an L2-resident workspace; "
                                     "peak = the dependent chain of one iteration with the data work removed (tools/fps_floor.hip)"),
                          frac=(fps_floor_us() / (1e3 * ms / 23999)) if fps_floor_us() else None))
    return dict(config="tgn_furthestsampling_dense_ws 100 000 -> 24 000 points (large-cloud owner-wave bucket kernel)", **out)


def pt_forward(device):
    from toothgroupnetwork_amd import nets, pointops as P
    torch.manual_seed(0)
    net = nets.PointTransformerSeg().to(device).eval()
    inp = torch.from_numpy(synth.scan_batch(1, 24000, "arch", 3).transpose(0, 2, 1).copy()).to(device)
    with torch.no_grad():
        eager = _events(lambda: net([inp]), 5, 2)
    out = dict(value=1e3 / eager, unit="scans/s", ms=eager,
               config="nets.PointTransformerSeg (tgnet_fps stage sizes, class + offset heads) forward, one 24 000-point scan, eval, fp32")
    out.update(_graph_ms(lambda x: net([x]), inp))
    if "graph_ms" in out:
        out["graph_scans_per_s"] = 1e3 / out["graph_ms"]
    floor_ms = 5999 * fps_floor_us() * 1e-3 if fps_floor_us() else None
    best = out.get("graph_ms", eager)
    out["roofline"] = dict(_roof("latency", best, floor_ms, "ms per forward (frac = floor / achieved)",
                                 note="a single-scan forward is bound by its sampling chain: 24 000 -> 6000 = 5 999 serial block-wide arg-maxes in one "
                                      "workgroup (the coarser levels sample an FPS result: identity, certificate checked on the device); peak = those "
                                      "iterations at the measured chain floor (tools/fps_floor.hip); the rest is ~330 small launches"),
                           frac=(floor_ms / best) if floor_ms else None)
    return out


def _graph_ms(fn, inp):
    """fn(static input) captured in one HIP graph after two warm-up runs on a side stream; median replay time, or the error."""
    from toothgroupnetwork_amd import pointops as P
    try:
        static_in = inp.clone()
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_), torch.no_grad():
            for _ in range(2):
                fn(static_in)
        torch.cuda.current_stream().wait_stream(s_)
        torch.cuda.synchronize()
        P.knn_cache_clear()
        P.fps_prefix_clear()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            static_out = fn(static_in)
        P.knn_cache_clear()
        P.fps_prefix_clear()
        ms = _events(g.replay, 8, 2)
        del g, static_out
        return {"graph_ms": ms}
    except Exception as e:  # noqa: BLE001
        return {"graph_error": f"{type(e).__name__}: {str(e)[:200]}"}


def pnpp_forward(device, B=8):
    """BASELINE config 2's network end to end: models/modules/pointnet_pp.py:43-70 (three multi-scale set-abstraction levels, three
    feature-propagation levels, heads) in eval mode on B scans of 24 000 points -- every SA branch one chained kernel."""
    from toothgroupnetwork_amd import nets
    torch.manual_seed(0)
    net = nets.PointNetPPSeg().to(device).eval()
    inp = torch.from_numpy(synth.scan_batch(B, 24000, "arch", 5).transpose(0, 2, 1).copy()).to(device)
    with torch.no_grad():
        ms = _events(lambda: net([inp]), 6, 3)
    graph_ms = _graph_ms(lambda x: net([x]), inp)
    # second-layer flops of the six SA branches + per-point first layers (the rest -- FP stack, heads -- is ~25 % on top)
    fl = fused_flops(hotpath.SHAPE_B)
    return dict(value=B * 1e3 / ms, unit="scans/s", ms=ms, **graph_ms,
                config=f"nets.PointNetPPSeg (pointnet_pp.py get_model, scale 4) forward, {B} x 24 000-point scans, eval, eager (graph_ms: one HIP graph)",
                roofline=_roof("mfma", fl * B / ms / 1e9, _sa_peak()[0], "TFLOP/s", fp32_mfma_peak=MFMA_FP32_PEAK,
                               note="set-abstraction flops only over the WHOLE forward time, which also holds sampling, ball queries, three "
                                    "feature-propagation levels and the heads; " + _sa_peak()[1]))


def train_step(device, steps=6):
    spec = importlib.util.spec_from_file_location("train_step_bench", os.path.join(REPO, "tools", "train_step_bench.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    feat, xyz, label = T.make_scan(24000, 3, device)
    torch.manual_seed(0)
    net = T.FirstStage().to(device).train()
    net.unet.presample = False
    opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True, capturable=True)
    r = T.run_graph(net, opt, feat, xyz, label, steps, False)
    ms = float(np.median([x["ms"] for x in r[1:]]))
    return dict(value=1e3 / ms, unit="steps/s", ms=ms, first_loss=r[0]["loss"], last_loss=r[-1]["loss"],
                config="tgnet_fps first-stage network (7.8 M parameters), one 24 000-point scan: forward (training-mode BatchNorm), "
                       "cross entropy + centroid-offset terms, backward, fused Adam -- one HIP graph, fp32",
                roofline=_roof("latency", ms, None, "ms per step",
                               note="batch 1: ~2000 small kernels; bf16 autocast measured no faster at this size (DESIGN.md 4.6)"))


def gather_family(device, n=24000, ns=36, c=32, wc=4, k=3, m_coarse=6000, batch_launches=10):
    """The gather / scatter operators of pointops_api.cpp:12-23 at the first Point-Transformer stage (n = 24 000 points, nsample = 36,
    c = 32, w_c = c / 8 = 4; interpolation: dec1's 6000 -> 24 000, k = 3), each priced against HBM: algorithmic bytes = every distinct
    operand byte once (an accumulated output counts read + write) / the launch time (HIP events around `batch_launches` back-to-back
    launches).  Neighbour lists are a real kNN of an arch scan, so the gathers have the locality the network sees."""
    from toothgroupnetwork_amd import pointops as P
    L, p = _lib.lib(), _lib.ptr
    xyz = torch.from_numpy(synth.arch_cloud(n, 1, False)).to(device)
    off = torch.tensor([n], dtype=torch.int32, device=device)
    P.knn_cache_clear()
    idx, _ = P.knnquery(ns, xyz, xyz, off, off)
    idx = idx.contiguous()
    coarse = xyz[torch.randperm(n, device=device)[:m_coarse]].contiguous()
    coff = torch.tensor([m_coarse], dtype=torch.int32, device=device)
    ik, dk = P.knnquery(k, coarse, xyz, coff, off)
    ik = ik.contiguous()
    wk = (1.0 / (dk + 1e-8))
    wk = (wk / wk.sum(1, keepdim=True)).contiguous()
    g = torch.Generator(device="cpu").manual_seed(5)
    R = lambda *sh: torch.randn(*sh, generator=g).to(device)   # noqa: E731
    x, y, pos, w = R(n, c), R(n, c), R(n, ns, c), R(n, ns, wc)
    go3, go2, xc = R(n, ns, c), R(n, c), R(m_coarse, c)
    out3, o2 = torch.empty(n, ns, c, device=device), torch.zeros(n, c, device=device)
    gi, g1, g2 = torch.zeros(n, c, device=device), torch.zeros(n, c, device=device), torch.zeros(n, c, device=device)
    ga, gp, gw = torch.zeros(n, c, device=device), torch.zeros(n, ns, c, device=device), torch.zeros(n, ns, wc, device=device)
    gc = torch.zeros(m_coarse, c, device=device)
    st = _lib.stream
    F = 4
    rows = n * ns
    ops = {
        "grouping_fwd": (lambda: L.tgn_grouping_forward(n, ns, c, p(x), p(idx), p(out3), st()), F * (n * c + rows + rows * c)),
        "grouping_bwd": (lambda: L.tgn_grouping_backward(n, ns, c, p(go3), p(idx), p(gi), st()), F * (rows * c + rows + 2 * n * c)),
        "subtraction_fwd": (lambda: L.tgn_subtraction_forward(n, ns, c, p(x), p(y), p(idx), p(out3), st()), F * (2 * n * c + rows + rows * c)),
        "subtraction_bwd": (lambda: L.tgn_subtraction_backward(n, ns, c, p(idx), p(go3), p(g1), p(g2), st()), F * (rows + rows * c + 4 * n * c)),
        "aggregation_fwd": (lambda: L.tgn_aggregation_forward(n, ns, c, wc, p(x), p(pos), p(w), p(idx), p(o2), st()),
                            F * (n * c + rows * c + rows * wc + rows + 2 * n * c)),
        "aggregation_bwd": (lambda: L.tgn_aggregation_backward(n, ns, c, wc, p(x), p(pos), p(w), p(idx), p(go2), p(ga), p(gp), p(gw), st()),
                            F * (n * c + rows * c + rows * wc + rows + n * c + 2 * n * c + rows * c + 2 * rows * wc)),
        "interpolation_fwd": (lambda: L.tgn_interpolation_forward(n, c, k, p(xc), p(ik), p(wk), p(o2), st()),
                              F * (m_coarse * c + 2 * n * k + 2 * n * c)),
        "interpolation_bwd": (lambda: L.tgn_interpolation_backward(n, c, k, p(go2), p(ik), p(wk), p(gc), st()),
                              F * (n * c + 2 * n * k + 2 * m_coarse * c)),
    }
    out = dict(config=f"tgn_* gather family at n = {n}, nsample = {ns}, c = {c}, w_c = {wc} (interpolation: {m_coarse} -> {n}, k = {k}); one launch each, "
                      f"timed over {batch_launches} back-to-back launches")
    for name, (fn, nbytes) in ops.items():
        def batch():
            for _ in range(batch_launches):
                _lib.check(fn(), name)
        ms = _events(batch, 7, 2) / batch_launches
        out[name] = dict(us=1e3 * ms, algorithmic_bytes=nbytes,
                         roofline=_roof("hbm", nbytes / ms / 1e6, HBM_PEAK_GBS, "GB/s"))
    out["interpolation_fwd"]["note"] = out["interpolation_bwd"]["note"] = \
        "a few MB per launch: ~1 us of HBM time, so the launch itself is the floor at this size"
    # The backward kernels scatter into data-dependent rows: fp32 atomics, one dword per element, executed by the L2 channels' atomic
    # units -- their ceiling, not HBM's, bounds these kernels.  tools/atomic_floor.hip measures it (register operands, whole-line
    # instructions, the same table size); `roofline_atomic` = this kernel's dword-atomics per second against that.
    floor = atomic_floor()
    out["atomic_floor"] = floor
    atomics = {"grouping_bwd": rows * c, "subtraction_bwd": rows * c, "aggregation_bwd": rows * c, "interpolation_bwd": n * k * c}
    for name, cnt in atomics.items():
        rate = cnt / (out[name]["us"] * 1e-6) / 1e9
        out[name]["roofline_atomic"] = _roof("l2_atomic", rate, floor.get("line_gatomics_per_s"), "G dword-atomics/s", atomics_per_launch=cnt)
    return out


_FPS_FLOOR = {}


def fps_floor_us():
    """us per FPS iteration of the dependent chain alone (tools/fps_floor.hip), measured once per process; None if the tool is missing"""
    if "v" not in _FPS_FLOOR:
        import subprocess
        try:
            r = subprocess.run([os.path.join(REPO, "tools", "_bin", "fps_floor"), "--json"], capture_output=True, text=True, timeout=120)
            _FPS_FLOOR["v"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["fps_floor"]["chain_us"]
        except Exception:  # noqa: BLE001
            _FPS_FLOOR["v"] = None
    return _FPS_FLOOR["v"]


def atomic_floor():
    import subprocess
    exe = os.path.join(REPO, "tools", "_bin", "atomic_floor")
    try:
        r = subprocess.run([exe, "--json"], capture_output=True, text=True, timeout=120)
        return {**json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["atomic_floor"], "source": "tools/_bin/atomic_floor on this GPU"}
    except Exception as e:  # noqa: BLE001
        return {"error": f"atomic_floor unavailable: {type(e).__name__}: {str(e)[:100]}"}


def measure_all(make_inputs, device, budget_s=270.0, checkpoint=None):
    """checkpoint(out): called after every entry (bench.py's child process rewrites its result file there)"""
    t0 = time.perf_counter()
    out = {}
    plan = [("shape_A_tree_ties", lambda: hot_path_tree_ties(make_inputs, device)),
            ("shape_A_with_h2d", lambda: hot_path_with_h2d(make_inputs, device)),
            ("shape_B_materialised", lambda: hot_path(make_inputs, device, "B", False, steps=10, warmup=3)),
            ("fused_shape_A", lambda: hot_path(make_inputs, device, "A", True, steps=10, warmup=3)),
            ("fused_shape_B", lambda: hot_path(make_inputs, device, "B", True, steps=8, warmup=3)),
            ("gather_family", lambda: gather_family(device)),
            ("ball_l1", lambda: ball_l1(make_inputs, device)),
            ("knn_24000_k36", lambda: knn(device)),
            ("fps_100k_to_24k", lambda: fps_large(device)),
            ("pnpp_forward_8x24000", lambda: pnpp_forward(device)),
            ("pt_forward_24000", lambda: pt_forward(device)),
            ("train_step_graph", lambda: train_step(device))]
    for name, fn in plan:
        if time.perf_counter() - t0 > budget_s:
            out[name] = {"skipped": f"time budget of {budget_s:.0f} s used up"}
            continue
        t1 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        out[name]["wall_s"] = round(time.perf_counter() - t1, 2)
        if checkpoint is not None:
            checkpoint(out)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    out["total_wall_s"] = round(time.perf_counter() - t0, 2)
    return out


if __name__ == "__main__":
    spec = importlib.util.spec_from_file_location("bench", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    json_out = sys.argv[sys.argv.index("--json-out") + 1] if "--json-out" in sys.argv else None
    from toothgroupnetwork_amd import sharding
    torch.set_num_threads(max(1, min(torch.get_num_threads(), sharding.effective_cpus())))   # (the CPU quota, not the hardware thread count)

    def _save(res):
        tmp = json_out + ".tmp"
        with open(tmp, "w") as f:
            json.dump(res, f)
        os.replace(tmp, json_out)

    result = measure_all(bench.make_inputs, torch.device("cuda", 0), checkpoint=_save if json_out else None)
    if json_out:
        _save(result)
    else:
        print(json.dumps(result, indent=1))
