#!/usr/bin/env python3
"""Phase 2 of the headline step in isolation: FPS levels 2-3 + the next step's level-1 grid build on one
stream, the ball queries of levels 1-2 on another.  Times every kernel alone, the two chains alone, and both together, on the
buffers of a HotPath that has run one step (256 distinct scans unless --unique).  Kernel variants are selected through the
TGN_* environment / tgn_set_tuning keys of the process, so an A/B is two invocations.

    python tools/phase2_bench.py [--reps 20] [--unique 0] [--set key=value ...]
"""
import argparse
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import bench  # noqa: E402
from toothgroupnetwork_amd import _lib, hotpath  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--unique", type=int, default=0)
    ap.add_argument("--set", action="append", default=[], help="tuning key=value (tgn_set_tuning)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    for kv in args.set:
        k, v = kv.split("=")
        assert L.tgn_set_tuning(k.encode(), int(v)) == 0, kv
    shape = hotpath.SHAPE_A
    xyz, feats, _ = bench.make_inputs(args.batch, dev, seed=100, shape=shape, unique=args.unique)
    hp = hotpath.HotPath(args.batch, dev, shape=shape, pipeline=True)
    for _ in range(3):
        hp.run(xyz, feats, inputs_on_current_stream=False)
    torch.cuda.synchronize()
    lv = hp.sets[0]
    sf, sh = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    pf, ph = _lib.c_void_p(sf.cuda_stream), _lib.c_void_p(sh.cuda_stream)
    hp.phased = True
    clouds = [xyz, lv[0]["new_xyz"], lv[1]["new_xyz"]]

    def fps(i, st):
        hp._fps(i, lv[i], clouds[i], lv, st)

    def ball(i, st, prebuilt):
        for br in lv[i]["branches"]:
            hp._ball(lv[i], br, clouds[i], st, prebuilt=prebuilt)

    def grid0(st):
        for br in lv[0]["branches"]:
            hp._ball_build(lv[0], br, xyz, st)

    ev_l2 = torch.cuda.Event()

    def chain_f():
        fps(1, pf)
        ev_l2.record(sf)
        fps(2, pf)
        grid0(pf)

    def chain_h(wait=False):
        ball(0, ph, True)
        if wait:
            sh.wait_event(ev_l2)       # the level-2 query needs FPS level 2 (as in the step)
        ball(1, ph, False)

    def timed(fn_f=None, fn_h=None):
        ts = []
        for _ in range(args.reps):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cur = torch.cuda.current_stream()
            a.record(cur)
            sf.wait_event(a)
            sh.wait_event(a)
            if fn_f:
                fn_f()
            if fn_h:
                fn_h()
            ef, eh = torch.cuda.Event(), torch.cuda.Event()
            ef.record(sf)
            eh.record(sh)
            cur.wait_event(ef)
            cur.wait_event(eh)
            b.record(cur)
            b.synchronize()
            ts.append(a.elapsed_time(b))
        ts.sort()
        return round(ts[len(ts) // 2], 4)

    out = {
        "fps_l2_alone": timed(lambda: fps(1, pf)),
        "fps_l3_alone": timed(lambda: fps(2, pf)),
        "grid_l1_alone": timed(lambda: grid0(pf)),
        "ball_l1_alone": timed(None, lambda: ball(0, ph, True)),
        "ball_l2_alone": timed(None, lambda: ball(1, ph, False)),
        "ball_l3_alone": timed(None, lambda: ball(2, ph, False)),
        "fps_l2_beside_ball_l1": timed(lambda: fps(1, pf), lambda: ball(0, ph, True)),
        "chain_f_alone": timed(chain_f),
        "chain_h_alone": timed(None, chain_h),
        "phase2_both": timed(chain_f, lambda: chain_h(True)),
        "tuning": {k: int(L.tgn_get_tuning(k.encode(), -999)) for k in ("fps_plain", "fps_bucket_min", "ball_bitmap")},
        "env": {k: v for k, v in os.environ.items() if k.startswith("TGN_")},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
