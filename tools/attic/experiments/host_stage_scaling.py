#!/usr/bin/env python3
"""Which HOST stage of the sharded preprocess runner stops scaling on a many-core node: P processes x T threads each run ONE stage
in a loop -- `load` (tgn_scan_open/take: json + OBJ parse + normals + rows, pooled scratch), `save` (np.save of a (24000, 7) float64
array, what preprocess_data.py:58 writes) or `pack` (copy of the coordinates into a page-locked staging buffer) -- and report scans/s
for the whole machine.  No GPU work.   python tools/experiments/host_stage_scaling.py <synthetic root> <out dir>"""
import json, os, subprocess, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import os, sys, time, threading
sys.path.insert(0, %r)
import numpy as np
from toothgroupnetwork_amd import preprocess
stage, root, out, T, rank, world, seconds = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7])
pairs = preprocess.list_scans(os.path.join(root, "obj"), os.path.join(root, "json"))[rank::world]
pools = (preprocess.ArrayPool(7, np.float64), preprocess.ArrayPool(3, np.float32))
rows = np.random.default_rng(rank).normal(size=(24000, 7))
count = [0] * T
stop = time.perf_counter() + seconds
def work(k):
    i = k
    while time.perf_counter() < stop:
        if stage == "load":
            lv, _, _, x32 = preprocess.load_scan_native(*pairs[i %% len(pairs)], with_xyz32=True, pools=pools)
            pools[0].give(lv); pools[1].give(x32)
        elif stage == "save":
            np.save(os.path.join(out, f"r{rank}_t{k}_{i %% 64}.npy"), rows)
        i += T
        count[k] += 1
ths = [threading.Thread(target=work, args=(k,)) for k in range(T)]
t0 = time.perf_counter()
[t.start() for t in ths]; [t.join() for t in ths]
print(sum(count) / (time.perf_counter() - t0))
''' % ROOT


def run(stage, root, out, P, T, seconds=3.0):
    os.makedirs(out, exist_ok=True)
    ps = [subprocess.Popen([sys.executable, "-c", WORKER, stage, root, out, str(T), str(r), str(P), str(seconds)], stdout=subprocess.PIPE, text=True)
          for r in range(P)]
    return sum(float(p.communicate()[0].strip().splitlines()[-1]) for p in ps)


if __name__ == "__main__":
    root, out = sys.argv[1], sys.argv[2]
    res = {}
    for stage in ("load", "save"):
        for P, T in ((1, 1), (1, 8), (1, 16), (1, 32), (2, 16), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8)):
            v = run(stage, root, out, P, T)
            res[f"{stage} {P}x{T}"] = round(v, 1)
            print(f"{stage:5s} {P:2d} processes x {T:2d} threads: {v:9.1f} scans/s  ({v / (P * T):7.1f} per thread)", flush=True)
    print(json.dumps(res))
