"""Where a sampler thread of preprocess.preprocess_scans spends its time: fps_batch taken apart (pack into the page-locked
buffer, copy to the device, kernel, indices back, per-scan post-processing), measured INSIDE the running loop (loaders and
writers active).  Usage: python tools/experiments/preprocess_stage_times.py <synthetic root> [batch] [samplers]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from toothgroupnetwork_amd import pointops, preprocess, resample, synth

root, batch, samplers = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 32, int(sys.argv[3]) if len(sys.argv) > 3 else 2
acc, lock, local = {}, threading.Lock(), threading.local()


def add(k, dt):
    with lock:
        acc[k] = acc.get(k, 0.0) + dt


def fps_batch(xyz_list, npoint):
    st = getattr(local, "stream", None)
    if st is None:
        st = local.stream = torch.cuda.Stream()
    with torch.cuda.stream(st):
        dev = torch.device("cuda")
        t0 = time.perf_counter()
        counts = np.array([x.shape[0] for x in xyz_list], dtype=np.int64)
        offset_np = np.cumsum(counts).astype(np.int32)
        total = int(counts.sum())
        stage = getattr(local, "buf", None)
        if stage is None or stage.shape[0] < total:
            stage = local.buf = torch.empty((max(total, 1 << 20), 3), dtype=torch.float32, pin_memory=True)
        t1 = time.perf_counter()
        host = stage[:total].numpy()
        pos = 0
        for x, n in zip(xyz_list, counts):
            np.copyto(host[pos:pos + n], x[:, :3], casting="unsafe")
            pos += int(n)
        t2 = time.perf_counter()
        pts = stage[:total].to(dev, non_blocking=True)
        offset = torch.from_numpy(offset_np).to(dev)
        new_offset = torch.arange(1, len(xyz_list) + 1, dtype=torch.int32, device=dev) * int(npoint)
        st.synchronize()
        t3 = time.perf_counter()
        idx_d = pointops.furthestsampling(pts, offset, new_offset)
        st.synchronize()
        t4 = time.perf_counter()
        idx = idx_d.cpu().numpy().reshape(len(xyz_list), npoint)
        t5 = time.perf_counter()
        starts = np.concatenate([[0], offset_np[:-1]]).astype(np.int32)
        out = [idx[i] - starts[i] for i in range(len(xyz_list))]
        t6 = time.perf_counter()
    for k, dt in (("alloc", t1 - t0), ("pack", t2 - t1), ("h2d", t3 - t2), ("kernel", t4 - t3), ("d2h", t5 - t4), ("post", t6 - t5)):
        add(k, dt)
    add("launches", 1)
    add("scans", len(xyz_list))
    return out


resample.fps_batch([synth.arch_cloud(30000, seed=1, with_normals=False)], 24000)
pairs = preprocess.list_scans(os.path.join(root, "obj"), os.path.join(root, "json"))
t0 = time.perf_counter()
st = preprocess.preprocess_scans(pairs, "/tmp/tgn_out_stage", batch=batch, fps_batch=fps_batch, samplers=samplers)
dt = time.perf_counter() - t0
print(f"{st['scans']} scans in {dt:.3f} s = {st['scans'] / dt:.0f} scans/s; waited for loads {st['seconds_load']:.3f} s; batch<={batch}, samplers {samplers}")
n = acc.pop("launches")
print(f"  {int(n)} launches, {acc.pop('scans') / n:.1f} scans each; per launch (ms): " + ", ".join(f"{k} {v / n * 1e3:.1f}" for k, v in acc.items()))
