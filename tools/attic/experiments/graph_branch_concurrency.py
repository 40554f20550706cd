"""Does a replayed HIP graph run independent branches concurrently?  Two long single-workgroup kernels (FPS of one cloud each)
captured on forked streams: replay time against the time of one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from toothgroupnetwork_amd import pointops as P, synth
dev = torch.device("cuda")
pts = [torch.from_numpy(synth.arch_cloud(24000, seed=i, with_normals=False)).to(dev) for i in range(2)]
off = P.register_offsets(torch.tensor([24000], dtype=torch.int32, device=dev), [24000])
noff = P.register_offsets(torch.tensor([6000], dtype=torch.int32, device=dev), [6000])

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

one = timed(lambda: P.furthestsampling(pts[0], off, noff))
side = torch.cuda.Stream()
def two_streams():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        P.furthestsampling(pts[1], off, noff)
    P.furthestsampling(pts[0], off, noff)
    torch.cuda.current_stream().wait_stream(side)
eager2 = timed(two_streams)
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
with torch.cuda.stream(cap):
    two_streams()
    cap.synchronize()
    with torch.cuda.graph(g, stream=cap):
        two_streams()
graph2 = timed(g.replay)
print(f"one FPS launch {one:.2f} ms; two on forked streams, eager {eager2:.2f} ms; the same as a replayed graph {graph2:.2f} ms")
