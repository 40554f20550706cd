import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from toothgroupnetwork_amd import resample, synth, pointops as P
xs = [np.ascontiguousarray(synth.arch_cloud(108000 - 1000 * (i % 5), seed=i, with_normals=False), dtype=np.float32) for i in range(32)]
resample.fps_batch(xs[:4], 24000)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); idx = resample.fps_batch(xs, 24000); t1 = time.perf_counter()
    print("fps_batch(32 scans)", round((t1 - t0) * 1e3, 1), "ms")
# phases
t0 = time.perf_counter(); packed = np.concatenate(xs, axis=0); t1 = time.perf_counter()
pts = torch.from_numpy(packed).cuda(); torch.cuda.synchronize(); t2 = time.perf_counter()
counts = np.array([x.shape[0] for x in xs]); off = torch.from_numpy(np.cumsum(counts).astype(np.int32)).cuda()
noff = torch.arange(1, 33, dtype=torch.int32, device="cuda") * 24000
torch.cuda.synchronize(); t3 = time.perf_counter()
i = P.furthestsampling(pts, off, noff); torch.cuda.synchronize(); t4 = time.perf_counter()
h = i.cpu().numpy(); t5 = time.perf_counter()
print("concat", round((t1-t0)*1e3,1), "h2d", round((t2-t1)*1e3,1), "fps", round((t4-t3)*1e3,1), "d2h", round((t5-t4)*1e3,1))
