#!/usr/bin/env python3
"""What slows FPS level 1 (24000 -> 4096, 256 scans) when another kernel shares its CUs?  Times the FPS launch alone and
beside four synthetic co-runners (pure VALU, LDS, streaming stores, L2 loads), each sized to outlast it."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from toothgroupnetwork_amd import _lib, synth
dev = torch.device("cuda")
C = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcorun.so"))
L = _lib.lib()
B, N, S = 256, 24000, 4096
xyz = torch.from_numpy(np.stack([synth.arch_cloud(N, s, False) for s in range(4)])).to(dev).repeat(B // 4, 1, 1).contiguous()
idx = torch.empty(B, S, dtype=torch.int32, device=dev)
big = torch.empty(1 << 30, dtype=torch.float32, device=dev)       # 4 GiB store target
src = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
sink = torch.zeros(16, device=dev)
s_fps, s_co = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
P = lambda t: ctypes.c_void_p(t.data_ptr())
ST = lambda s: ctypes.c_void_p(s.cuda_stream)
def fps():
    _lib.check(L.tgn_furthestsampling_dense(B, N, S, _lib.ptr(xyz), None, _lib.ptr(idx), None, _lib.FPS_LOCAL_INDEX, ST(s_fps)))
co = {
    "alone": lambda: None,
    "pure VALU (fma chain), 2048 blocks": lambda: C.corun_valu(P(sink), 2048, 60000, ST(s_co)),
    "LDS reads, 2048 blocks": lambda: C.corun_lds(P(sink), 2048, 30000, ST(s_co)),
    "streaming stores 4 GiB x 8": lambda: C.corun_store(P(big), ctypes.c_size_t(big.numel()), 4096, 8, ST(s_co)),
    "L2-resident loads, 2048 blocks": lambda: C.corun_l2load(P(src), P(sink), (1 << 20) - 1, 2048, 40000, ST(s_co)),
}
for name, launch in co.items():
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s_fps); fps(); b.record(s_fps)          # FPS first: its workgroups own the CUs, the co-runner squeezes in
        c0.record(s_co); launch(); c1.record(s_co)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
        cot = c0.elapsed_time(c1)
    print(f"FPS level 1 beside [{name:38s}]: {best:6.3f} ms   (co-runner ran {cot:7.2f} ms)", flush=True)
