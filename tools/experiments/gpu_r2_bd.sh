#!/bin/bash
# wave-uniform hand-off words stored by every lane in the remaining FPS kernels (resident, streaming, large-cloud): parity + timing
set -u
timeout 400 python -m pytest tests -m gpu -q -k "fps or FPS or prefix or resample or preprocess" 2>&1 | tail -2
timeout 100 python tools/fps_stats.py > /dev/null 2>&1
timeout 200 python - <<'PY'
import torch, numpy as np, sys
sys.path.insert(0, '.')
from toothgroupnetwork_amd import pointnet2_utils as U, synth, resample
dev = torch.device('cuda')
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts)
for (B, N, S) in [(256, 4096, 1024), (256, 1024, 256), (1, 100000, 24000), (64, 100000, 24000)]:
    xyz = torch.from_numpy(np.stack([synth.arch_cloud(N, s, False) for s in range(min(B, 4))])).to(dev).repeat((B + 3) // 4, 1, 1)[:B].contiguous()
    U.fps_prefix_clear()
    print(f"farthest_point_sample B={B} N={N} S={S}: {t(lambda: (U.fps_prefix_clear(), U.farthest_point_sample(xyz, S))):.3f} ms", flush=True)
PY
