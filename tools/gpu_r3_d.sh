#!/bin/bash
# round 3: flag hand-off in the 8-wave bucket FPS kernel (TGN_FPS_HANDOFF=1) -- parity, then A/B timing
set -u
mkdir -p gpurun_out/r3d
export TMPDIR=/tmp
O=gpurun_out/r3d
echo "== parity with the flag hand-off"
TGN_FPS_HANDOFF=1 timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fps_prefix.py tests/test_gpu_modules.py -m gpu -q -x -k "fps or FPS or sample or pipelined or shape_b" > $O/parity.log 2>&1
rc=$?; tail -4 $O/parity.log
if [ $rc -ne 0 ]; then echo "PARITY FAILED rc=$rc"; exit 0; fi
for h in 0 1 0 1; do
  echo "== bench TGN_FPS_HANDOFF=$h"; TGN_FPS_HANDOFF=$h timeout 200 python bench.py --steps 30 --warmup 5 --cpu-meshes 0 --no-alt > $O/bench_h$h.json 2> $O/bench_h$h.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_h$h.json").read().strip().splitlines()[-1])
    print({k:round(d[k],3) for k in ("value","ms_per_step")}, d["kernel_ms_per_step"])
except Exception as e: print("no result", e)
PY
done
