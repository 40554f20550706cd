#!/bin/bash
# round 3: AMAX hand-off in the bucket FPS kernels: parity, bench
set -u
mkdir -p gpurun_out/r3i
export TMPDIR=/tmp
O=gpurun_out/r3i
echo "== parity"; timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fps_prefix.py -m gpu -q -x -k fps > $O/parity.log 2>&1; rc=$?; tail -3 $O/parity.log
if [ $rc -ne 0 ]; then echo "PARITY FAILED"; exit 0; fi
for r in 1; do timeout 300 python bench.py --steps 40 --warmup 5 --cpu-meshes 0 --no-alt > $O/bench_$r.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("$O/bench_$r.json").read().strip().splitlines()[-1])
print({k:round(d[k],3) for k in ("value","ms_per_step")}, d["kernel_ms_per_step"])
PY
done
