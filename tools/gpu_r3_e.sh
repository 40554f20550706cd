#!/bin/bash
# flag hand-off variants: TGN_FPS_HANDOFF = 1 plain, 3 s_sleep in the polls, 5 low priority while polling, 7 both
set -u
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
O=gpurun_out/r3e
echo "== parity with the flag hand-off"
TGN_FPS_HANDOFF=25 timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fps" > $O/parity.log 2>&1
rc=$?; tail -3 $O/parity.log
if [ $rc -ne 0 ]; then echo "PARITY FAILED rc=$rc"; exit 0; fi
for h in 0 9 17 25 27; do
  TGN_FPS_HANDOFF=$h timeout 200 python bench.py --steps 20 --warmup 4 --cpu-meshes 0 --no-alt > $O/bench_h$h.json 2> $O/bench_h$h.err; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_h$h.json").read().strip().splitlines()[-1])
    print("handoff $h", {k:round(d[k],3) for k in ("value","ms_per_step")}, d["kernel_ms_per_step"])
except Exception as e: print("handoff $h: no result", e)
PY
done
