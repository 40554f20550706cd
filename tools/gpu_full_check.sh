#!/bin/bash
# checkpoint: the whole GPU suite, smoke, the driver's bench command
set -u
mkdir -p gpurun_out/full_check
export TMPDIR=/tmp
O=gpurun_out/full_check
t0=$SECONDS
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
echo "pytest took $((SECONDS-t0)) s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/full_check/bench_20.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","steps")}, d["roofline"]["frac"], d["roofline_group"]["frac"], d["cpu_baseline"]["value"])
PY
echo "total $((SECONDS-t0)) s"
