#!/bin/bash
# round-4 checkpoint: the new parity cases (verbose), the whole GPU suite, smoke, the driver's bench command with `secondary`
set -u
mkdir -p gpurun_out/r4_check
export TMPDIR=/tmp
O=gpurun_out/r4_check
t0=$SECONDS
timeout 900 python -m pytest tests/test_gpu_r4_parity.py -q -s -m gpu 2>&1 | tail -60 | tee $O/pytest_r4.txt
echo "r4 parity took $((SECONDS-t0)) s"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $O/pytest_gpu.txt
echo "pytest took $((SECONDS-t0)) s"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
tail -3 $O/bench_20.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4_check/bench_20.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","steps")}, d["roofline"]["achieved"], d["roofline_group"]["frac"], d["cpu_baseline"]["value"])
print(json.dumps(d.get("secondary"), indent=1)[:6000])
PY
echo "total $((SECONDS-t0)) s"
