#!/bin/bash
# round 2, call A: GPU tests (parity seams + grouping v2), grouping kernel sweep, bench with grouping variants
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30
echo "== group bench"; timeout 600 python tools/group_bench.py > gpurun_out/group_bench.log 2>&1; cat gpurun_out/group_bench.log
for v in "--group-impl 1" "--group-impl 2 --group-policy 16" "--group-impl 2 --group-policy 0" "--group-impl 2 --group-policy 2" "--group-impl 2 --group-policy 16 --group-max-blocks 256" "--group-impl 2 --group-policy 16 --group-max-blocks 1024" "--group-impl 2 --group-policy 16 --group-max-blocks 0"; do
  echo "== bench $v"; timeout 300 python bench.py --steps 10 --warmup 3 --cpu-meshes 0 --no-alt $v 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'group frac', round(d['roofline_group']['frac'],3))"
done
echo "== bench one stream v2"; timeout 300 python bench.py --steps 10 --warmup 3 --cpu-meshes 0 --no-alt --pipeline 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'])"
