#!/bin/bash
set -u
export TMPDIR=/tmp
for v in "--group-impl 7 --group-policy 16" "--group-impl 7 --group-policy 2" "--group-impl 7 --group-policy 0" "--group-impl 7 --group-policy 16 --group-max-blocks 128" "--group-impl 7 --group-policy 16 --group-max-blocks 256" "--group-impl 7 --group-policy 16 --group-max-blocks 0" "--group-impl 7 --group-policy 2 --group-max-blocks 0" "--group-impl 7 --group-policy 16 --pipeline 0"; do
  echo "== bench $v"; timeout 300 python bench.py --steps 10 --warmup 3 --cpu-meshes 0 --no-alt $v 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'group frac', round(d['roofline_group']['frac'],3))"
done
