#!/bin/bash
# store policy of the grouping kernels per level, inside the phased schedule
set -u
for pol in 16,16,16 2,16,16 2,2,2 2,2,16 16,2,2 0,0,0; do echo "== --group-policy $pol"; timeout 300 python bench.py --steps 30 --warmup 5 --cpu-meshes 0 --no-alt --group-policy $pol 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'])"; done
