#!/bin/bash
# rows-kernel image size again, with nt stores and the level-3 query in front of the groupings
set -u
for f in 1572 2176 2620 3144; do echo "== TGN_GROUP_IMAGE_FLOATS=$f"; TGN_GROUP_IMAGE_FLOATS=$f timeout 300 python bench.py --steps 50 --warmup 5 --cpu-meshes 0 --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'])"; done
