#!/bin/bash
# late ball queries in front of the groupings (beside the next step's FPS level 1): --ball-split 4 (level 3), 5 (levels 2-3)
set -u
for sp in 0 4 5; do for dl in -1 100; do echo "== --ball-split $sp --group-delay-us $dl"; timeout 300 python bench.py --steps 50 --warmup 5 --cpu-meshes 0 --no-alt --ball-split $sp --group-delay-us $dl 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'])"; done; done
