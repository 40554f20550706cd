#!/bin/bash
# fused set-abstraction bench: one stream vs the software-pipelined schedules
set -u
for opt in "--pipeline 0" "--pipeline 1" "--pipeline 1 --ball-stream 2 --group-gate 1" "--pipeline 1 --ball-stream 1"; do echo "== --fused 1 $opt"; timeout 300 python bench.py --fused 1 $opt --steps 20 --warmup 3 --cpu-meshes 0 --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d.get('kernel_ms_per_step'), d['config']['schedule'][:60])"; done
