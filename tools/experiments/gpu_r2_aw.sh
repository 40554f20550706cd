#!/bin/bash
# default --ball-split 4: equality with the one-stream schedule, then the spacer length again
set -u
echo "== hotpath_check"; timeout 600 python tools/hotpath_check.py 2>&1 | tail -8
echo "== pipeline_stress"; timeout 600 python tools/pipeline_stress.py 2>&1 | tail -4
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -k "hotpath or pipelin or schedule" 2>&1 | tail -3
for dl in 60 100 130 150; do echo "== --group-delay-us $dl"; timeout 300 python bench.py --steps 50 --warmup 5 --cpu-meshes 0 --no-alt --group-delay-us $dl 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], d['config']['schedule'][-90:])"; done
echo "== shape B"; for sp in 0 4; do timeout 300 python bench.py --shape B --steps 20 --warmup 3 --cpu-meshes 0 --no-alt --ball-split $sp 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"; done
