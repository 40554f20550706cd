#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest sa"; timeout 900 python -m pytest tests/test_gpu_sa_fused.py tests/test_gpu_modules.py -m gpu -x -q 2>&1 | tail -25
echo "== sa bench"; timeout 300 python tools/sa_bench.py 2>&1 | tail -5
echo "== bench fused"; timeout 300 python bench.py --fused 1 --steps 10 --warmup 3 --cpu-meshes 0 2>&1 | tail -1 | tee gpurun_out/bench_fused.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], d.get('fused_levels'))"
echo "== bench fused one stream"; timeout 300 python bench.py --fused 1 --pipeline 0 --steps 10 --warmup 3 --cpu-meshes 0 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], d.get('fused_levels'))"
