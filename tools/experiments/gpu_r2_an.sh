#!/bin/bash
# pairs kernel with index loads one trip ahead: parity, stand-alone and in-bench timing
set -u
mkdir -p gpurun_out
echo "== pytest group/hotpath"; timeout 900 python -m pytest tests -m gpu -q -k "group or hotpath or Group or pipelin" 2>&1 | tail -4
echo "== group_bench pairs"; timeout 300 python tools/group_bench.py pairs 2>&1 | grep -E "impl|^ +(1|2|10) " | cut -c1-60
echo "== bench 20"; timeout 600 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt > gpurun_out/bench_an20.log 2>&1; tail -1 gpurun_out/bench_an20.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
echo "== bench 50"; timeout 600 python bench.py --steps 50 --warmup 5 --cpu-meshes 0 --no-alt > gpurun_out/bench_an50.log 2>&1; tail -1 gpurun_out/bench_an50.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
echo "== bench 50, delay 100"; timeout 600 python bench.py --steps 50 --warmup 5 --cpu-meshes 0 --no-alt --group-delay-us 100 > gpurun_out/bench_an50b.log 2>&1; tail -1 gpurun_out/bench_an50b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
