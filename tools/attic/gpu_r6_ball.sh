#!/bin/bash
# round 6: the chunked ball-query kernel -- parity, then the step and the per-kernel times against round 2's bitmap kernel
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_ball
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ball or sample_and_group" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_hotpath_configs.py -x -q 2>&1 | tail -4
for V in 2 1 2 1; do
  TGN_BALL_BITMAP=$V timeout 600 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --secondary 0 --no-alt 2>/dev/null | tail -1 > $O/bench_v$V.json
  python -c "import json; d=json.load(open('$O/bench_v$V.json')); print('ball_bitmap', $V, round(d['value']), round(d['ms_per_step'],4), d['kernel_ms_per_step'])"
done
timeout 600 python tools/phase2_bench.py > $O/phase2_v2.json 2>$O/phase2.err; cat $O/phase2_v2.json
TGN_BALL_BITMAP=1 timeout 600 python tools/phase2_bench.py > $O/phase2_v1.json 2>>$O/phase2.err; cat $O/phase2_v1.json
