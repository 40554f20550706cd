#!/bin/bash
# FPS level 2 (4096 -> 1024, beside the level-1 ball query) on the 8-wave bucket kernel instead of the 4-wave one
set -u
for rep in 1 2; do for cfg in "" "512,16"; do echo "== TGN_FPS_BUCKET_CONFIG=$cfg"; TGN_FPS_BUCKET_CONFIG=$cfg timeout 300 python bench.py --steps 50 --warmup 5 --cpu-meshes 0 --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],3), k['fps_l1'], k['fps_l2'], k['ball_l1'])"; done; done
