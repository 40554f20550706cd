#!/bin/bash
# FPS: result row parked by wave 0 out of line (untaken scalar branch for the other waves) -- A/B on one box
# (libtgn_alt.so = the build before the change)
set -u
for rep in 1 2 3; do for lib in toothgroupnetwork_amd/csrc/libtgn_alt.so ""; do echo "== TGN_LIB_PATH=$lib"; TGN_LIB_PATH=$lib timeout 300 python bench.py --steps 50 --warmup 5 --cpu-meshes 0 --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],3), k['fps_l1'], k['fps_l2'], k['ball_l1'])"; done; done
