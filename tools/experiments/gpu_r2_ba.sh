#!/bin/bash
# with the faster FPS level 1 (record written by every lane) the work beside it is what binds: spacer length / level-3 query placement again
set -u
export TGN_LIB_PATH=toothgroupnetwork_amd/csrc/libtgn_alt2.so
for rep in 1 2; do for opt in "--group-delay-us 150" "--group-delay-us 100" "--group-delay-us 50" "--group-delay-us 0" "--ball-split 0 --group-delay-us 150" "--ball-split 0 --group-delay-us 80"; do echo "== $opt"; timeout 300 python bench.py --steps 50 --warmup 5 --cpu-meshes 0 --no-alt $opt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print(round(d['value']), round(d['ms_per_step'],3), k['fps_l1'], k['fps_l2'], k['ball_l1'], k['group_l3'])"; done; done
