#!/bin/bash
set -u
export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --steps 12 --warmup 4 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d.get('kernel_ms_per_step'))
except Exception as e: print('FAILED', e)"
}
echo "== parity (default lib = perm1)"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fps_prefix.py -m gpu -q -x -k "fps" 2>&1 | tail -2
for lib in "" tools/libtgn_perm0.so tools/libtgn_perm2.so; do
  echo "== lib=$lib"
  TGN_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} run --pipeline 0 --ball-split 0
  TGN_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} run --ball-split 0
  TGN_LIB_PATH=${lib:+$GRAFT_REPO_ROOT/$lib} python tools/fps_stats.py 2>&1 | grep "B=256" | head -1
done
