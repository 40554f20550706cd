#!/bin/bash
set -u
export TMPDIR=/tmp
run() {
  timeout 300 python bench.py --steps 30 --warmup 5 --cpu-meshes 0 --no-alt "$@" 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d.get('kernel_ms_per_step')['fps_l1'])
except Exception as e: print('FAILED', e)"
}
for rep in 1; do
for lib in A C D E F; do
  echo "== lib $lib"
  TGN_LIB_PATH=$GRAFT_REPO_ROOT/tools/libtgn_$lib.so run --pipeline 0
  TGN_LIB_PATH=$GRAFT_REPO_ROOT/tools/libtgn_$lib.so run
done
done
