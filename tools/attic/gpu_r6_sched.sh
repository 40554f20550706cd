#!/bin/bash
# round 6: phase-2 schedule variants on the chunked ball-query kernel
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_sched
mkdir -p $O
run() {  # name, args...
  local name=$1; shift
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --secondary 0 --no-alt "$@" 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), round(d['ms_per_step'],4), d['kernel_ms_per_step'])" 2>&1 | tail -1
}
run base_F
run grid_H --grid-stream H
run grid_own --grid-stream own
run plain_F --fps23 plain
run plain_H --fps23 plain --grid-stream H
run plain_own --fps23 plain --grid-stream own
run base_F2
run grid_H2 --grid-stream H
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --cpu-meshes 0 --no-alt --no-kernel-timing --secondary 0 --grid-stream H > /dev/null 2>&1)
python tools/timeline.py $(find $O/tl -name "*kernel_trace.csv" | head -1) > $O/timeline_H.txt; tail -24 $O/timeline_H.txt
rm -rf $O/tl
