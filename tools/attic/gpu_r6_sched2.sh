#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_sched2
mkdir -p $O
run() {  # name, env..., -- args
  local name=$1; shift
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --secondary 0 --no-alt "$@" 2>$O/$name.err | tail -1 > $O/$name.json
  python -c "import json; d=json.load(open('$O/$name.json')); print('$name', round(d['value']), round(d['ms_per_step'],4), d['kernel_ms_per_step'])" 2>&1 | tail -1
}
run own --grid-stream own
TGN_FPS_BUCKET_CONFIG=512,16 run own_b512 --grid-stream own
TGN_FPS_BUCKET_CONFIG=256,16 run own_b256 --grid-stream own
run own2 --grid-stream own
TGN_FPS_BUCKET_CONFIG=512,16 run own_b512_2 --grid-stream own
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --cpu-meshes 0 --no-alt --no-kernel-timing --secondary 0 --grid-stream own > /dev/null 2>&1)
python tools/timeline.py $(find $O/tl -name "*kernel_trace.csv" | head -1) > $O/timeline_own.txt; tail -26 $O/timeline_own.txt
rm -rf $O/tl
timeout 900 python tools/secondary_bench.py > $O/secondary.json 2>$O/secondary.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6_sched2/secondary.json"))
for k in ("shape_A_tree_ties","shape_A_with_h2d"):
    print(k, json.dumps(d.get(k))[:900])
print({k:(v.get("value"),v.get("ms"),v.get("error")) for k,v in d.items() if isinstance(v,dict)})
PY
