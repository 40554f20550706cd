#!/bin/bash
# round 3, call 3: cleaned-up library (group.hip without the ring kernel, HotPath with a measured plan, two-layer fused levels):
# all GPU tests, the headline bench, the fused bench + its rocprof kernel stats, Shape B
set -u
mkdir -p gpurun_out/r3c
export TMPDIR=/tmp
O=gpurun_out/r3c
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tee $O/pytest.log | tail -8
echo "== bench"; timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["config"]["schedule"]); print(d["kernel_ms_per_step"]); print(d.get("with_fps_prefix_identity"))
PY
echo "== bench fused"; timeout 600 python bench.py --fused 1 --steps 10 --warmup 2 --cpu-meshes 0 > $O/bench_fused.json 2> $O/bench_fused.err; tail -c 1500 $O/bench_fused.json; tail -3 $O/bench_fused.err
echo "== bench shape B"; timeout 600 python bench.py --shape B --steps 20 --warmup 3 --cpu-meshes 0 --no-alt > $O/bench_shapeB.json 2> $O/bench_shapeB.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3c/bench_shapeB.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["config"]["schedule"]); print(d["path_hbm"])
PY
echo "== rocprof fused"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_fused -o r3 -- python $GRAFT_REPO_ROOT/bench.py --fused 1 --steps 5 --warmup 2 --cpu-meshes 0 > $GRAFT_REPO_ROOT/$O/rocprof_fused.log 2>&1); tail -c 600 $O/rocprof_fused.log; f=$(find $O/prof_fused -name "*kernel_stats.csv" | head -1); cut -c1-180 $f | head -14
