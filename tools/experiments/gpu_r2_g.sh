#!/bin/bash
# round 2: full check -- smoke, all GPU tests, headline bench, rocprof kernel stats, PMC traffic, extra benches
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12
echo "== bench"; timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-1500
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r2 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-meshes 0 --no-alt > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1); cat $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) | cut -c1-160 | head -16
echo "== rocprof fused"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_fused -o r2f -- python $GRAFT_REPO_ROOT/bench.py --fused 1 --pipeline 0 --steps 3 --warmup 1 --cpu-meshes 0 > $GRAFT_REPO_ROOT/gpurun_out/rocprof_fused.log 2>&1); cat $(find gpurun_out/prof_fused -name "*kernel_stats.csv" | head -1) | cut -c1-160 | head -12
echo "== pmc"; bash tools/gpu_pmc.sh 2>&1 | tail -30
echo "== preprocess sharded (synthetic)"; timeout 900 python tools/preprocess_sharded.py --synthetic 64 --save_data_path /tmp/tgn_pre_out 2>&1 | tail -1
echo "== pointnetpp forward"; timeout 300 python tools/pointnetpp_forward_bench.py 2>&1 | tail -4
echo "== pt bench"; timeout 300 python tools/pt_bench.py 2>&1 | tail -12
