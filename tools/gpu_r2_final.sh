#!/bin/bash
# round 2, final build: smoke, all GPU tests, headline bench line, rocprofv3 kernel stats (headline + fused), PMC traffic,
# extra benches.  Everything lands under gpurun_out/ (scratch); the summaries that matter are copied into profiles/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6
echo "== bench"; timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-2500
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r2 -- python $R/bench.py --steps 3 --warmup 1 --cpu-meshes 0 --no-alt > $R/gpurun_out/rocprof.log 2>&1); cat $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1) | cut -c1-170 | head -16
echo "== rocprof fused"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fused -o r2f -- python $R/bench.py --fused 1 --pipeline 0 --steps 3 --warmup 1 --cpu-meshes 0 > $R/gpurun_out/rocprof_fused.log 2>&1); grep "^{\"metric\"" gpurun_out/rocprof_fused.log | tail -1 | cut -c1-600; cat $(find gpurun_out/prof_fused -name "*kernel_stats.csv" | head -1) | cut -c1-170 | head -10
echo "== pmc"; bash tools/gpu_pmc.sh 2>&1 | tail -30
{
echo "## bench.py --shape B (not the headline)"; timeout 300 python bench.py --shape B --steps 10 --warmup 3 --cpu-meshes 0 --no-alt 2>&1 | tail -1 | cut -c1-1200
echo "## bench.py --pipeline 0 (one stream)"; timeout 300 python bench.py --pipeline 0 --steps 10 --warmup 3 --cpu-meshes 0 --no-alt 2>&1 | tail -1 | cut -c1-1500
echo "## bench.py --ball-stream 0 --group-gate 0 (round-1 two-stream schedule, this round's kernels)"; timeout 300 python bench.py --ball-stream 0 --group-gate 0 --steps 20 --warmup 5 --cpu-meshes 0 --no-alt 2>&1 | tail -1 | cut -c1-1500
echo "## tools/train_step_bench.py (BASELINE config 3)"; timeout 600 python tools/train_step_bench.py 2>&1 | tail -1
echo "## tools/pt_forward_bench.py (BASELINE config 4)"; timeout 600 python tools/pt_forward_bench.py 2>&1 | tail -2
echo "## tools/pointnetpp_forward_bench.py"; timeout 300 python tools/pointnetpp_forward_bench.py 2>&1 | tail -6
echo "## tools/sa_bench.py"; timeout 300 python tools/sa_bench.py 2>&1 | tail -3
echo "## tools/preprocess_sharded.py --synthetic 64"; timeout 900 python tools/preprocess_sharded.py --synthetic 64 --save_data_path /tmp/tgn_pre_out 2>&1 | tail -1
echo "## tools/group_bench.py rows"; timeout 300 python tools/group_bench.py rows 2>&1 | grep -E "impl|^ +7 +(16|2) "
echo "## tools/group_bench.py pairs"; timeout 300 python tools/group_bench.py pairs 2>&1 | grep -E "impl|^ +(1|2|10) " | cut -c1-40
echo "## tools/fps_stats.py"; timeout 120 python tools/fps_stats.py 2>&1 | grep "N=24000"
} > gpurun_out/extra_bench.txt 2>&1
cat gpurun_out/extra_bench.txt | cut -c1-400
