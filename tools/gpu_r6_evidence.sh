#!/bin/bash
# round-6 evidence: the driver's bench command (with `secondary`), rocprofv3 kernel stats + timeline of it, HBM traffic of the hot-path
# kernels (PMC, separate passes, kernel-trace only), the FPS floors, the two-rank gloo runs on the one GPU
set -u
export TMPDIR=/tmp
O=gpurun_out/evidence6
mkdir -p $O
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r6 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt --secondary 0 > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; cut -c1-140 $O/kernel_stats.csv | head -14
tail -1 $O/rocprof.log | cut -c1-400 > $O/bench_under_rocprof.json
rm -rf $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --cpu-meshes 0 --no-alt --no-kernel-timing --secondary 0 > /dev/null 2>&1)
python tools/timeline.py $(find $O/tl -name "*kernel_trace.csv" | head -1) > $O/timeline.txt; tail -26 $O/timeline.txt
rm -rf $O/tl
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --pipeline 0 --group-max-blocks 256 --steps 2 --warmup 1 --cpu-meshes 0 --no-alt --no-kernel-timing --secondary 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1)
  tail -1 gpurun_out/pmc_$C.log | cut -c1-120
done
python tools/pmc_summary.py gpurun_out $O r06 2>&1 | tail -12
tools/_bin/fps_floor > $O/fps_floor.txt 2>&1; cut -c1-300 $O/fps_floor.txt
timeout 600 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --cpu-meshes 0 --secondary 0 2>/dev/null | tail -1 > $O/bench_2gloo_ranks_one_gpu.json; python -c "
import json; d=json.load(open('$O/bench_2gloo_ranks_one_gpu.json')); print({k:d.get(k) for k in ('value','n_gpus','backend','self_spawned','distinct_devices','error')}, [ (r['rank'],r['device_index'],r['pci_bus_id'],r.get('numa_pin',{}).get('pinned')) for r in d.get('ranks',[])])"
timeout 300 python bench.py --gpus 4 --steps 3 --warmup 1 --cpu-meshes 0 --secondary 0 2>/dev/null | tail -1 | cut -c1-600 > $O/bench_4ranks_on_1gpu_error.json; cat $O/bench_4ranks_on_1gpu_error.json; echo
timeout 300 python tools/forward_sharded.py --gpus 2 --backend gloo --synthetic 32 --model pointnetpp 2>/dev/null | tail -1 | cut -c1-700 > $O/forward_sharded_2gloo_ranks_one_gpu.json; cut -c1-300 $O/forward_sharded_2gloo_ranks_one_gpu.json
