#!/bin/bash
# round-5 evidence: the driver's bench command (with `secondary`), rocprofv3 kernel stats + timeline of it, HBM traffic of the hot-path
# kernels (PMC, separate passes) with the groupings at the grid they have in the phased schedule, the FPS floor, the atomic floor, the
# PointNet++ forward kernel sequence
set -u
export TMPDIR=/tmp
O=gpurun_out/evidence5
mkdir -p $O
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err; echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r5 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt --secondary 0 > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; cut -c1-140 $O/kernel_stats.csv | head -14
tail -1 $O/rocprof.log | cut -c1-400 > $O/bench_under_rocprof.json
rm -rf $O/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --cpu-meshes 0 --no-alt --no-kernel-timing --secondary 0 > /dev/null 2>&1)
python tools/timeline.py $(find $O/tl -name "*kernel_trace.csv" | head -1) > $O/timeline.txt; tail -22 $O/timeline.txt
rm -rf $O/tl
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -o pmc -- \
      python $GRAFT_REPO_ROOT/bench.py --pipeline 0 --group-max-blocks 256 --steps 2 --warmup 1 --cpu-meshes 0 --no-alt --no-kernel-timing --secondary 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1)
  tail -1 gpurun_out/pmc_$C.log | cut -c1-120
done
python tools/pmc_summary.py gpurun_out $O r05 2>&1 | tail -12
timeout 300 python tools/fps_floor_report.py > $O/fps_floor.txt 2>/dev/null; tail -6 $O/fps_floor.txt | cut -c1-200
tools/_bin/atomic_floor > $O/atomic_floor.txt; cat $O/atomic_floor.txt | cut -c1-200
python tools/gather_family_variants.py 0 3 5 > $O/gather_variants.txt 2>/dev/null; cat $O/gather_variants.txt
(cd /tmp && REPS=5 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o pnpp -- python $GRAFT_REPO_ROOT/tools/pnpp_forward_run.py > $GRAFT_REPO_ROOT/$O/pnpp_run.log 2>&1)
grep "ms per forward" $O/pnpp_run.log
python tools/forward_sequence.py $(find $O/tr -name "*kernel_trace.csv" | head -1) 8 > $O/pnpp_forward_sequence.txt
cp $(find $O/tr -name "*kernel_stats.csv" | head -1) $O/pnpp_forward_kernel_stats.csv
rm -rf $O/tr
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/forward_sharded.py --gpus 1 --synthetic 32 --model pointnetpp 2>/dev/null | tail -1 | cut -c1-400 > $O/forward_sharded_1rank.json; cat $O/forward_sharded_1rank.json | cut -c1-300
timeout 300 python tools/forward_sharded.py --gpus 2 --backend gloo --synthetic 32 --model pointnetpp 2>/dev/null | tail -1 | cut -c1-700 > $O/forward_sharded_2gloo_ranks_one_gpu.json; cut -c1-300 $O/forward_sharded_2gloo_ranks_one_gpu.json
timeout 600 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 --cpu-meshes 0 --secondary 0 2>/dev/null | tail -1 > $O/bench_2gloo_ranks_one_gpu.json; python -c "
import json; d=json.load(open('$O/bench_2gloo_ranks_one_gpu.json')); print({k:d[k] for k in ('value','n_gpus','backend','self_spawned','distinct_devices')}, [ (r['rank'],r['device_index'],r['pci_bus_id']) for r in d['ranks']])"
