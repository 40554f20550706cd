#!/bin/bash
# round-4 evidence: rocprofv3 kernel stats of the driver's bench command (with `secondary`), the step timeline, HBM traffic (PMC, separate
# passes), SQ counters of the bf16x3 set-abstraction kernel at its 256 x 256 tile
set -u
mkdir -p gpurun_out/evidence4
export TMPDIR=/tmp
O=gpurun_out/evidence4
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o r4 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-meshes 0 --no-alt > $GRAFT_REPO_ROOT/$O/rocprof.log 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats.csv; cut -c1-140 $O/kernel_stats.csv | head -40
tail -1 $O/rocprof.log | cut -c1-300 > $O/bench_under_rocprof.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --cpu-meshes 0 --no-alt --no-kernel-timing --secondary 0 > /dev/null 2>&1)
python tools/timeline.py $(find $O/tl -name "*kernel_trace.csv" | head -1) > $O/timeline.txt; tail -30 $O/timeline.txt
bash tools/gpu_pmc.sh > $O/pmc.log 2>&1; tail -30 $O/pmc.log
python tools/pmc_summary.py gpurun_out $O r04 2>&1 | tail -5
bash tools/gpu_pmc_sa.sh > $O/pmc_sa_256.txt 2>&1; tail -22 $O/pmc_sa_256.txt
