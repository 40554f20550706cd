#!/bin/bash
# timeline of the phased schedule + SQ instruction counters of the final build
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tl -o t -- python $R/bench.py --steps 8 --warmup 2 --cpu-meshes 0 --no-alt --no-kernel-timing > $R/gpurun_out/tl.log 2>&1)
python tools/timeline.py gpurun_out/prof_tl/t_kernel_trace.csv > gpurun_out/timeline.txt; tail -30 gpurun_out/timeline.txt
bash tools/gpu_pmc_sq.sh > gpurun_out/sq_counters.txt 2>&1; grep -A18 "fps_bucket_kernel<512, 48" gpurun_out/sq_counters.txt | head -40
