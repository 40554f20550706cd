#!/bin/bash
# round-6 baseline: the headline on 256 distinct scans against the 16-tiled inputs of rounds 1-5, the phase-2 kernels in isolation,
# the timeline, the SQ instruction mix of the ball query
set -u
export TMPDIR=/tmp
O=gpurun_out/r6_base
mkdir -p $O
for U in 0 16 0 16; do
  timeout 600 python bench.py --steps 20 --warmup 5 --cpu-meshes 0 --secondary 0 --no-alt --unique $U 2>/dev/null | tail -1 > $O/bench_u$U.json
  python -c "import json; d=json.load(open('$O/bench_u$U.json')); print('unique', $U, d['value'], d['ms_per_step'], d['kernel_ms_per_step'])"
done
timeout 600 python tools/phase2_bench.py > $O/phase2.json 2>$O/phase2.err; cat $O/phase2.json
TGN_FPS_BUCKET_MIN=4097 timeout 600 python tools/phase2_bench.py > $O/phase2_plain.json 2>$O/phase2_plain.err; cat $O/phase2_plain.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tl -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --cpu-meshes 0 --no-alt --no-kernel-timing --secondary 0 > /dev/null 2>&1)
python tools/timeline.py $(find $O/tl -name "*kernel_trace.csv" | head -1) > $O/timeline.txt; tail -24 $O/timeline.txt
rm -rf $O/tl
bash tools/gpu_pmc_sq.sh > $O/sq.txt 2>&1; grep -A14 "bitmap" $O/sq.txt | head -40
