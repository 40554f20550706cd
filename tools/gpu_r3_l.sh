#!/bin/bash
set -u
mkdir -p gpurun_out/r3l
export TMPDIR=/tmp
O=gpurun_out/r3l
export TGN_SYNTH_DIR=/tmp/tgn_synth
timeout 600 python tools/preprocess_sharded.py --synthetic 512 --batch 32 --save_data_path /tmp/tgn_out 2>&1 | grep -v amdgpu.ids | tail -1 > $O/first.txt
for cfg in "32 1" "32 2" "64 1" "64 2"; do
  set -- $cfg
  timeout 300 python tools/experiments/preprocess_stage_times.py /tmp/tgn_synth $1 $2 2>&1 | grep -v amdgpu.ids | tee -a $O/stages.txt
done
