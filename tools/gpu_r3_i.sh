#!/bin/bash
# round 3: preprocess runner with the native per-scan loader and the adaptive FPS batch
set -u
mkdir -p gpurun_out/r3i
export TMPDIR=/tmp
O=gpurun_out/r3i
echo "cores: $(nproc)" | tee $O/preprocess.txt
echo "== pytest"; timeout 600 python -m pytest tests/test_preprocess_io.py -q -x 2>&1 | tail -3
export TGN_SYNTH_DIR=/tmp/tgn_synth
t0=$(date +%s.%N)
for cfg in "32 0 2" "32 0 3" "64 0 2" "48 0 2" "32 16 2" "32 48 2" "16 0 3" "32 0 1"; do
  set -- $cfg
  echo "== batch<=$1 workers=$2 (0 = default) samplers=$3" | tee -a $O/preprocess.txt
  TGN_PREPROCESS_SAMPLERS=$3 TGN_PREPROCESS_WORKERS=$2 timeout 600 python tools/preprocess_sharded.py --synthetic 512 --batch $1 --save_data_path /tmp/tgn_out 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/preprocess.txt
done

