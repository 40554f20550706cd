#!/bin/bash
# round 3: owner-wave form of the large-cloud FPS kernel: parity, then timing against the touched-list form
set -u
mkdir -p gpurun_out/r3j
export TMPDIR=/tmp
O=gpurun_out/r3j
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fps_prefix.py -m gpu -q -x -k "fps or large or prefix" 2>&1 | tail -4
timeout 600 python -m pytest tests/test_gpu_modules.py tests/test_preprocess_io.py -m gpu -q -x 2>&1 | tail -3
echo "== timing" | tee $O/fps_large.txt
timeout 300 python tools/attic/experiments/fps_large_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/fps_large.txt
echo "== preprocess runner" | tee $O/preprocess.txt
export TGN_SYNTH_DIR=/tmp/tgn_synth
for cfg in "32 0 2" "32 0 1" "64 0 2" "32 0 3"; do
  set -- $cfg
  echo "== batch<=$1 workers=$2 (0 = default) samplers=$3" | tee -a $O/preprocess.txt
  TGN_PREPROCESS_SAMPLERS=$3 TGN_PREPROCESS_WORKERS=$2 timeout 600 python tools/preprocess_sharded.py --synthetic 512 --batch $1 --save_data_path /tmp/tgn_out 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/preprocess.txt
done
