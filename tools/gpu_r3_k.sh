#!/bin/bash
# round 3: preprocess runner after the owner-wave FPS kernel and the faster OBJ token parsers
set -u
mkdir -p gpurun_out/r3k
export TMPDIR=/tmp
O=gpurun_out/r3k
echo "cores: $(nproc)" | tee $O/preprocess.txt
timeout 600 python -m pytest tests/test_preprocess_io.py -q -x 2>&1 | tail -2
export TGN_SYNTH_DIR=/tmp/tgn_synth
for cfg in "32 0 2" "32 48 2" "32 64 2" "48 48 2" "64 64 2" "32 32 1" "64 96 3"; do
  set -- $cfg
  echo "== batch<=$1 workers=$2 (0 = default) samplers=$3" | tee -a $O/preprocess.txt
  TGN_PREPROCESS_SAMPLERS=$3 TGN_PREPROCESS_WORKERS=$2 timeout 600 python tools/preprocess_sharded.py --synthetic 512 --batch $1 --save_data_path /tmp/tgn_out 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/preprocess.txt
done
