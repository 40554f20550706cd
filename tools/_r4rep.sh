export TMPDIR=/tmp
mkdir -p gpurun_out/r4_check
for i in 1 2 3 4; do
timeout 600 python -m pytest tests/test_gpu_r4_parity.py -q -s -m gpu -k "point_transformer_whole or training_step" 2>&1 | grep -E "PointTransformerSeg at|loss terms|train-mode|gradient samples|passed|failed|AssertionError" | cut -c1-700
done > gpurun_out/r4_check/flaky.txt 2>&1
cat gpurun_out/r4_check/flaky.txt
