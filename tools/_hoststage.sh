export TMPDIR=/tmp TGN_SYNTH_DIR=/tmp/tgn_synth_scaling
mkdir -p gpurun_out/preprocess_scaling
python tools/preprocess_sharded.py --synthetic 256 --save_data_path /tmp/pp_out_1 --batch 64 > /dev/null 2>&1
df -h /tmp | tail -1; mount | grep -E " /tmp | / " | head -3
python tools/experiments/host_stage_scaling.py /tmp/tgn_synth_scaling /tmp/pp_stage_out 2>&1 | tee gpurun_out/preprocess_scaling/host_stage_scaling.txt
