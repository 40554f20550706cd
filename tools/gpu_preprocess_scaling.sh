#!/bin/bash
# BASELINE config 5, the part a one-GPU box CAN measure: how the HOST side of the sharded preprocess runner scales when 1 / 2 / 4 / 8
# ranks (gloo; all on the one GPU of the box) share the host cores -- OBJ parsing, normals, packing, np.save are what bound an 8-GPU
# node (the FPS launches of 8 ranks on ONE GPU serialise here, so the 8-rank number is a LOWER bound for 8 GPUs).
set -u
export TMPDIR=/tmp
export TGN_SYNTH_DIR=/tmp/tgn_synth_scaling
N=${N:-1536}
O=gpurun_out/preprocess_scaling
mkdir -p $O
nproc | tee $O/host.txt; lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread" | tee -a $O/host.txt
# the synthetic scans are written once (first run), every later run finds them
timeout 900 python tools/preprocess_sharded.py --synthetic $N --save_data_path /tmp/pp_out_1 --batch 64 > $O/world1_first.json 2> $O/world1_first.err
for W in 1 2 4 8; do
  for WORKERS in 0 8 16; do
    [ "$W" = 1 ] && [ "$WORKERS" != 0 ] && [ "$WORKERS" != 16 ] && continue
    rm -rf /tmp/pp_out_$W
    if [ "$W" = 1 ]; then
      TGN_PREPROCESS_WORKERS=$WORKERS timeout 600 python tools/preprocess_sharded.py --synthetic $N --save_data_path /tmp/pp_out_$W --batch 64 \
          > $O/world${W}_w${WORKERS}.json 2> $O/world${W}_w${WORKERS}.err
    else
      TGN_PREPROCESS_WORKERS=$WORKERS timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port $((29600 + W)) \
          tools/preprocess_sharded.py --synthetic $N --save_data_path /tmp/pp_out_$W --batch 64 --backend gloo \
          > $O/world${W}_w${WORKERS}.json 2> $O/world${W}_w${WORKERS}.err
    fi
    echo "world $W workers/rank ${WORKERS:-default}: $(tail -1 $O/world${W}_w${WORKERS}.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: (round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('value','scans','seconds','seconds_load','seconds_fps','batches','n_gpus')})" 2>/dev/null || tail -2 $O/world${W}_w${WORKERS}.err)"
  done
done | tee $O/summary.txt
