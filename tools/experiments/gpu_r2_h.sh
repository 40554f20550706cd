#!/bin/bash
set -u
export TMPDIR=/tmp
for v in "" "--group-impl 2,7,7 --group-max-blocks 512,128,128" "--group-impl 2,7,7 --group-max-blocks 512,256,256" "--group-impl 2,2,7 --group-max-blocks 512,512,128" "--group-impl 1,2,2 --group-max-blocks 0,512,512" "--group-impl 2 --group-max-blocks 768" "--group-impl 2 --group-max-blocks 1024,512,512" "--group-impl 2,7,7 --group-max-blocks 512,128,128 --group-policy 16,2,2"; do
  echo "== bench $v"; timeout 300 python bench.py --steps 10 --warmup 3 --cpu-meshes 0 --no-alt $v 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), d['kernel_ms_per_step'], 'group frac', round(d['roofline_group']['frac'],3))"
done
echo "== pointnetpp forward"; timeout 300 python tools/pointnetpp_forward_bench.py 2>&1 | tail -8
echo "== preprocess"; timeout 600 python tools/preprocess_sharded.py --synthetic 64 --save_data_path /tmp/tgn_pre_out 2>&1 | tail -1
timeout 300 python -m pytest tests/test_preprocess_io.py -m gpu -q 2>&1 | tail -2
